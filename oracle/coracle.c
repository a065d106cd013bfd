/*
 * oracle/coracle.c -- CPU restatement (plain C + OpenMP) of the heavy loops of the InsMOS
 * inference hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker / reported baseline.  The product path
 * (insmos_amd/) never links, imports or calls anything in here.
 *
 * PARITY STATUS
 *   - rotated BEV IoU / NMS and the point-in-box one-hot features restate in-repo reference
 *     code and are pinned bit-exactly against the reference's own sources compiled into
 *     oracle/_ref (tests/golden/*.npz, tests/test_oracle_golden.py).
 *   - sparse-conv / neighbour-search restate the *semantics* of MinkowskiEngine / spconv 2.3.6
 *     ([dep-knowledge]; neither library is in /root/reference or this image) at the reference's
 *     call sites models/MinkowskiEngine/minkunet.py:55-137,139-181 and
 *     models/backbones_3d/spconv_unet.py:120-207,267-416.  PARITY UNPINNED for those two
 *     (no golden vectors exist in the reference); they are pinned only against dense
 *     torch.nn.functional.conv3d on densified grids and the known voxel/pair counts of the
 *     survey's synthetic scene (tests/test_oracle_semantics.py).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/coracle.c -o oracle/_build/libcoracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int co_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Neighbour table ("rulebook") by sorted-key binary search.
 *   out_q[(k*n_out + o)] is the 64-bit key of the input coordinate that output o reads through
 *   kernel offset k, or UINT64_MAX when that coordinate is invalid (out of range / not divisible).
 *   in_keys: ascending unique keys of the input coordinate set, in_perm[pos] = input row id
 *   (or -1 when that voxel was dropped by the max-voxel cap).
 *   nbr[k*n_out + o] = input row id or -1.
 * Semantics restated: ME kernel maps (minkunet.py:55-124 call sites), spconv indice pairs
 * (spconv_unet.py:120-207 call sites).  The key construction lives in oracle/ref_ops.py.
 * ---------------------------------------------------------------------------------------- */
void co_nbr_lookup(const uint64_t* q, int64_t nq, const uint64_t* in_keys, const int32_t* in_perm,
                   int64_t n_in, int32_t* nbr) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        uint64_t key = q[i];
        int32_t r = -1;
        if (key != UINT64_MAX && n_in > 0) {
            int64_t lo = 0, hi = n_in; /* first pos with in_keys[pos] >= key */
            while (lo < hi) {
                int64_t mid = (lo + hi) >> 1;
                if (in_keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < n_in && in_keys[lo] == key) r = in_perm ? in_perm[lo] : (int32_t)lo;
        }
        nbr[i] = r;
    }
}

/* ------------------------------------------------------------------------------------------
 * Sparse convolution forward, output-stationary:  out[o,:] = sum_k in[nbr[k][o],:] @ W[k]
 *   in  (n_in , ld_in ) fp32, first cin  columns used
 *   W   (K, cin, cout) fp32
 *   out (n_out, ld_out) fp32, first cout columns written
 * nbr == NULL means K == 1 with the identity map (1x1 convolution / Linear).
 * Accumulation order: k ascending, cin ascending, fp32 -- a plain fmaf-free restatement of the
 * gather -> GEMM -> scatter-add that ME (minkunet.py:139-181) and spconv (spconv_unet.py:297-402)
 * perform; their own summation order is unspecified (atomics / cuBLAS).
 * ---------------------------------------------------------------------------------------- */
void co_sparse_conv(const float* in, int64_t ld_in, int cin, const int32_t* nbr, int K,
                    int64_t n_out, const float* W, int cout, float* out, int64_t ld_out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t o = 0; o < n_out; ++o) {
        float acc[512];
        for (int c = 0; c < cout; ++c) acc[c] = 0.f;
        for (int k = 0; k < K; ++k) {
            int64_t i = nbr ? nbr[(int64_t)k * n_out + o] : o;
            if (i < 0) continue;
            const float* x = in + i * ld_in;
            const float* w = W + (int64_t)k * cin * cout;
            for (int ci = 0; ci < cin; ++ci) {
                float xv = x[ci];
                const float* wr = w + (int64_t)ci * cout;
                for (int c = 0; c < cout; ++c) acc[c] += xv * wr[c];
            }
        }
        float* y = out + o * ld_out;
        for (int c = 0; c < cout; ++c) y[c] = acc[c];
    }
}

/* ------------------------------------------------------------------------------------------
 * Rotated BEV IoU.  Restates models/bbox_post_process/src/iou3d_cpu.cpp:59-229 (== the device
 * code of iou3d_nms_kernel.cu:35-234): polygon clipping of two rotated rectangles, <=16 edge
 * intersections + <=8 contained corners, angular bubble sort, shoelace area; fp32 throughout.
 * The reference's unqualified cos/sin/atan2/fabs on float arguments resolve (C++ <math.h>
 * overloads; likewise in CUDA device code) to the float versions, restated as cosf/sinf/atan2f/fabsf.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y; } pt2;
static const float IOU_EPS = 1e-8f;

static inline float crs2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }
static inline float crs3(pt2 p1, pt2 p2, pt2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static inline float fmin2(float a, float b) { return a > b ? b : a; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }

static int seg_bbox_overlap(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
    return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
           fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

/* iou3d_cpu.cpp:75-87 */
static int corner_in_box(const float* box, pt2 p) {
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float ac = cosf(-box[6]), as = sinf(-box[6]); /* C++ overload resolution: cos(float) -> cosf */
    float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}

/* iou3d_cpu.cpp:89-117 */
static int seg_intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2* ans) {
    if (!seg_bbox_overlap(p0, p1, q0, q1)) return 0;
    float s1 = crs3(q0, p1, p0);
    float s2 = crs3(p1, q1, p0);
    float s3 = crs3(p0, q1, q0);
    float s4 = crs3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = crs3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static inline pt2 rot_about(pt2 c, float ac, float as, pt2 p) {
    pt2 r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

/* iou3d_cpu.cpp:128-219 */
float co_box_overlap(const float* A, const float* B) {
    float a_ang = A[6], b_ang = B[6];
    float adx = A[3] / 2, bdx = B[3] / 2, ady = A[4] / 2, bdy = B[4] / 2;
    pt2 ca = {A[0], A[1]}, cb = {B[0], B[1]};
    pt2 pa[5] = {{A[0] - adx, A[1] - ady}, {A[0] + adx, A[1] - ady}, {A[0] + adx, A[1] + ady}, {A[0] - adx, A[1] + ady}, {0, 0}};
    pt2 pb[5] = {{B[0] - bdx, B[1] - bdy}, {B[0] + bdx, B[1] - bdy}, {B[0] + bdx, B[1] + bdy}, {B[0] - bdx, B[1] + bdy}, {0, 0}};
    float aac = cosf(a_ang), aas = sinf(a_ang);
    float bac = cosf(b_ang), bas = sinf(b_ang);
    for (int k = 0; k < 4; ++k) {
        pa[k] = rot_about(ca, aac, aas, pa[k]);
        pb[k] = rot_about(cb, bac, bas, pb[k]);
    }
    pa[4] = pa[0];
    pb[4] = pb[0];
    pt2 poly[16];
    pt2 ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            if (seg_intersection(pa[i + 1], pa[i], pb[j + 1], pb[j], &poly[cnt])) {
                /* Point(double,double) constructor in the reference: the sum is formed in
                   double from two floats, then stored to float -- identical to a float add. */
                ctr.x = ctr.x + poly[cnt].x;
                ctr.y = ctr.y + poly[cnt].y;
                cnt++;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (corner_in_box(A, pb[k])) { ctr.x += pb[k].x; ctr.y += pb[k].y; poly[cnt++] = pb[k]; }
        if (corner_in_box(B, pa[k])) { ctr.x += pa[k].x; ctr.y += pa[k].y; poly[cnt++] = pa[k]; }
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            /* point_cmp, iou3d_cpu.cpp:124-126: atan2(float,float) resolves to the float overload */
            if (atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x) > atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x)) {
                pt2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t;
            }
        }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        pt2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
        pt2 v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
        area += crs2(u, v);
    }
    return (float)(fabsf(area) / 2.0);
}

/* iou3d_cpu.cpp:221-229 */
float co_iou_bev(const float* A, const float* B) {
    float sa = A[3] * A[4], sb = B[3] * B[4];
    float so = co_box_overlap(A, B);
    return so / fmaxf(sa + sb - so, IOU_EPS);
}

void co_iou_bev_matrix(const float* a, int na, const float* b, int nb, float* out) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(int64_t)i * nb + j] = co_iou_bev(a + i * 7, b + j * 7);
}

/* Greedy rotated NMS over boxes already sorted by descending score.
 * Restates iou3d_nms_kernel.cu:267-311 (mask bit (i,j), j>i, set iff iou_bev(i,j) > thresh) and the
 * host reduce iou3d_nms.cpp:116-132 (keep i unless a kept earlier box suppressed it).
 * Returns number kept; keep[] receives ascending indices. */
int co_nms_bev(const float* boxes, int n, float thresh, int64_t* keep) {
    if (n <= 0) return 0;
    int cb = (n + 63) / 64;
    uint64_t* mask = (uint64_t*)calloc((size_t)n * cb, sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (co_iou_bev(boxes + i * 7, boxes + j * 7) > thresh) mask[(size_t)i * cb + j / 64] |= 1ULL << (j % 64);
    uint64_t* remv = (uint64_t*)calloc(cb, sizeof(uint64_t));
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (!(remv[i / 64] & (1ULL << (i % 64)))) {
            keep[nk++] = i;
            for (int j = i / 64; j < cb; ++j) remv[j] |= mask[(size_t)i * cb + j];
        }
    }
    free(mask);
    free(remv);
    return nk;
}

/* ------------------------------------------------------------------------------------------
 * Point-in-rotated-box one-hot class features.
 * Restates models/utils/src/Array_Index.cpp:14-79 including its order-dependent early-skip
 * (:48-51): after a box's first hit, voxels farther than extend[d] from that first-hit voxel
 * on any axis are skipped before the inside test.  coords (V,3) int32 [x,y,z] voxel indices,
 * boxes (M,8) fp32 [cx,cy,cz,dx,dy,dz,yaw,label] in voxel units, feat (V,C) int32 in/out.
 * ---------------------------------------------------------------------------------------- */
void co_boxes_to_onehot(const int32_t* coords, int64_t V, const float* boxes, int M, int32_t* feat, int C,
                        int quirk) {
    for (int i = 0; i < M; ++i) {
        const float* b = boxes + (int64_t)i * 8;
        float center[3] = {b[0], b[1], b[2]};
        float extend[3] = {b[3], b[4], b[5]};
        float theta = b[6];
        float cos_t = (float)cos(theta), sin_t = (float)sin(theta);
        int label = (int)b[7];
        int first[3] = {0, 0, 0};
        int have_first = 0;
        for (int64_t j = 0; j < V; ++j) {
            int x = coords[j * 3 + 0], y = coords[j * 3 + 1], z = coords[j * 3 + 2];
            if (quirk && have_first &&
                (x > (first[0] + extend[0]) || x < (first[0] - extend[0]) || y > (first[1] + extend[1]) ||
                 y < (first[1] - extend[1]) || z > (first[2] + extend[2]) || z < (first[2] - extend[2])))
                continue;
            float c0 = x - center[0], c1 = y - center[1], c2 = z - center[2];
            float r0 = c0 * cos_t + c1 * sin_t;
            float r1 = -c0 * sin_t + c1 * cos_t;
            if ((r0 <= extend[0] / 2) && (r0 >= -extend[0] / 2) && (r1 <= extend[1] / 2) && (r1 >= -extend[1] / 2) &&
                (c2 <= extend[2] / 2) && (c2 >= -extend[2] / 2)) {
                if (label > 0 && label <= C) feat[j * C + label - 1] = 1;
                if (!have_first) { have_first = 1; first[0] = x; first[1] = y; first[2] = z; }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Array_Index.find_point_in_instance_bbox_with_yaw (models/utils/src/Array_Index.cpp:85-154), used by
 * scripts/refine.py:196: points (N, ld >= 3) float xyz, boxes (M, 8) [x,y,z,dx,dy,dz,yaw,label]; a point inside box i
 * gets index[j][label-1] = i+1.  Box z is lifted by `out_ground` (float add).  The same order-dependent early skip as
 * find_features_by_bbox_with_yaw: once a box has a first inside point, later points farther than extend[d] from
 * THAT POINT are skipped.  The reference walks the boxes in an OpenMP parallel-for (a write race when two boxes of
 * one class share a point); restated sequentially, i.e. the LARGEST box index wins.
 * ------------------------------------------------------------------------------------------ */
void co_points_in_instance_boxes(const float* pts, int64_t N, int ld, const float* boxes, int M, int32_t* index, int C,
                                 float out_ground, int quirk) {
    for (int i = 0; i < M; ++i) {
        const float* b = boxes + (int64_t)i * 8;
        float center[3] = {b[0], b[1], b[2] + out_ground};
        float extend[3] = {b[3], b[4], b[5]};
        float theta = b[6];
        float cos_t = (float)cos(theta), sin_t = (float)sin(theta);
        int label = (int)b[7];
        float first[3] = {0.f, 0.f, 0.f};
        int have_first = 0;
        for (int64_t j = 0; j < N; ++j) {
            float x = pts[j * ld + 0], y = pts[j * ld + 1], z = pts[j * ld + 2];
            if (quirk && have_first &&
                (x > (first[0] + extend[0]) || x < (first[0] - extend[0]) || y > (first[1] + extend[1]) ||
                 y < (first[1] - extend[1]) || z > (first[2] + extend[2]) || z < (first[2] - extend[2])))
                continue;
            float c0 = x - center[0], c1 = y - center[1], c2 = z - center[2];
            float r0 = c0 * cos_t + c1 * sin_t;
            float r1 = -c0 * sin_t + c1 * cos_t;
            if ((r0 <= extend[0] / 2) && (r0 >= -extend[0] / 2) && (r1 <= extend[1] / 2) && (r1 >= -extend[1] / 2) &&
                (c2 <= extend[2] / 2) && (c2 >= -extend[2] / 2)) {
                if (label > 0 && label <= C) index[j * C + label - 1] = i + 1;
                if (!have_first) { have_first = 1; first[0] = x; first[1] = y; first[2] = z; }
            }
        }
    }
}

/* pairwise rotated-rectangle overlap AREA (iou3d_cpu.cpp:128-206 box_overlap; the CUDA twin is
 * iou3d_nms_kernel.cu:105-196) -- the `boxes_overlap_bev_gpu` half of boxes_iou3d_gpu (iou3d_nms_utils.py:28-61) */
void co_overlap_bev_matrix(const float* a, int na, const float* b, int nb, float* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(int64_t)i * nb + j] = co_box_overlap(a + i * 7, b + j * 7);
}
