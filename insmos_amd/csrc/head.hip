// insmos_amd/csrc/head.hip -- detection head post-processing on device: CenterHead box decode +
// candidate selection, rotated-BEV NMS with an on-device greedy reduce, final gathers, and the
// point-in-rotated-box instance features.  The whole library is compiled with -ffp-contract=off so
// the fp32 expressions below evaluate exactly like the reference's separately-rounded torch / C ops.
#include <cstring>
#include "common.h"

namespace insmos {

// ---------------------------------------------------------------------------------------------------
// CenterHead decode + class-agnostic candidate selection
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cell_of_row(int64_t r, int W, int up, int& row, int& col) {
    if (up == 2) {
        int W0 = W >> 1;
        int64_t site = r >> 2;
        int sub = (int)(r & 3);
        row = 2 * (int)(site / W0) + (sub >> 1);
        col = 2 * (int)(site % W0) + (sub & 1);
    } else {
        row = (int)(r / W);
        col = (int)(r % W);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Candidate sort key, ascending = (window, score descending, cell ascending): window << 54 | (bits(1.0f) - bits(score)) << 24
// | cell.  A sigmoid lies in [0, 1], so its float bits are <= 0x3F800000 and the inverted score fits 30 bits.
#define CAND_CELL_BITS 24
#define CAND_WIN_SHIFT 54

// head rows of the B windows of a batch are stacked: row r = b * n_cells + (row inside the window)
__global__ void k_score_keys(const float* __restrict__ head, int ld, int ncls, int64_t n_cells, int B, int W, int up,
                             float thresh, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                             int32_t* __restrict__ counts) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = r < n_cells * B;
    const int b = in ? (int)(r / n_cells) : -1;
    bool pass = false;
    uint64_t key = INSMOS_INVALID_KEY;
    if (in) {
        const float* h = head + r * ld;
        float best = sigmoidf_(h[0]);
        for (int c = 1; c < ncls; ++c) {
            float p = sigmoidf_(h[c]);
            if (p > best) best = p;  // first max wins
        }
        if (best >= thresh) {
            int row, col;
            cell_of_row(r - (int64_t)b * n_cells, W, up, row, col);
            uint32_t cell = (uint32_t)(row * W + col);
            key = ((uint64_t)b << CAND_WIN_SHIFT) | ((uint64_t)(0x3F800000u - __float_as_uint(best)) << CAND_CELL_BITS) | cell;
            pass = true;
        }
        keys[r] = key;
        vals[r] = (uint32_t)r;
    }
    // candidate count per window: ONE atomic per wave and window (a head that fires on most cells -- an untrained or a
    // mid-training one -- put 300 000 same-address atomics in a row: 3.4 ms for this kernel in the cfg-5 training step)
    const unsigned long long m = __ballot(pass);
    if (m) {
        const int lane = threadIdx.x & 63;
        const int b_first = __shfl(b, __builtin_ctzll(m)), b_last = __shfl(b, 63 - __builtin_clzll(m));
        if (b_first == b_last) {
            if (lane == __builtin_ctzll(m)) atomicAdd(&counts[b_first * 4 + 1], __builtin_popcountll(m));
        } else if (pass) {
            atomicAdd(&counts[b * 4 + 1], 1);   // (a wave that straddles two windows: rare)
        }
    }
}

// grid (candidate blocks, B): window b's candidates follow those of the windows before it in the sorted array
__global__ void k_select_decode(const float* __restrict__ head, int ld, int ncls, int W, int up, float out_factor,
                                float vx, float vy, float x0, float y0, const uint64_t* __restrict__ keys_s,
                                const uint32_t* __restrict__ vals_s, int64_t n_cells, int pre_max,
                                float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ labels,
                                int32_t* __restrict__ cells, int32_t* __restrict__ counts) {
    const int b = blockIdx.y;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int64_t seg = 0;
    for (int q = 0; q < b; ++q) seg += counts[q * 4 + 1];
    const int n_cand = counts[b * 4 + 1];
    if (t == 0) counts[b * 4] = n_cand < pre_max ? n_cand : pre_max;
    if (t >= pre_max) return;
    bool valid = t < n_cand;
    float b7[7] = {0, 0, 0, 0, 0, 0, 0};
    float sc = 0.f;
    int lab = 0, cell = -1;
    if (valid) {
        int64_t r = vals_s[seg + t];
        const float* h = head + r * ld;
        float best = sigmoidf_(h[0]);
        int arg = 0;
        for (int c = 1; c < ncls; ++c) {
            float p = sigmoidf_(h[c]);
            if (p > best) { best = p; arg = c; }
        }
        int row, col;
        cell_of_row(r - (int64_t)b * n_cells, W, up, row, col);
        const float* bx = h + ncls;
        // center_head.py:263-267: (idx + reg) * OUT_SIZE_FACTOR * VOXEL_SIZE + range, left to right in fp32
        float xs = ((float)col + bx[0]) * out_factor * vx + x0;
        float ys = ((float)row + bx[1]) * out_factor * vy + y0;
        b7[0] = xs; b7[1] = ys; b7[2] = bx[2];
        b7[3] = expf(bx[3]); b7[4] = expf(bx[4]); b7[5] = expf(bx[5]);
        b7[6] = atan2f(bx[6], bx[7]);
        sc = best; lab = arg + 1; cell = row * W + col;
    }
    const int64_t o = (int64_t)b * pre_max + t;
    for (int d = 0; d < 7; ++d) boxes[o * 7 + d] = b7[d];
    scores[o] = sc;
    labels[o] = lab;
    cells[o] = cell;
}

// ---------------------------------------------------------------------------------------------------
// rotated BEV IoU (restates iou3d_nms_kernel.cu:35-234; float trig overloads as in the CUDA device code)
// ---------------------------------------------------------------------------------------------------
struct P2 { float x, y; };
__device__ __forceinline__ float crs3(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
__device__ __forceinline__ float fmin2(float a, float b) { return a > b ? b : a; }
__device__ __forceinline__ float fmax2(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ int corner_in_box(const float* box, P2 p) {
    const float MARGIN = 1e-2f;
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p.x - box[0]) * ac + (p.y - box[1]) * (-as);
    float ry = (p.x - box[0]) * as + (p.y - box[1]) * ac;
    return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}

__device__ __forceinline__ int seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2& ans) {
    int ov = fmin2(p0.x, p1.x) <= fmax2(q0.x, q1.x) && fmin2(q0.x, q1.x) <= fmax2(p0.x, p1.x) &&
             fmin2(p0.y, p1.y) <= fmax2(q0.y, q1.y) && fmin2(q0.y, q1.y) <= fmax2(p0.y, p1.y);
    if (!ov) return 0;
    float s1 = crs3(q0, p1, p0), s2 = crs3(p1, q1, p0), s3 = crs3(p0, q1, q0), s4 = crs3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = crs3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

__device__ float box_overlap_bev(const float* A, const float* B) {
    float adx = A[3] / 2, bdx = B[3] / 2, ady = A[4] / 2, bdy = B[4] / 2;
    P2 pa[5] = {{A[0] - adx, A[1] - ady}, {A[0] + adx, A[1] - ady}, {A[0] + adx, A[1] + ady}, {A[0] - adx, A[1] + ady}, {0, 0}};
    P2 pb[5] = {{B[0] - bdx, B[1] - bdy}, {B[0] + bdx, B[1] - bdy}, {B[0] + bdx, B[1] + bdy}, {B[0] - bdx, B[1] + bdy}, {0, 0}};
    float aac = cosf(A[6]), aas = sinf(A[6]), bac = cosf(B[6]), bas = sinf(B[6]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        P2 p = pa[k];
        pa[k].x = (p.x - A[0]) * aac + (p.y - A[1]) * (-aas) + A[0];
        pa[k].y = (p.x - A[0]) * aas + (p.y - A[1]) * aac + A[1];
        P2 q = pb[k];
        pb[k].x = (q.x - B[0]) * bac + (q.y - B[1]) * (-bas) + B[0];
        pb[k].y = (q.x - B[0]) * bas + (q.y - B[1]) * bac + B[1];
    }
    pa[4] = pa[0];
    pb[4] = pb[0];
    P2 poly[16];
    P2 ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_x(pa[i + 1], pa[i], pb[j + 1], pb[j], x)) {
                poly[cnt] = x;
                ctr.x = ctr.x + x.x;
                ctr.y = ctr.y + x.y;
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
            if (atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x) > atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x)) {
                P2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t;
            }
        }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        float ux = poly[k].x - poly[0].x, uy = poly[k].y - poly[0].y;
        float vx = poly[k + 1].x - poly[0].x, vy = poly[k + 1].y - poly[0].y;
        area += ux * vy - uy * vx;
    }
    return (float)(fabsf(area) / 2.0);
}

__device__ __forceinline__ float iou_bev_dev(const float* A, const float* B) {
    float sa = A[3] * A[4], sb = B[3] * B[4];
    float so = box_overlap_bev(A, B);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

__global__ void k_iou_bev(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                          float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)na * nb) return;
    int i = (int)(t / nb), j = (int)(t % nb);
    float A[7], B[7];
    for (int d = 0; d < 7; ++d) { A[d] = a[i * 7 + d]; B[d] = b[j * 7 + d]; }
    out[t] = iou_bev_dev(A, B);
}

// boxes_iou3d_gpu (iou3d_nms_utils.py:28-61), fp32 op for op: BEV overlap x height overlap / union volume
__global__ void k_iou3d(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)na * nb) return;
    int i = (int)(t / nb), j = (int)(t % nb);
    float A[7], B[7];
    for (int d = 0; d < 7; ++d) { A[d] = a[i * 7 + d]; B[d] = b[j * 7 + d]; }
    const float a_max = A[2] + A[5] / 2, a_min = A[2] - A[5] / 2, b_max = B[2] + B[5] / 2, b_min = B[2] - B[5] / 2;
    const float ov_h = fmaxf(fminf(a_max, b_max) - fmaxf(a_min, b_min), 0.f);
    const float ov3 = box_overlap_bev(A, B) * ov_h;
    const float vol_a = A[3] * A[4] * A[5], vol_b = B[3] * B[4] * B[5];
    out[t] = ov3 / fmaxf(vol_a + vol_b - ov3, 1e-6f);
}

// 64 x 64 block of the suppression bitmask (only col block >= row block is ever consumed), 256 threads, two phases:
//   1. every (row, column) pair of the block gets the cheap test -- pairs whose circumscribed circles (grown by the corner MARGIN
//      of check_in_box2d) are disjoint cannot overlap: the reference's clipping yields exactly 0 for them -- and the pairs that
//      pass (~0.7 % on LiDAR scenes) are COMPACTED into a list in LDS (wave ballots, one LDS counter);
//   2. the list is walked with every lane busy: one rotated-IoU evaluation (polygon clipping, ~2000 instructions) per lane, the
//      result OR-ed into the row's 64-bit word in LDS (order-independent).
// Before (one phase: a wave ballot per row right after the test) a wave ran the clipping code whenever ANY of its 64 columns was
// near, i.e. for a third of all (row, 64-column) slots with one or two lanes active.  Same pairs, same function: same words.
__global__ void __launch_bounds__(256) k_nms_mask(const float* __restrict__ boxes, const int32_t* __restrict__ n_dev,
                                                  int max_n, float thresh, uint64_t* __restrict__ mask) {
    const int win = blockIdx.z;                    // one suppression matrix per window of the batch
    const int n = min(n_dev[win * 4], max_n);
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
    const int cbs = (max_n + 63) / 64;
    boxes += (int64_t)win * max_n * 7;
    mask += (int64_t)win * max_n * cbs;
    __shared__ float rowb[64 * 8], colb[64 * 8];   // [box][x, y, z, dx, dy, dz, yaw, circle radius]
    __shared__ unsigned long long words[64];
    __shared__ uint16_t pairs[64 * 64];            // (row << 6) | column of the block's near pairs
    __shared__ int n_pairs;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 128) {
        const int t = threadIdx.x & 63;
        float* dst = (threadIdx.x < 64 ? rowb : colb) + t * 8;
        const int i = (threadIdx.x < 64 ? rb : cb) * 64 + t;
        for (int d = 0; d < 7; ++d) dst[d] = i < n ? boxes[(int64_t)i * 7 + d] : (d >= 3 && d < 6 ? 1.f : 0.f);
        dst[7] = i < n ? 0.5f * sqrtf(dst[3] * dst[3] + dst[4] * dst[4]) + 0.05f : 0.f;
        if (threadIdx.x < 64) words[t] = 0ull;
        if (threadIdx.x == 0) n_pairs = 0;
    }
    __syncthreads();
    const int j = cb * 64 + lane;
    const float bx = colb[lane * 8 + 0], by = colb[lane * 8 + 1], rj = colb[lane * 8 + 7];
    for (int r = 0; r < 16; ++r) {
        const int il = w * 16 + r, i = rb * 64 + il;
        if (i >= n) break;  // wave-uniform
        const float* A = rowb + il * 8;
        bool near = false;
        if (j < n && j > i) {
            const float dx = A[0] - bx, dy = A[1] - by, rr = A[7] + rj;
            near = dx * dx + dy * dy <= rr * rr;
        }
        const unsigned long long bal = __ballot(near);
        if (bal) {   // (wave-uniform)
            int base = 0;
            if (lane == 0) base = atomicAdd(&n_pairs, __popcll(bal));
            base = __shfl(base, 0);
            if (near) pairs[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)((il << 6) | lane);
        }
    }
    __syncthreads();
    const int np = n_pairs;
    for (int t = threadIdx.x; t < np; t += 256) {
        const int il = pairs[t] >> 6, cl = pairs[t] & 63;
        float A[7], B[7];
#pragma unroll
        for (int d = 0; d < 7; ++d) { A[d] = rowb[il * 8 + d]; B[d] = colb[cl * 8 + d]; }
        if (iou_bev_dev(A, B) > thresh) atomicOr(&words[il], 1ull << cl);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = rb * 64 + threadIdx.x;
        if (i < n) mask[(int64_t)i * cbs + cb] = words[threadIdx.x];
    }
}

// greedy reduce of iou3d_nms.cpp:116-132 by ONE wave: lane l owns word l of the removed-set
__global__ void __launch_bounds__(64) k_nms_reduce(const uint64_t* __restrict__ mask, const int32_t* __restrict__ n_dev,
                                                   int max_n, int post_max, int32_t* __restrict__ keep,
                                                   int32_t* __restrict__ counts) {
    const int lane = threadIdx.x;
    const int win = blockIdx.x;                    // one wave per window of the batch
    const int n = min(n_dev[win * 4], max_n);
    const int cbs = (max_n + 63) / 64;
    const int nb = (n + 63) / 64;
    mask += (int64_t)win * max_n * cbs;
    keep += (int64_t)win * post_max;
    counts += win * 4;
    uint64_t remv = 0;  // word `lane`
    int nk = 0;
    uint64_t diag_next = (lane < n) ? mask[(int64_t)lane * cbs] : 0ull;
    for (int blk = 0; blk < nb && nk < post_max; ++blk) {
        const int i_l = blk * 64 + lane;
        const uint64_t diag = diag_next;
        {   // the next block's diagonal words do not depend on this block's outcome: request them now
            const int i_n = i_l + 64;
            diag_next = (i_n < n) ? mask[(int64_t)i_n * cbs + blk + 1] : 0ull;
        }
        // this block's removed word, wave-uniform (scalar registers: v_readlane with a uniform lane index, not an LDS permute)
        const uint32_t ub = __builtin_amdgcn_readfirstlane((uint32_t)blk);
        uint64_t rb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(remv >> 32), ub) << 32) |
                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)remv, ub);
        uint64_t kept = 0;
        const int lim = min(64, n - blk * 64);
        const uint64_t in_range = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
        // walk the SURVIVORS only (the next clear bit of the removed word), not all 64 candidates: a scalar chain of
        // ctz -> readlane -> or per kept box
        uint64_t avail = ~rb & in_range;
        while (avail) {
            const uint32_t i = (uint32_t)__builtin_ctzll(avail);
            kept |= 1ull << i;
            const uint32_t ui = __builtin_amdgcn_readfirstlane(i);
            rb |= ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(diag >> 32), ui) << 32) |
                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)diag, ui);
            avail = ~rb & in_range & ~((2ull << i) - 1ull);
        }
        // emit kept indices in ascending order
        if ((kept >> lane) & 1ull) {
            int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < post_max) keep[pos] = i_l;
        }
        nk += __popcll(kept);
        // fold the kept rows into the later words.  The rows of a block are requested 32 at a time, UNCONDITIONALLY (independent
        // loads, one memory round trip per half block instead of one per 4 kept rows -- the reduce is a chain of L2 latencies,
        // not of bytes: a half block is 32 x 512 B), and only the kept ones are OR-ed in
        if (lane > blk && lane < nb) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t kh = (uint32_t)(kept >> (32 * h));
                if (kh == 0u) continue;   // (wave-uniform)
                uint64_t m[32];
                const uint64_t* row = mask + (int64_t)(blk * 64 + 32 * h) * cbs + lane;
#pragma unroll
                for (int t = 0; t < 32; ++t) m[t] = (blk * 64 + 32 * h + t < n) ? row[(int64_t)t * cbs] : 0ull;
#pragma unroll
                for (int t = 0; t < 32; ++t)
                    if ((kh >> t) & 1u) remv |= m[t];
            }
        }
    }
    if (lane == 0) counts[0] = nk < post_max ? nk : post_max;
}

__global__ void k_gather_preds(const float* __restrict__ cb, const float* __restrict__ cs, const int32_t* __restrict__ cl,
                               const int32_t* __restrict__ keep, const int32_t* __restrict__ nk_dev, int pre_max, int post_max,
                               float* __restrict__ pb, float* __restrict__ psc, int64_t* __restrict__ pl) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= post_max) return;
    const int win = blockIdx.y;
    int nk = nk_dev[win * 4];
    cb += (int64_t)win * pre_max * 7; cs += (int64_t)win * pre_max; cl += (int64_t)win * pre_max;
    keep += (int64_t)win * post_max;
    pb += (int64_t)win * post_max * 7; psc += (int64_t)win * post_max; pl += (int64_t)win * post_max;
    if (t < nk) {
        int s = keep[t];
        for (int d = 0; d < 7; ++d) pb[t * 7 + d] = cb[(int64_t)s * 7 + d];
        psc[t] = cs[s];
        pl[t] = cl[s];
    } else {
        for (int d = 0; d < 7; ++d) pb[t * 7 + d] = 0.f;
        psc[t] = 0.f;
        pl[t] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// point-in-rotated-box one-hot features (Array_Index.cpp:14-79), exact incl. the first-hit early skip
// ---------------------------------------------------------------------------------------------------
// box in voxel units; h = e/2 (division by 2 is exact, so hoisting it changes nothing); rad = conservative
// xy radius (+1 voxel) for an integer-free cull that can only reject voxels the exact test rejects too
struct BoxVox { float c[3], e[3], cs, sn; int label; float h[3]; float rad; float pad[3]; };  // 16 x 4 bytes

__device__ __forceinline__ bool inside_box(const BoxVox& b, int x, int y, int z) {
    float c0 = x - b.c[0], c1 = y - b.c[1], c2 = z - b.c[2];
    if (fabsf(c0) > b.rad || fabsf(c1) > b.rad || fabsf(c2) > b.h[2] + 1.0f) return false;
    float r0 = c0 * b.cs + c1 * b.sn;
    float r1 = -c0 * b.sn + c1 * b.cs;
    return (r0 <= b.h[0]) && (r0 >= -b.h[0]) && (r1 <= b.h[1]) && (r1 >= -b.h[1]) && (c2 <= b.h[2]) && (c2 >= -b.h[2]);
}

struct OneHotP { float lo[3], iv[3]; float istr, mult; };

// boxes -> voxel units of the level (one thread per box of every window: box slot = window * max_boxes + i)
__global__ void k_onehot_boxes(const float* __restrict__ boxes, const int64_t* __restrict__ labels,
                               const int32_t* __restrict__ m_dev, int max_boxes, int B, OneHotP P, int32_t* __restrict__ first,
                               BoxVox* __restrict__ bv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= max_boxes * B) return;
    first[b] = 0x7fffffff;
    const int win = b / max_boxes;
    if (b - win * max_boxes >= min(m_dev[win * 4], max_boxes)) return;
    BoxVox sb;
    const float* bx = boxes + (int64_t)b * 7;
    for (int d = 0; d < 3; ++d) {
        // spconv_unet.py:324-329 on the CUDA path: (x - lo) * (1/v) * (1/stride), then *2 per level
        sb.c[d] = (((bx[d] - P.lo[d]) * P.iv[d]) * P.istr) * P.mult;
        sb.e[d] = ((bx[3 + d] * P.iv[d]) * P.istr) * P.mult;
        sb.h[d] = sb.e[d] / 2;
    }
    const double th = (double)bx[6];
    sb.cs = (float)cos(th);
    sb.sn = (float)sin(th);
    sb.label = (int)(float)labels[b];
    // |rotated coords| <= h  =>  |c0|,|c1| <= sqrt(h0^2 + h1^2); +1 voxel and 1e-3 relative slack
    sb.rad = sqrtf(sb.h[0] * sb.h[0] + sb.h[1] * sb.h[1]) * 1.001f + 1.0f;
    sb.pad[0] = sb.pad[1] = sb.pad[2] = 0.f;
    bv[b] = sb;
}

// grid (voxel blocks, box chunks of 64): the chunk's boxes sit in LDS, each thread owns one voxel.
// PASS 0: first hit per box (min voxel row inside) -> atomicMin.   PASS 1: class bits per voxel -> atomicOr.
// Batches: voxel rows are window-major (coords[:, 0] ascending) and a voxel meets the boxes of ITS window only; a block whose
// 256 rows straddle a window boundary walks the windows it touches, one box chunk in LDS at a time.
template <int PASS>
__global__ void __launch_bounds__(256) k_onehot_scan(const int32_t* __restrict__ m_dev, int max_boxes,
                                                     const int32_t* __restrict__ coords, int64_t n,
                                                     int32_t* __restrict__ first, const BoxVox* __restrict__ bv, int ncls,
                                                     int quirk, uint32_t* __restrict__ vbits,
                                                     unsigned long long* __restrict__ inside,
                                                     const int32_t* __restrict__ orig_of_row,
                                                     const int32_t* __restrict__ row_of_orig) {
    // orig_of_row / row_of_orig (both or neither): the rows were re-ordered (coords.hip: insmos_regroup_rows3d) -- "first hit"
    // means first in the ORIGINAL row order, the one the reference walks (Array_Index.cpp:40-56)
    const int b0 = blockIdx.y * 64;
    __shared__ BoxVox sb[64];
    __shared__ int sfirst[64];
    __shared__ int sfx[64], sfy[64], sfz[64];
    const int64_t jb = (int64_t)blockIdx.x * blockDim.x;
    const int64_t jl = (jb + blockDim.x - 1 < n ? jb + blockDim.x - 1 : n - 1);
    const int w_lo = coords[jb * 4], w_hi = coords[jl * 4];   // block-uniform window range
    const int64_t j = jb + threadIdx.x;
    const bool livej = j < n;
    int4 q = make_int4(-1, 0, 0, 0);
    if (livej) q = *(const int4*)(coords + j * 4);  // [b,z,y,x]
    const int x = q.w, y = q.z, z = q.y;
    const int jo = (livej && orig_of_row) ? orig_of_row[j] : (int)j;   // this voxel's place in the reference's walk
    // PASS 0 does the geometry once and leaves, per voxel and 64-box chunk, the bitmask of boxes containing it;
    // PASS 1 only walks those bits (typically none) and applies the order-dependent rule.
    unsigned long long* im = inside + (size_t)blockIdx.y * (size_t)n + (livej ? j : 0);
    for (int win = w_lo; win <= w_hi; ++win) {
        const int m = min(m_dev[win * 4], max_boxes);
        if (b0 >= m) continue;  // block-uniform
        const int nb = min(64, m - b0);
        const int g0 = win * max_boxes + b0;  // first box slot of this chunk
        __syncthreads();        // the previous window's boxes are no longer read
        if ((int)threadIdx.x < nb) {
            sb[threadIdx.x] = bv[g0 + threadIdx.x];
            if (PASS == 1) {
                const int f = first[g0 + threadIdx.x];
                sfirst[threadIdx.x] = f;
                if (f != 0x7fffffff) {
                    int4 fq = *(const int4*)(coords + (int64_t)(row_of_orig ? row_of_orig[f] : f) * 4);
                    sfx[threadIdx.x] = fq.w; sfy[threadIdx.x] = fq.z; sfz[threadIdx.x] = fq.y;
                }
            }
        }
        __syncthreads();
        if (!livej || q.x != win) continue;
        if (PASS == 0) {
            unsigned long long hit = 0ull;
            for (int i = 0; i < nb; ++i) {
                if (inside_box(sb[i], x, y, z)) {
                    hit |= 1ull << i;
                    atomicMin(&first[g0 + i], jo);
                }
            }
            *im = hit;
        } else {
            unsigned long long hit = *im;
            unsigned bits = 0;
            while (hit) {
                const int i = __ffsll(hit) - 1;
                hit &= hit - 1;
                const BoxVox& bb = sb[i];
                const int f = sfirst[i];
                if (quirk && jo != f) {
                    // Array_Index.cpp:48-51: once a first hit exists (rows after it), skip voxels farther than
                    // extend[d] from the first-hit voxel.  (f <= j here: f is the smallest inside row.)
                    if (x > (sfx[i] + bb.e[0]) || x < (sfx[i] - bb.e[0]) || y > (sfy[i] + bb.e[1]) || y < (sfy[i] - bb.e[1]) ||
                        z > (sfz[i] + bb.e[2]) || z < (sfz[i] - bb.e[2]))
                        continue;
                }
                if (bb.label > 0 && bb.label <= ncls) bits |= 1u << (bb.label - 1);
            }
            if (bits) atomicOr(&vbits[j], bits);
        }
    }
}

__global__ void k_onehot_write(const uint32_t* __restrict__ vbits, int64_t n, int ncls, int pad_to,
                               float* __restrict__ out, int ld_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * pad_to) return;
    const int64_t j = t / pad_to;
    const int c = (int)(t % pad_to);
    out[j * ld_out + c] = (c < ncls && ((vbits[j] >> c) & 1u)) ? 1.0f : 0.0f;
}


// ---- refine stage (scripts/refine.py:196): point -> instance id, Array_Index.cpp:85-154 -------------------------
__device__ __forceinline__ bool inside_box_pt(const BoxVox& b, float x, float y, float z) {
    float c0 = x - b.c[0], c1 = y - b.c[1], c2 = z - b.c[2];
    if (fabsf(c0) > b.rad || fabsf(c1) > b.rad || fabsf(c2) > b.h[2] + 1e-3f) return false;  // can only reject true misses
    float r0 = c0 * b.cs + c1 * b.sn;
    float r1 = -c0 * b.sn + c1 * b.cs;
    return (r0 <= b.h[0]) && (r0 >= -b.h[0]) && (r1 <= b.h[1]) && (r1 >= -b.h[1]) && (c2 <= b.h[2]) && (c2 >= -b.h[2]);
}

__global__ void k_inst_boxes(const float* __restrict__ boxes, const int64_t* __restrict__ labels, int m, float ground,
                             int32_t* __restrict__ first, BoxVox* __restrict__ bv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= m) return;
    first[b] = 0x7fffffff;
    BoxVox sb;
    const float* bx = boxes + (int64_t)b * 7;
    sb.c[0] = bx[0]; sb.c[1] = bx[1]; sb.c[2] = bx[2] + ground;  // Array_Index.cpp:108
    for (int d = 0; d < 3; ++d) { sb.e[d] = bx[3 + d]; sb.h[d] = sb.e[d] / 2; }
    const double th = (double)bx[6];
    sb.cs = (float)cos(th);
    sb.sn = (float)sin(th);
    sb.label = (int)(float)labels[b];  // refine.py:184 carries the label as a float column
    sb.rad = sqrtf(sb.h[0] * sb.h[0] + sb.h[1] * sb.h[1]) * 1.001f + 1e-3f;
    sb.pad[0] = sb.pad[1] = sb.pad[2] = 0.f;
    bv[b] = sb;
}

// grid (point blocks, box chunks of 64).  PASS 0: first inside point per box (atomicMin over the point index).
// PASS 1: index[j][label-1] = max over boxes containing j of (box + 1) -- the sequential walk's "last writer".
template <int PASS>
__global__ void __launch_bounds__(256) k_inst_scan(int m, const float* __restrict__ pts, int ld, int64_t n,
                                                   int32_t* __restrict__ first, const BoxVox* __restrict__ bv, int ncls,
                                                   int quirk, int32_t* __restrict__ index) {
    const int b0 = blockIdx.y * 64;
    if (b0 >= m) return;
    const int nb = min(64, m - b0);
    __shared__ BoxVox sb[64];
    __shared__ int sfirst[64];
    __shared__ float sfx[64], sfy[64], sfz[64];
    if ((int)threadIdx.x < nb) {
        sb[threadIdx.x] = bv[b0 + threadIdx.x];
        if (PASS == 1) {
            const int f = first[b0 + threadIdx.x];
            sfirst[threadIdx.x] = f;
            if (f != 0x7fffffff) {
                const float* q = pts + (int64_t)f * ld;
                sfx[threadIdx.x] = q[0]; sfy[threadIdx.x] = q[1]; sfz[threadIdx.x] = q[2];
            }
        }
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float x = pts[j * ld], y = pts[j * ld + 1], z = pts[j * ld + 2];
    for (int i = 0; i < nb; ++i) {
        const BoxVox& bb = sb[i];
        if (PASS == 0) {
            if (inside_box_pt(bb, x, y, z)) atomicMin(&first[b0 + i], (int)j);
        } else {
            const int f = sfirst[i];
            if (f == 0x7fffffff) continue;
            if (quirk && j != f) {
                if (j < f) continue;  // rows before the first hit are not inside by definition
                if (x > (sfx[i] + bb.e[0]) || x < (sfx[i] - bb.e[0]) || y > (sfy[i] + bb.e[1]) || y < (sfy[i] - bb.e[1]) ||
                    z > (sfz[i] + bb.e[2]) || z < (sfz[i] - bb.e[2]))
                    continue;
            }
            if (bb.label > 0 && bb.label <= ncls && inside_box_pt(bb, x, y, z))
                atomicMax(&index[j * ncls + bb.label - 1], b0 + i + 1);
        }
    }
}

// per-instance point statistics of one class column (refine.py:210-217) and the relabel pass (refine.py:243-285)
__global__ void k_inst_stats(const int32_t* __restrict__ index, int ncls, int col, const int32_t* __restrict__ mos,
                             const float* __restrict__ conf, int64_t n, int m, int32_t* __restrict__ stats) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int id = index[j * ncls + col];
    if (id <= 0 || id > m) return;
    atomicAdd(&stats[(id - 1) * 3 + 0], 1);
    if (mos[j] == 2) atomicAdd(&stats[(id - 1) * 3 + 1], 1);
    if (conf && conf[j * 2 + 1] >= 0.00001f) atomicAdd(&stats[(id - 1) * 3 + 2], 1);
}

__global__ void k_inst_relabel(const int32_t* __restrict__ index, int ncls, int col, const int32_t* __restrict__ decision,
                               int64_t n, int m, int32_t* __restrict__ mos) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int id = index[j * ncls + col];
    if (id <= 0 || id > m) return;
    const int d = decision[id - 1];
    if (d > 0) mos[j] = d;
}

}  // namespace insmos

using namespace insmos;

extern "C" size_t insmos_center_decode_select_ws_bytes(int64_t n_cells) {
    size_t N = (size_t)n_cells;
    return pad256(N * 8) * 2 + pad256(N * 4) * 2 + sort_pairs_u64_u32_temp(N) + 1024;
}

// B windows: head rows stacked window-major (B * H * W rows); candidate arrays are (B, pre_max, .), counts (B, 4) int32 with
// [b][0] = candidates kept (<= pre_max), [b][1] = cells above the threshold.  The workspace is sized for B * H * W cells.
extern "C" int insmos_center_decode_select_b(const float* head, int ld_head, int ncls, int H, int W, int up, int B,
                                             float out_factor, float vx, float vy, float x0, float y0, float score_thresh,
                                             int pre_max, float* cand_boxes, float* cand_scores, int32_t* cand_labels,
                                             int32_t* cand_cell, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    if (!head || ncls <= 0 || ld_head < ncls + 8 || H <= 0 || W <= 0 || (up != 1 && up != 2) || pre_max <= 0 || B < 1 ||
        B > INSMOS_MAX_BATCH || (int64_t)H * W >= (1ll << CAND_CELL_BITS))
        return INSMOS_EINVAL;
    if (up == 2 && ((H & 1) || (W & 1))) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nc = (int64_t)H * W, n = nc * B;
    Bump b(ws, ws_bytes);
    uint64_t* keys = b.take<uint64_t>((size_t)n);
    uint64_t* keys_s = b.take<uint64_t>((size_t)n);
    uint32_t* vals = b.take<uint32_t>((size_t)n);
    uint32_t* vals_s = b.take<uint32_t>((size_t)n);
    size_t st = sort_pairs_u64_u32_temp((size_t)n);
    char* tmp = b.take<char>(st);
    if (!b.ok) return INSMOS_EWORKSPACE;
    HIP_TRY(hipMemsetAsync(counts, 0, (size_t)B * 4 * sizeof(int32_t), s));
    {
        ProfScope ps(KK_DECODE, s);
        INSMOS_LAUNCH(k_score_keys, dim3(cdiv(n, 256)), dim3(256), 0, s, head, ld_head, ncls, nc, B, W, up, score_thresh,
                           keys, vals, counts);
    }
    // (cells below the threshold carry the all-ones key: last in any key width)
    int end_bit = CAND_WIN_SHIFT;
    while ((1 << (end_bit - CAND_WIN_SHIFT)) < B) ++end_bit;
    int rc = sort_pairs_u64_u32(tmp, st, keys, keys_s, vals, vals_s, (size_t)n, 0, end_bit, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_SELECT, s);
        INSMOS_LAUNCH(k_select_decode, dim3(cdiv(pre_max, 256), B), dim3(256), 0, s, head, ld_head, ncls, W, up,
                           out_factor, vx, vy, x0, y0, keys_s, vals_s, nc, pre_max, cand_boxes, cand_scores, cand_labels,
                           cand_cell, counts);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_center_decode_select(const float* head, int ld_head, int ncls, int H, int W, int up,
                                           float out_factor, float vx, float vy, float x0, float y0,
                                           float score_thresh, int pre_max, float* cand_boxes, float* cand_scores,
                                           int32_t* cand_labels, int32_t* cand_cell, int32_t* counts, void* ws,
                                           size_t ws_bytes, void* stream) {
    return insmos_center_decode_select_b(head, ld_head, ncls, H, W, up, 1, out_factor, vx, vy, x0, y0, score_thresh, pre_max,
                                         cand_boxes, cand_scores, cand_labels, cand_cell, counts, ws, ws_bytes, stream);
}

extern "C" size_t insmos_nms_ws_bytes_b(int max_n, int B) {
    size_t cbs = ((size_t)max_n + 63) / 64;
    return pad256((size_t)max_n * cbs * 8 * (size_t)(B < 1 ? 1 : B)) + 1024;
}
extern "C" size_t insmos_nms_ws_bytes(int max_n) { return insmos_nms_ws_bytes_b(max_n, 1); }

// B independent candidate sets: boxes (B, max_n, 7), n_dev / counts (B, 4) int32 (slot 0 used), keep (B, post_max)
extern "C" int insmos_nms_rotated_bev_b(const float* boxes, const int32_t* n_dev, int max_n, float thresh, int post_max, int B,
                                        int32_t* keep, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    if (!boxes || !n_dev || max_n <= 0 || max_n > 4096 || post_max <= 0 || B < 1 || B > INSMOS_MAX_BATCH) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    const int cbs = (max_n + 63) / 64;
    uint64_t* mask = b.take<uint64_t>((size_t)max_n * cbs * B);
    if (!b.ok) return INSMOS_EWORKSPACE;
    {
        ProfScope ps(KK_NMS_MASK, s);
        INSMOS_LAUNCH(k_nms_mask, dim3(cbs, cbs, B), dim3(256), 0, s, boxes, n_dev, max_n, thresh, mask);
    }
    {
        ProfScope ps(KK_NMS_REDUCE, s);
        INSMOS_LAUNCH(k_nms_reduce, dim3(B), dim3(64), 0, s, mask, n_dev, max_n, post_max, keep, counts);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_nms_rotated_bev(const float* boxes, const int32_t* n_dev, int max_n, float thresh, int post_max,
                                      int32_t* keep, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    return insmos_nms_rotated_bev_b(boxes, n_dev, max_n, thresh, post_max, 1, keep, counts, ws, ws_bytes, stream);
}

extern "C" int insmos_iou_bev(const float* a, int na, const float* b, int nb, float* out, void* stream) {
    if (na <= 0 || nb <= 0) return INSMOS_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_IOU, s);
    INSMOS_LAUNCH(k_iou_bev, dim3(cdiv((int64_t)na * nb, 128)), dim3(128), 0, s, a, na, b, nb, out);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_iou3d(const float* a, int na, const float* b, int nb, float* out, void* stream) {
    if (na <= 0 || nb <= 0) return INSMOS_OK;
    if (!a || !b || !out) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_IOU, s);
    INSMOS_LAUNCH(k_iou3d, dim3(cdiv((int64_t)na * nb, 128)), dim3(128), 0, s, a, na, b, nb, out);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// candidate arrays (B, pre_max, .), keep (B, post_max), n_keep_dev (B, 4) -> pred arrays (B, post_max, .)
extern "C" int insmos_gather_preds_b(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                                     const int32_t* keep, const int32_t* n_keep_dev, int pre_max, int post_max, int B,
                                     float* pred_boxes, float* pred_scores, int64_t* pred_labels, void* stream) {
    if (B < 1 || B > INSMOS_MAX_BATCH || post_max <= 0) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_GATHER_PREDS, s);
    INSMOS_LAUNCH(k_gather_preds, dim3(cdiv(post_max, 256), B), dim3(256), 0, s, cand_boxes, cand_scores, cand_labels,
                       keep, n_keep_dev, pre_max, post_max, pred_boxes, pred_scores, pred_labels);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_gather_preds(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                                   const int32_t* keep, const int32_t* n_keep_dev, int post_max, float* pred_boxes,
                                   float* pred_scores, int64_t* pred_labels, void* stream) {
    return insmos_gather_preds_b(cand_boxes, cand_scores, cand_labels, keep, n_keep_dev, 0, post_max, 1, pred_boxes, pred_scores,
                                 pred_labels, stream);
}

extern "C" size_t insmos_boxes_to_onehot_scratch_ints_b(int max_boxes, int B, int64_t n) {
    if (max_boxes <= 0 || n < 0 || B < 1) return 0;
    return 20 * (size_t)max_boxes * (size_t)B + (((size_t)n + 1) & ~(size_t)1) + 2 * (size_t)n * (size_t)cdiv(max_boxes, 64);
}
extern "C" size_t insmos_boxes_to_onehot_scratch_ints(int max_boxes, int64_t n) {
    return insmos_boxes_to_onehot_scratch_ints_b(max_boxes, 1, n);
}

// B windows: pred arrays (B, max_boxes, .), n_boxes_dev (B, 4) int32 (slot 0), voxel rows window-major with the window in
// coords[:, 0]; a voxel is tested against its own window's boxes only.
// insmos_boxes_to_onehot_b over RE-ORDERED rows (insmos_regroup_rows3d): orig_of_row[r] = the row's place in the reference's
// order, row_of_orig = its inverse (both null: the rows are in the reference's order).  Rows must still be window-major.
extern "C" int insmos_boxes_to_onehot_rows(const float* pred_boxes, const int64_t* pred_labels, const int32_t* n_boxes_dev,
                                           int max_boxes, int B, const float* range_lo_host, const float* vsize_host, float stride,
                                           float mult, const int32_t* coords, const int32_t* orig_of_row,
                                           const int32_t* row_of_orig, int64_t n, int ncls, int pad_to, int quirk_exact,
                                           float* out, int ld_out, int32_t* scratch, void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!pred_boxes || !pred_labels || !n_boxes_dev || max_boxes <= 0 || !coords || !out || !scratch || ncls <= 0 ||
        ncls > 31 || pad_to < ncls || ld_out < pad_to || B < 1 || B > INSMOS_MAX_BATCH || (!orig_of_row != !row_of_orig))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    OneHotP P;
    for (int d = 0; d < 3; ++d) {
        P.lo[d] = range_lo_host[d];
        P.iv[d] = 1.0f / vsize_host[d];
    }
    P.istr = 1.0f / stride;
    P.mult = mult;
    const size_t nbx = (size_t)max_boxes * (size_t)B;
    int32_t* first = scratch;
    BoxVox* bv = (BoxVox*)(scratch + ((nbx + 3) & ~(size_t)3));  // 16 ints per box
    uint32_t* vbits = (uint32_t*)(scratch + 20 * nbx);
    unsigned long long* inside = (unsigned long long*)(scratch + 20 * nbx + (((size_t)n + 1) & ~(size_t)1));
    ProfScope ps(KK_ONEHOT, s);
    HIP_TRY(hipMemsetAsync(vbits, 0, (size_t)n * sizeof(uint32_t), s));
    INSMOS_LAUNCH(k_onehot_boxes, dim3(cdiv((int64_t)nbx, 64)), dim3(64), 0, s, pred_boxes, pred_labels, n_boxes_dev,
                       max_boxes, B, P, first, bv);
    dim3 grid(cdiv(n, 256), cdiv(max_boxes, 64));
    INSMOS_LAUNCH(k_onehot_scan<0>, grid, dim3(256), 0, s, n_boxes_dev, max_boxes, coords, n, first, bv, ncls,
                       quirk_exact, vbits, inside, orig_of_row, row_of_orig);
    INSMOS_LAUNCH(k_onehot_scan<1>, grid, dim3(256), 0, s, n_boxes_dev, max_boxes, coords, n, first, bv, ncls,
                       quirk_exact, vbits, inside, orig_of_row, row_of_orig);
    INSMOS_LAUNCH(k_onehot_write, dim3(cdiv(n * pad_to, 256)), dim3(256), 0, s, vbits, n, ncls, pad_to, out, ld_out);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_boxes_to_onehot_b(const float* pred_boxes, const int64_t* pred_labels, const int32_t* n_boxes_dev,
                                        int max_boxes, int B, const float* range_lo_host, const float* vsize_host, float stride,
                                        float mult, const int32_t* coords, int64_t n, int ncls, int pad_to,
                                        int quirk_exact, float* out, int ld_out, int32_t* scratch, void* stream) {
    return insmos_boxes_to_onehot_rows(pred_boxes, pred_labels, n_boxes_dev, max_boxes, B, range_lo_host, vsize_host, stride, mult,
                                       coords, nullptr, nullptr, n, ncls, pad_to, quirk_exact, out, ld_out, scratch, stream);
}

extern "C" int insmos_boxes_to_onehot(const float* pred_boxes, const int64_t* pred_labels, const int32_t* n_boxes_dev,
                                      int max_boxes, const float* range_lo_host, const float* vsize_host, float stride,
                                      float mult, const int32_t* coords, int64_t n, int ncls, int pad_to,
                                      int quirk_exact, float* out, int ld_out, int32_t* scratch, void* stream) {
    return insmos_boxes_to_onehot_b(pred_boxes, pred_labels, n_boxes_dev, max_boxes, 1, range_lo_host, vsize_host, stride, mult,
                                    coords, n, ncls, pad_to, quirk_exact, out, ld_out, scratch, stream);
}

extern "C" int insmos_points_in_instance_boxes(const float* points, int64_t n, int ld_pts, const float* boxes,
                                               const int64_t* labels, int m, float ground_offset, int ncls,
                                               int quirk_exact, int32_t* index, int32_t* scratch, void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!points || ld_pts < 3 || !index || ncls <= 0 || m < 0 || (m > 0 && (!boxes || !labels || !scratch)))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_ONEHOT, s);
    HIP_TRY(hipMemsetAsync(index, 0, (size_t)n * ncls * sizeof(int32_t), s));
    if (m == 0) return INSMOS_OK;
    int32_t* first = scratch;
    BoxVox* bv = (BoxVox*)(scratch + ((m + 3) & ~3));  // 16 ints per box
    INSMOS_LAUNCH(k_inst_boxes, dim3(cdiv(m, 64)), dim3(64), 0, s, boxes, labels, m, ground_offset, first, bv);
    dim3 grid((unsigned)cdiv(n, 256), (unsigned)cdiv(m, 64));
    INSMOS_LAUNCH(k_inst_scan<0>, grid, dim3(256), 0, s, m, points, ld_pts, n, first, bv, ncls, quirk_exact, index);
    INSMOS_LAUNCH(k_inst_scan<1>, grid, dim3(256), 0, s, m, points, ld_pts, n, first, bv, ncls, quirk_exact, index);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_instance_stats(const int32_t* index, int ncls, int col, const int32_t* mos, const float* conf,
                                     int64_t n, int m, int32_t* stats, void* stream) {
    if (m <= 0) return INSMOS_OK;
    if (!index || !mos || !stats || col < 0 || col >= ncls) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(stats, 0, (size_t)m * 3 * sizeof(int32_t), s));
    if (n > 0) INSMOS_LAUNCH(k_inst_stats, dim3(cdiv(n, 256)), dim3(256), 0, s, index, ncls, col, mos, conf, n, m, stats);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_instance_relabel(const int32_t* index, int ncls, int col, const int32_t* decision, int64_t n, int m,
                                       int32_t* mos, void* stream) {
    if (m <= 0 || n <= 0) return INSMOS_OK;
    if (!index || !decision || !mos || col < 0 || col >= ncls) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    INSMOS_LAUNCH(k_inst_relabel, dim3(cdiv(n, 256)), dim3(256), 0, s, index, ncls, col, decision, n, m, mos);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
