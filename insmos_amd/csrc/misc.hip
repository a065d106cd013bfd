// insmos_amd/csrc/misc.hip -- row gathers, constant fills and the confusion-matrix histogram.
#include "common.h"

namespace insmos {

__global__ void k_gather_rows(const float* __restrict__ src, int ld_src, int c, const int64_t* __restrict__ idx,
                              int64_t n, float* __restrict__ out, int ld_out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    int64_t i = t / c;
    int ch = (int)(t % c);
    int64_t s = idx[i];
    out[i * ld_out + ch] = s >= 0 ? src[s * ld_src + ch] : 0.f;
}

__global__ void k_current_points(const float* __restrict__ pts, int ld_pts, const float* __restrict__ motion,
                                 int ld_motion, const int32_t* __restrict__ inverse,
                                 const int32_t* __restrict__ cur_index, int64_t n_cur, float* __restrict__ cur,
                                 int ld_cur) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cur) return;
    int p = cur_index[j];
    const float* pp = pts + (int64_t)p * ld_pts;
    const float* mm = motion + (int64_t)inverse[p] * ld_motion;
    float* o = cur + j * ld_cur;
    o[0] = pp[0]; o[1] = pp[1]; o[2] = pp[2]; o[3] = pp[3];
    o[4] = mm[0]; o[5] = mm[1]; o[6] = mm[2];
    for (int c = 7; c < ld_cur; ++c) o[c] = 0.f;
}

// the same for the B windows of a batch: cur_index holds batch-wide (window-major) point indices
// PART 0: the whole row; 1: the point's own columns [x, y, z, r] (and zeros behind them) -- known before MotionNet has run;
// 2: the motion columns only
template <int PART>
__global__ void k_current_points_w(WinPts W, int ld_pts, const float* __restrict__ motion, int ld_motion,
                                   const int32_t* __restrict__ inverse, const int32_t* __restrict__ cur_index,
                                   int64_t n_cur, float* __restrict__ cur, int ld_cur) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cur) return;
    int p = cur_index[j];
    float* o = cur + j * ld_cur;
    if (PART != 2) {
        int b;
        const float* pp = win_point(W, (int64_t)p, ld_pts, b);
        o[0] = pp[0]; o[1] = pp[1]; o[2] = pp[2]; o[3] = pp[3];
        for (int c = PART == 1 ? 4 : 7; c < ld_cur; ++c) o[c] = 0.f;
    }
    if (PART != 1) {
        const float* mm = motion + (int64_t)inverse[p] * ld_motion;
        o[4] = mm[0]; o[5] = mm[1]; o[6] = mm[2];
    }
}

__global__ void k_fill_cols(float* __restrict__ dst, int64_t n, int ld, int c0, int c, float v) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    dst[(t / c) * ld + c0 + (t % c)] = v;
}

__global__ void __launch_bounds__(256) k_confusion(const float* __restrict__ logits, int ld, const int64_t* __restrict__ gt,
                                                   int64_t n, int ncls, unsigned ignore_mask,
                                                   unsigned long long* __restrict__ cm) {
    __shared__ unsigned int h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* l = logits + i * ld;
        int best = 0;
        float bv = -INFINITY;
        bool have = false;
        for (int c = 0; c < ncls; ++c) {
            float v = ((ignore_mask >> c) & 1u) ? -INFINITY : l[c];
            if (!have || v > bv) { bv = v; best = c; have = true; }  // first max wins (torch.argmax)
        }
        int g = (int)gt[i];
        if (g >= 0 && g < ncls) atomicAdd(&h[best * ncls + g], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < ncls * ncls && h[threadIdx.x]) atomicAdd(&cm[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

struct Mat34d { double m[12]; };

// pose alignment + timestamp + concat of one scan (scripts/predict_mos.py:131-166): xyz' = (T @ [x,y,z,1]) in float64,
// rounded to float32 on store; intensity copied; t appended.
__global__ void k_stack_scan(const float* __restrict__ scan, int64_t n, Mat34d T, float t, float* __restrict__ out, int ld) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = *(const float4*)(scan + i * 4);
    const double x = p.x, y = p.y, z = p.z;
    float* o = out + i * ld;
    o[0] = (float)(((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3]);
    o[1] = (float)(((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7]);
    o[2] = (float)(((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11]);
    o[3] = p.w;
    o[4] = t;
}

// output stage of the driver (scripts/predict_mos.py:440-453): ignored classes -> -inf, softmax, confidence =
// softmax[:, 1:], argmax (first max), learning_map_inv lookup -> int32 labels
__global__ void k_output_stage(const float* __restrict__ logits, int ld, int64_t n, int ncls, unsigned ignore_mask,
                               const int32_t* __restrict__ lut, int32_t* __restrict__ labels, float* __restrict__ conf) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* l = logits + i * ld;
    float v[8], m = -INFINITY;
    for (int c = 0; c < ncls; ++c) {
        v[c] = ((ignore_mask >> c) & 1u) ? -INFINITY : l[c];
        m = fmaxf(m, v[c]);
    }
    float s = 0.f;
    for (int c = 0; c < ncls; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
    int best = 0;
    float bv = -1.f;
    for (int c = 0; c < ncls; ++c) {
        const float pr = v[c] / s;
        if (pr > bv) { bv = pr; best = c; }
        if (c >= 1) conf[i * (ncls - 1) + c - 1] = pr;
    }
    labels[i] = lut[best];
}


__global__ void k_copy_cols(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int64_t n, int c) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    const int64_t i = t / c;
    const int k = (int)(t % c);
    dst[i * ld_dst + k] = src[i * ld_src + k];
}

}  // namespace insmos

using namespace insmos;

extern "C" int insmos_gather_rows(const float* src, int ld_src, int c, const int64_t* idx, int64_t n, float* out,
                                  int ld_out, void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!src || !idx || !out || c <= 0) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_GATHER_ROWS, s);
    INSMOS_LAUNCH(k_gather_rows, dim3(cdiv(n * c, 256)), dim3(256), 0, s, src, ld_src, c, idx, n, out, ld_out);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_build_current_points(const float* points, int ld_pts, const float* motion, int ld_motion,
                                           const int32_t* inverse, const int32_t* cur_index, int64_t n_cur, float* cur,
                                           int ld_cur, void* stream) {
    if (n_cur <= 0) return INSMOS_OK;
    if (!points || !motion || !inverse || !cur_index || !cur || ld_cur < 7 || ld_pts < 5 || ld_motion < 3)
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_CUR_POINTS, s);
    INSMOS_LAUNCH(k_current_points, dim3(cdiv(n_cur, 256)), dim3(256), 0, s, points, ld_pts, motion, ld_motion,
                       inverse, cur_index, n_cur, cur, ld_cur);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// part 0: whole rows; part 1: [x, y, z, r, 0, ...] (motion may be NULL) -- all the voxeliser's coordinate phase reads; part 2: the
// motion columns 4..6 into rows part 1 wrote.  part 1 + part 2 == part 0.
extern "C" int insmos_build_current_points_part(const float* const* pts_host, const int64_t* n_pts_host, int B, int ld_pts,
                                                const float* motion, int ld_motion, const int32_t* inverse,
                                                const int32_t* cur_index, int64_t n_cur, float* cur, int ld_cur, int part,
                                                void* stream) {
    if (n_cur <= 0) return INSMOS_OK;
    WinPts W;
    int64_t n = 0;
    int rc = make_win_pts(pts_host, n_pts_host, B, &W, &n);
    if (rc) return rc;
    if (part < 0 || part > 2 || (part != 1 && (!motion || !inverse || ld_motion < 3)) || !cur_index || !cur || ld_cur < 7 || ld_pts < 5)
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_CUR_POINTS, s);
    const dim3 grid(cdiv(n_cur, 256)), block(256);
    if (part == 0) INSMOS_LAUNCH(k_current_points_w<0>, grid, block, 0, s, W, ld_pts, motion, ld_motion, inverse, cur_index, n_cur, cur, ld_cur);
    else if (part == 1) INSMOS_LAUNCH(k_current_points_w<1>, grid, block, 0, s, W, ld_pts, motion, ld_motion, inverse, cur_index, n_cur, cur, ld_cur);
    else INSMOS_LAUNCH(k_current_points_w<2>, grid, block, 0, s, W, ld_pts, motion, ld_motion, inverse, cur_index, n_cur, cur, ld_cur);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_build_current_points_windows(const float* const* pts_host, const int64_t* n_pts_host, int B, int ld_pts,
                                                   const float* motion, int ld_motion, const int32_t* inverse,
                                                   const int32_t* cur_index, int64_t n_cur, float* cur, int ld_cur,
                                                   void* stream) {
    return insmos_build_current_points_part(pts_host, n_pts_host, B, ld_pts, motion, ld_motion, inverse, cur_index, n_cur, cur, ld_cur, 0,
                                            stream);
}

extern "C" int insmos_fill_cols(float* dst, int64_t n, int ld, int c0, int c, float value, void* stream) {
    if (n <= 0 || c <= 0) return INSMOS_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_FILL, s);
    INSMOS_LAUNCH(k_fill_cols, dim3(cdiv(n * c, 256)), dim3(256), 0, s, dst, n, ld, c0, c, value);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_confusion3(const float* logits, int ld, const int64_t* gt, int64_t n, int ncls,
                                 unsigned ignore_mask, int64_t* cm, void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!logits || !gt || !cm || ncls <= 0 || ncls > 8) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_CONFUSION, s);
    unsigned g = cdiv(n, 256);
    if (g > 1024) g = 1024;
    INSMOS_LAUNCH(k_confusion, dim3(g), dim3(256), 0, s, logits, ld, gt, n, ncls, ignore_mask,
                       (unsigned long long*)cm);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_stack_scan(const float* scan, int64_t n, const double* T_host, float t, float* out, int ld_out,
                                 void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!scan || !T_host || !out || ld_out < 5 || ((uintptr_t)scan & 15)) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Mat34d T;
    for (int i = 0; i < 12; ++i) T.m[i] = T_host[i];
    ProfScope ps(KK_CUR_POINTS, s);
    INSMOS_LAUNCH(k_stack_scan, dim3(cdiv(n, 256)), dim3(256), 0, s, scan, n, T, t, out, ld_out);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_output_stage(const float* logits, int ld, int64_t n, int ncls, unsigned ignore_mask,
                                   const int32_t* lut, int32_t* labels, float* confidence, void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!logits || !lut || !labels || !confidence || ncls < 2 || ncls > 8) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_GATHER_ROWS, s);
    INSMOS_LAUNCH(k_output_stage, dim3(cdiv(n, 256)), dim3(256), 0, s, logits, ld, n, ncls, ignore_mask, lut, labels,
                       confidence);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t n, int c, void* stream) {
    if (n <= 0 || c <= 0) return INSMOS_OK;
    if (!src || !dst || ld_src < c || ld_dst < c) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_FILL, s);
    INSMOS_LAUNCH(k_copy_cols, dim3(cdiv(n * c, 256)), dim3(256), 0, s, src, ld_src, dst, ld_dst, n, c);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
