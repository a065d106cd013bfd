// insmos_amd/csrc/spconv.hip -- output-stationary sparse convolution on the CDNA4 matrix cores.
//
//   out[o, :] = epilogue( sum_k  in[nbr[k][o], :] @ W[k]  + bias )
//
// One wave owns 64 consecutive output rows x (16*COT) output channels.  For every tap k the wave
// reads its 64 neighbour indices (coalesced), skips the tap when none of its rows has that
// neighbour (wave-uniform branch; outputs are Morton-/raster-ordered so taps are spatially
// correlated), gathers the input rows straight into MFMA B-fragments (each lane one 16-byte
// load per 16-channel chunk -- four 16-lane groups cover one 64-byte sector of a row) and streams the
// pre-packed weight A-fragments (one coalesced 16-byte load per lane, L1/L2 resident).  The
// contraction runs on v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fmaf chain), i = output
// channel, j = output row, so the accumulator of lane (g, j) holds 4 consecutive channels of row j
// and the epilogue (folded-BN bias, ReLU, residual / channel-pair residual) stores one float4 per
// tile.  No atomics, no scatter: results are deterministic.
//
// Fragment maps used (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[i = 4*(lane>>4) + reg][j = lane&15].  The contraction index of one
// MFMA step s inside a 16-channel chunk is channel c0 + 4*(lane>>4) + s (any assignment is legal as
// long as A and B agree), which is what makes the gather a contiguous float4 per lane.
#include <cstring>
#include "common.h"

namespace insmos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvP {
    const float* in;
    const int32_t* nbr;
    const float* w;
    const float* bias;
    float* out;
    const float* res;
    int64_t n_out;
    int ld_in, cin, K, ld_out, cout, ld_res, res_mode, relu_pre, relu_post;
    int n16, has8, has4, nblk, ntile_co, n_otiles, vec_store;
};

template <int COT>
__global__ void __launch_bounds__(256) k_sparse_conv(ConvP P) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int n_cg = P.ntile_co / COT;
    if (gw >= (int64_t)P.n_otiles * n_cg) return;  // wave-uniform
    const int cg = (int)(gw / P.n_otiles);
    const int64_t ot = gw % P.n_otiles;
    const int g = lane >> 4, j = lane & 15;

    int64_t orow[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) orow[jt] = ot * 64 + jt * 16 + j;

    f32x4 acc[COT][4];
#pragma unroll
    for (int it = 0; it < COT; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t tap_stride = (int64_t)P.nblk * P.ntile_co * 256;
    for (int k = 0; k < P.K; ++k) {
        int idx[4];
        int anyv = 0;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            int v = -1;
            if (orow[jt] < P.n_out) v = P.nbr ? P.nbr[(int64_t)k * P.n_out + orow[jt]] : (int)orow[jt];
            idx[jt] = v;
            anyv |= (v >= 0);
        }
        if (!__any(anyv)) continue;
        const float* wk = P.w + (int64_t)k * tap_stride + ((int64_t)cg * COT) * 256 + lane * 4;
        int blk = 0;
        for (int c = 0; c < P.n16; ++c, ++blk) {
            f32x4 b[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                b[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx[jt] >= 0) b[jt] = *(const f32x4*)(P.in + (int64_t)idx[jt] * P.ld_in + c * 16 + 4 * g);
            }
            f32x4 a[COT];
            const float* wb = wk + (int64_t)blk * P.ntile_co * 256;
#pragma unroll
            for (int it = 0; it < COT; ++it) a[it] = *(const f32x4*)(wb + it * 256);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int it = 0; it < COT; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it][s], b[jt][s], acc[it][jt], 0, 0, 0);
        }
        int c0 = P.n16 * 16;
        if (P.has8) {
            f32x2 b[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                b[jt] = (f32x2){0.f, 0.f};
                if (idx[jt] >= 0) b[jt] = *(const f32x2*)(P.in + (int64_t)idx[jt] * P.ld_in + c0 + 2 * g);
            }
            f32x4 a[COT];
            const float* wb = wk + (int64_t)blk * P.ntile_co * 256;
#pragma unroll
            for (int it = 0; it < COT; ++it) a[it] = *(const f32x4*)(wb + it * 256);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int it = 0; it < COT; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it][s], b[jt][s], acc[it][jt], 0, 0, 0);
            ++blk;
            c0 += 8;
        }
        if (P.has4) {
            float b[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                b[jt] = 0.f;
                if (idx[jt] >= 0) b[jt] = P.in[(int64_t)idx[jt] * P.ld_in + c0 + g];
            }
            f32x4 a[COT];
            const float* wb = wk + (int64_t)blk * P.ntile_co * 256;
#pragma unroll
            for (int it = 0; it < COT; ++it) a[it] = *(const f32x4*)(wb + it * 256);
#pragma unroll
            for (int it = 0; it < COT; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it][0], b[jt], acc[it][jt], 0, 0, 0);
        }
    }

    // ---- epilogue: lane (g, j) holds channels co0..co0+3 of row orow[jt]
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const int co0 = (cg * COT + it) * 16 + 4 * g;
        const f32x4 bz = *(const f32x4*)(P.bias + co0);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int64_t o = orow[jt];
            if (o >= P.n_out || co0 >= P.cout) continue;
            f32x4 v = acc[it][jt] + bz;
            if (P.relu_pre) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (P.res_mode == 1) {
                if (P.vec_store && co0 + 3 < P.cout) {
                    v += *(const f32x4*)(P.res + o * P.ld_res + co0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co0 + r < P.cout) v[r] += P.res[o * P.ld_res + co0 + r];
                }
            } else if (P.res_mode == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < P.cout) {
                        const float* rp = P.res + o * P.ld_res + 2 * (co0 + r);
                        v[r] += rp[0] + rp[1];
                    }
            }
            if (P.relu_post) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            float* op = P.out + o * P.ld_out + co0;
            if (P.vec_store && co0 + 3 < P.cout) {
                *(f32x4*)op = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < P.cout) op[r] = v[r];
            }
        }
    }
}

__global__ void k_dense_nbr2d(int H, int W, int32_t* __restrict__ nbr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = (int64_t)H * W;
    if (t >= n * 9) return;
    int k = (int)(t / n);
    int site = (int)(t % n);
    int y = site / W + k / 3 - 1, x = site % W + k % 3 - 1;
    nbr[t] = (y >= 0 && y < H && x >= 0 && x < W) ? y * W + x : -1;
}

__global__ void k_sparse_to_bev(const float* __restrict__ feat, int ld_feat, int C, const int32_t* __restrict__ coords,
                                int64_t n, int D, int H, int W, float* __restrict__ bev) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int64_t i = t / C;
    int c = (int)(t % C);
    int4 q = *(const int4*)(coords + i * 4);  // [b, d, y, x]
    bev[((int64_t)q.z * W + q.w) * ((int64_t)C * D) + (int64_t)c * D + q.y] = feat[i * ld_feat + c];
}

}  // namespace insmos

using namespace insmos;

static void chunking(int cin, int& n16, int& has8, int& has4) {
    n16 = cin / 16;
    int rem = cin % 16;
    has8 = (rem & 8) ? 1 : 0;
    has4 = (rem & 4) ? 1 : 0;
}

extern "C" size_t insmos_packed_weight_floats(int K, int cin, int cout) {
    int n16, h8, h4;
    chunking(cin, n16, h8, h4);
    int ntile = (cout + 15) / 16;
    return (size_t)K * (size_t)(n16 + h8 + h4) * (size_t)ntile * 256;
}

extern "C" int insmos_pack_weights_host(const float* taps, int K, int cin_real, int cout_real, int cin, int cout,
                                        float* packed) {
    if (!taps || !packed || K <= 0 || cin % 4 != 0 || cin < cin_real || cout < cout_real) return INSMOS_EINVAL;
    int n16, h8, h4;
    chunking(cin, n16, h8, h4);
    const int nblk = n16 + h8 + h4, ntile = (cout + 15) / 16;
    auto W = [&](int k, int ci, int co) -> float {
        return (ci < cin_real && co < cout_real) ? taps[((size_t)k * cin_real + ci) * cout_real + co] : 0.f;
    };
    for (int k = 0; k < K; ++k) {
        int blk = 0;
        auto emit = [&](int c0, int width) {  // width = channels per lane group: 4, 2 or 1
            for (int t = 0; t < ntile; ++t)
                for (int l = 0; l < 64; ++l) {
                    int g = l >> 4, i = l & 15;
                    float* dst = packed + ((((size_t)k * nblk + blk) * ntile + t) * 64 + l) * 4;
                    for (int s = 0; s < 4; ++s) dst[s] = (s < width) ? W(k, c0 + width * g + s, t * 16 + i) : 0.f;
                }
            ++blk;
        };
        for (int c = 0; c < n16; ++c) emit(c * 16, 4);
        int c0 = n16 * 16;
        if (h8) { emit(c0, 2); c0 += 8; }
        if (h4) emit(c0, 1);
    }
    return INSMOS_OK;
}

extern "C" int insmos_sparse_conv(const float* in, int ld_in, int cin, const int32_t* nbr, int K, int64_t n_out,
                                  const float* wpacked, const float* bias, float* out, int ld_out, int cout,
                                  const float* res, int ld_res, int res_mode, int relu_pre, int relu_post,
                                  void* stream) {
    if (n_out <= 0) return INSMOS_OK;
    if (!in || !wpacked || !bias || !out || cin <= 0 || cin % 4 != 0 || ld_in % 4 != 0 || ld_in < cin || K <= 0 ||
        (!nbr && K != 1) || cout <= 0 || ld_out < cout || (res_mode != 0 && !res) || ((uintptr_t)in & 15))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ConvP P;
    P.in = in; P.nbr = nbr; P.w = wpacked; P.bias = bias; P.out = out; P.res = res; P.n_out = n_out;
    P.ld_in = ld_in; P.cin = cin; P.K = K; P.ld_out = ld_out; P.cout = cout; P.ld_res = ld_res; P.res_mode = res_mode;
    P.relu_pre = relu_pre; P.relu_post = relu_post;
    chunking(cin, P.n16, P.has8, P.has4);
    P.nblk = P.n16 + P.has8 + P.has4;
    P.ntile_co = (cout + 15) / 16;
    P.n_otiles = (int)((n_out + 63) / 64);
    P.vec_store = (cout % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
                   (res_mode != 1 || (ld_res % 4 == 0 && ((uintptr_t)res & 15) == 0)))
                      ? 1
                      : 0;
    // channel tiles per wave: as many as possible while the grid still fills the 1024 SIMDs
    int cot = 1;
    for (int c = 8; c >= 2; c >>= 1)
        if (P.ntile_co % c == 0 && (int64_t)P.n_otiles * (P.ntile_co / c) >= 1024) { cot = c; break; }
    int64_t waves = (int64_t)P.n_otiles * (P.ntile_co / cot);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    ProfScope ps(KK_SPARSE_CONV, s);
    switch (cot) {
        case 8: hipLaunchKernelGGL(k_sparse_conv<8>, grid, block, 0, s, P); break;
        case 4: hipLaunchKernelGGL(k_sparse_conv<4>, grid, block, 0, s, P); break;
        case 2: hipLaunchKernelGGL(k_sparse_conv<2>, grid, block, 0, s, P); break;
        default: hipLaunchKernelGGL(k_sparse_conv<1>, grid, block, 0, s, P); break;
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_dense_nbr2d(int H, int W, int32_t* nbr, void* stream) {
    if (H <= 0 || W <= 0 || !nbr) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_DENSE_NBR, s);
    int64_t n = (int64_t)H * W * 9;
    hipLaunchKernelGGL(k_dense_nbr2d, dim3(cdiv(n, 256)), dim3(256), 0, s, H, W, nbr);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_sparse_to_bev(const float* feat, int ld_feat, int C, const int32_t* coords, int64_t n, int D,
                                    int H, int W, float* bev, void* stream) {
    if (!feat || !coords || !bev || C <= 0 || D <= 0) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps(KK_MEMSET, s);
        HIP_TRY(hipMemsetAsync(bev, 0, (size_t)H * W * C * D * sizeof(float), s));
    }
    if (n > 0) {
        ProfScope ps(KK_TO_BEV, s);
        hipLaunchKernelGGL(k_sparse_to_bev, dim3(cdiv(n * C, 256)), dim3(256), 0, s, feat, ld_feat, C, coords, n, D, H, W,
                           bev);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
