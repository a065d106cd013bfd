// insmos_amd/csrc/spconv.hip -- output-stationary sparse convolution on the CDNA4 matrix cores.
//
//   out[o, :] = epilogue( sum_k  in[nbr[k][o], :] @ W[k]  + bias )
//
// One wave owns 16*JT consecutive output rows x 16*COT output channels (tile shape picked per layer
// so that the launch keeps >= 2 waves per SIMD whenever the layer is big enough).  Taps are walked
// through a per-16-row-group ACTIVE-TAP BITMASK produced when the neighbour table is built
// (insmos_build_nbr): a tap none of the tile's rows uses is never touched, and inside a tile each
// 16-row group skips its own empty taps (rows are Morton-/raster-ordered, so occupancy is spatially
// coherent: ~50 % of (64-row tile, tap) slots and ~60 % of (16-row group, tap) slots are empty on
// LiDAR data).  Because the active-tap list is known up front the loop is software-pipelined:
// neighbour indices are fetched two taps ahead, the gathered rows (B fragments: one 16-byte load
// per lane per 16-channel chunk; four 16-lane groups cover one 64-byte sector of a row) and the
// pre-packed weight A fragments (one coalesced 16-byte load per lane, L1/L2 resident) one step
// ahead of the MFMAs that consume them.  All offsets are 32-bit so loads use SGPR-base + VGPR-offset
// addressing.  The contraction runs on v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fmaf chain),
// i = output channel, j = output row, so lane (g, j) ends up with 4 consecutive channels of row j and
// the epilogue (folded-BN bias, ReLU, residual / channel-pair residual) is one float4 store per tile.
// No atomics, no scatter: results are deterministic.
//
// Fragment maps (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[i = 4*(lane>>4) + reg][j = lane&15].  The contraction index of MFMA
// step s inside a 16-channel chunk is channel c0 + 4*(lane>>4) + s (any assignment is legal as long
// as A and B agree), which is what makes the gather a contiguous float4 per lane.
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace insmos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvP {
    const float* in;
    const int32_t* nbr;
    const uint32_t* mask16;  // [ceil(n_out/16)][4] active-tap bits per 16-row group, or null (all active)
    const float* w;
    const float* bias;
    float* out;
    const float* res;
    uint32_t n_out;
    uint32_t in_bytes;  // extent of the `in` view: (n_in - 1) * ld_in * 4 + cin * 4
    int ld_in, cin, K, ld_out, cout, ld_res, res_mode, relu_pre, relu_post;
    int n16, has8, has4, nblk, ntile_co, n_otiles, vec_store, xcd_remap;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// next active tap of the 128-bit set, or `keep` when the set is exhausted (branch-free scalar code)
__device__ __forceinline__ int pop_or_keep(uint64_t& lo, uint64_t& hi, int keep) {
    const bool use_lo = lo != 0;
    const uint64_t w = use_lo ? lo : hi;
    const int k = (w ? __builtin_ctzll(w) : 0) + (use_lo ? 0 : 64);
    const uint64_t cleared = w & (w - 1);
    const bool any = w != 0;
    lo = use_lo ? cleared : lo;
    hi = use_lo ? hi : cleared;
    return any ? k : keep;
}

// CK == 0: Cin is a multiple of 16, a tap is Cin/16 chunks of 16 channels (4 MFMA steps each)
// CK == 4 / 8: the whole contraction of a tap is ONE chunk of that width (small-C 4D layers)
// IDENT: no neighbour table -- a 1x1 convolution / Linear (row o reads row o)
// R: depth of the operand register ring; operands of item i+R-1 are requested before item i's MFMAs
//
// The main loop is a COUNTED, branch-free software pipeline over work items (tap, chunk).  A LOAD
// CURSOR walks R-1 items ahead of the MFMAs: it owns the tap ring (current tap's row offsets, next
// tap's raw indices, and the indices of the tap after that already in flight), so neighbour indices
// are requested >= one full tap before they are needed.  All gathers are BUFFER loads through an SGPR
// descriptor with 32-bit offsets: a missing neighbour (index -1) wraps to an out-of-range offset and
// the hardware returns 0 -- no clamp, no select, no exec-masked branch (those made hipcc wait
// vmcnt(0) inside each branch and shuttle the accumulators between AGPRs and VGPRs).  Running off
// the end of the tap list re-requests the last tap (clamp instead of guard); rows past n_out compute
// on re-read data and are masked at the store.
// SPLIT > 1 (generic layers with >= 64 output channels): the tile covers ALL its channel tiles in one wave
// and the SPLIT consecutive waves of a block share the tile's row group but take every SPLIT-th active
// tap, so each input row chunk is gathered exactly once per tile (instead of once per channel group);
// the partial accumulators are summed through LDS in fixed wave order (deterministic) in the epilogue.
template <int COT, int JT, int CK, bool IDENT, int R, int SPLIT>
__global__ void __launch_bounds__(256) k_sparse_conv(ConvP P) {
    const int lane = threadIdx.x & 63;
    // readfirstlane: tell the compiler the wave id (hence every tile-level quantity) is wave-uniform
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware remap (bijective for any grid size): workgroup b runs on XCD b % 8, so give each XCD one
    // CONTIGUOUS range of tiles -- rows are Morton-/raster-ordered, a contiguous range is spatially compact
    // and its gathered rows (own range + halo) stay in that XCD's private 4 MiB L2.  Speed only, never correctness.
    uint32_t bid = blockIdx.x;
    if (P.xcd_remap) {
        const uint32_t nb = gridDim.x, xcd = bid & 7u, q = nb >> 3, r = nb & 7u;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const uint32_t gw = bid * 4 + wib;
    const uint32_t n_cg = P.ntile_co / COT;
    const uint32_t n_tiles = (uint32_t)P.n_otiles * n_cg;
    const uint32_t tile_raw = gw / SPLIT, ws = gw % SPLIT;
    const bool live = tile_raw < n_tiles;
    if (SPLIT == 1 && !live) return;  // wave-uniform; split kernels keep dead waves alive for the block barrier
    const uint32_t tile = live ? tile_raw : n_tiles - 1;
    const uint32_t cg = tile / P.n_otiles;
    const uint32_t ot = tile % P.n_otiles;
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;  // row pitch in bytes

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb =
        __builtin_amdgcn_make_buffer_rsrc((void*)(IDENT ? (const void*)P.in : (const void*)P.nbr), 0,
                                          IDENT ? 4 : (int)((uint32_t)P.K * n_out * 4u), 0x00020000);

    uint32_t orow[JT], rowoff[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        orow[jt] = ot * (16 * JT) + jt * 16 + j;
        rowoff[jt] = (orow[jt] < n_out ? orow[jt] : n_out - 1) * 4u;  // byte offset inside one tap row of nbr
    }
    constexpr uint32_t LW = CK == 4 ? 4u : CK == 8 ? 8u : 16u;  // bytes per lane per gather
    const uint32_t goff = (uint32_t)g * LW;

    // ---- active-tap set of the tile = union over its 16-row groups (SGPRs)
    uint64_t tlo = 0, thi = 0;
    if constexpr (IDENT) {
        tlo = 1;
    } else {
        const int K = P.K;
        const uint32_t ngrp = (n_out + 15) >> 4;
        if (P.mask16) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const uint32_t grp = ot * JT + jt;
                const uint32_t* mp = P.mask16 + (size_t)(grp < ngrp ? grp : ngrp - 1) * 4;
                uint32_t w0 = __builtin_amdgcn_readfirstlane(mp[0]), w1 = __builtin_amdgcn_readfirstlane(mp[1]);
                uint32_t w2 = __builtin_amdgcn_readfirstlane(mp[2]), w3 = __builtin_amdgcn_readfirstlane(mp[3]);
                if (grp < ngrp) {
                    tlo |= ((uint64_t)w1 << 32) | w0;
                    thi |= ((uint64_t)w3 << 32) | w2;
                }
            }
        } else {
            tlo = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
            thi = K > 64 ? (K >= 128 ? ~0ull : ((1ull << (K - 64)) - 1ull)) : 0ull;
        }
    }
    int nt = __builtin_popcountll(tlo) + __builtin_popcountll(thi);
    if constexpr (SPLIT > 1) {
        for (uint32_t q = 0; q < ws; ++q) (void)pop_or_keep(tlo, thi, 0);  // my taps: ranks ws, ws+SPLIT, ...
        nt = (live && nt > (int)ws) ? (nt - (int)ws + SPLIT - 1) / SPLIT : 0;
    }
    // next tap of THIS wave (skips the taps owned by the other waves of the tile)
    auto next_tap = [&](int keep) {
        const int k = pop_or_keep(tlo, thi, keep);
        if constexpr (SPLIT > 1) {
#pragma unroll
            for (int q = 1; q < SPLIT; ++q) (void)pop_or_keep(tlo, thi, 0);
        }
        return k;
    };

    f32x4 acc[COT][JT];
#pragma unroll
    for (int it = 0; it < COT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) acc[it][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t blk_stride = (uint32_t)P.ntile_co * 256u;   // floats between chunk blocks of one tap
    const uint32_t tap_stride = (uint32_t)P.nblk * blk_stride;  // floats between taps
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    const uint32_t woff = ((cg * COT) * 256u + lane * 4u) * 4u;
    constexpr int NS = CK ? CK / 4 : 4;  // MFMA steps per chunk
    const int nchunk = CK ? 1 : P.n16;

    // raw neighbour indices of tap k (one dword per 16-row group and lane)
    auto load_idx = [&](int k, uint32_t (&idx)[JT]) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if constexpr (IDENT) idx[jt] = rowoff[jt] >> 2;
            else idx[jt] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, rowoff[jt], (uint32_t)k * n_out * 4u, 0);
        }
    };
    // byte offsets of the gathered rows; index -1 wraps past the end of the buffer -> loads return 0
    auto row_offsets = [&](const uint32_t (&idx)[JT], uint32_t (&off)[JT]) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) off[jt] = idx[jt] * ld4 + goff;
    };
    auto load_ab = [&](int k, int c, const uint32_t (&off)[JT], f32x4 (&a)[COT], f32x4 (&b)[JT]) {
        const uint32_t so = (uint32_t)c * 64u;  // chunk byte offset inside a row (scalar)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if constexpr (CK == 4) {
                b[jt] = (f32x4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off[jt], so, 0)), 0.f,
                                0.f, 0.f};
            } else if constexpr (CK == 8) {
                f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, off[jt], so, 0));
                b[jt] = (f32x4){t[0], t[1], 0.f, 0.f};
            } else {
                b[jt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off[jt], so, 0));
            }
        }
        // weight fragments through a buffer descriptor too: one load KIND in the loop keeps hipcc's
        // vmcnt accounting exact (mixing global_ and buffer_ loads made it drain to vmcnt(0))
        const uint32_t sw = ((uint32_t)k * tap_stride + (uint32_t)c * blk_stride) * 4u;
#pragma unroll
        for (int it = 0; it < COT; ++it)
            a[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff + (uint32_t)it * 1024u, sw, 0));
    };
    auto mma = [&](const f32x4 (&a)[COT], const f32x4 (&b)[JT]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int it = 0; it < COT; ++it) acc[it][jt] = MFMA(a[it][s], b[jt][s], acc[it][jt]);
    };

    if (nt > 0) {
        // ---- load cursor: (kL, cL) = next item to request; offL = its rows; tap ring kN/idxN, kNN/idxNN
        int kL = next_tap(0);
        int kN = next_tap(kL);
        int kNN = next_tap(kN);
        int cL = 0;
        uint32_t offL[JT], idxN[JT], idxNN[JT];
        {
            uint32_t idx0[JT];
            load_idx(kL, idx0);
            load_idx(kN, idxN);
            load_idx(kNN, idxNN);
            row_offsets(idx0, offL);
        }
        f32x4 bs[R][JT], as[R][COT];
        const int nitems = nt * nchunk;
#define REQUEST(slot)                                                                     \
    {                                                                                     \
        load_ab(kL, cL, offL, as[slot], bs[slot]);                                        \
        if (cL + 1 < nchunk) {                                                            \
            ++cL;                                                                         \
        } else {                                                                          \
            cL = 0;                                                                       \
            kL = kN; kN = kNN;                                                            \
            row_offsets(idxN, offL);                                                      \
            _Pragma("unroll") for (int jt = 0; jt < JT; ++jt) idxN[jt] = idxNN[jt];       \
            kNN = next_tap(kNN);                                                          \
            load_idx(kNN, idxNN);                                                         \
        }                                                                                 \
    }
#pragma unroll
        for (int r = 0; r < R - 1; ++r) REQUEST(r)
        int i = 0;
        for (; i + R <= nitems; i += R) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                REQUEST((r + R - 1) % R)
                mma(as[r], bs[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R - 1; ++r) {
            if (i + r < nitems) {  // wave-uniform tail, at most R-1 items (their operands are already in flight)
                mma(as[r], bs[r]);
            }
        }
#undef REQUEST
    }

    // ---- epilogue: lane (g, j) holds channels co0..co0+3 of row orow[jt]
    const uint32_t cout = P.cout;
    auto finish = [&](int it, int jt, f32x4 v) {
        const uint32_t co0 = (cg * COT + it) * 16 + 4 * g;
        const uint32_t o = orow[jt];
        if (o >= n_out || co0 >= cout) return;
        v += *(const f32x4*)(P.bias + co0);
        if (P.relu_pre) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (P.res_mode == 1) {
            const float* rp = P.res + (size_t)o * P.ld_res + co0;
            if (P.vec_store && co0 + 3 < cout) {
                v += *(const f32x4*)rp;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < cout) v[r] += rp[r];
            }
        } else if (P.res_mode == 2) {
            const float* rp = P.res + (size_t)o * P.ld_res + 2 * co0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co0 + r < cout) v[r] += rp[2 * r] + rp[2 * r + 1];
        }
        if (P.relu_post) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        float* op = P.out + (size_t)o * P.ld_out + co0;
        if (P.vec_store && co0 + 3 < cout) {
            *(f32x4*)op = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co0 + r < cout) op[r] = v[r];
        }
    };
    if constexpr (SPLIT == 1) {
#pragma unroll
        for (int it = 0; it < COT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) finish(it, jt, acc[it][jt]);
    } else {
        static_assert(SPLIT == 1 || (JT == 1 && COT % SPLIT == 0), "tap-split tiles are 16 rows x all channels");
        __shared__ f32x4 red[4][COT][64];  // [wave in block][channel tile][lane]
#pragma unroll
        for (int it = 0; it < COT; ++it) red[wib][it][lane] = acc[it][0];
        __syncthreads();
        if (live) {
            const uint32_t w0 = wib - ws;  // first wave of this tile inside the block
            constexpr int PER = COT / SPLIT;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int it = (int)ws * PER + q;
                f32x4 v = red[w0][it][lane];
#pragma unroll
                for (int p = 1; p < SPLIT; ++p) v += red[w0 + p][it][lane];  // fixed order -> deterministic
                finish(it, 0, v);
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

namespace {
typedef void (*ConvKernel)(ConvP);
struct Cfg { int cot, jt; };

// ring depth by tile area: small tiles need more items in flight to cover L2 latency
template <int COT, int JT> constexpr int ring_depth() { return COT * JT <= 2 ? 3 : 2; }

template <int CK, bool IDENT>
ConvKernel pick_kernel(int cot, int jt) {
#define CASE(C, J) if (cot == C && jt == J) return k_sparse_conv<C, J, CK, IDENT, ring_depth<C, J>(), 1>;
    if constexpr (CK == 0) {
        CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(4, 1) CASE(4, 2) CASE(4, 4)
        CASE(8, 1) CASE(8, 2) CASE(8, 4)
    } else {
        CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4)
    }
#undef CASE
    return nullptr;
}

// tuning hook (insmos_debug_conv_force): generic non-identity layers at an explicit (COT, JT, ring)
int g_force_cot = 0, g_force_jt = 0, g_force_ring = 0;
ConvKernel pick_forced(int cot, int jt, int ring) {
#define CASE(C, J, RR) if (cot == C && jt == J && ring == RR) return k_sparse_conv<C, J, 0, false, RR, 1>;
#define CASES(C, J) CASE(C, J, 2) CASE(C, J, 3) CASE(C, J, 4)
    CASES(1, 1) CASES(1, 2) CASES(1, 4) CASES(2, 1) CASES(2, 2) CASES(2, 4) CASES(4, 1) CASES(4, 2) CASES(4, 4)
    CASE(8, 1, 2) CASE(8, 1, 3) CASE(8, 2, 2) CASE(8, 2, 3) CASE(8, 4, 2)
#undef CASES
#undef CASE
    return nullptr;
}
}  // namespace

extern "C" int insmos_sparse_conv(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr,
                                  const uint32_t* mask16, int K, int64_t n_out, const float* wpacked, const float* bias,
                                  float* out, int ld_out, int cout, const float* res, int ld_res, int res_mode,
                                  int relu_pre, int relu_post, void* stream) {
    if (n_out <= 0) return INSMOS_OK;
    if (!in || n_in <= 0 || !wpacked || !bias || !out || cin <= 0 || ld_in % 4 != 0 || ld_in < cin || K <= 0 || K > 128 ||
        (!nbr && (K != 1 || n_in < n_out)) || cout <= 0 || ld_out < cout || (res_mode != 0 && !res) ||
        ((uintptr_t)in & 15) || n_out * (int64_t)K * 4 >= (1ll << 31) || n_in * (int64_t)ld_in * 4 >= (1ll << 31))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ConvP P;
    P.in = in; P.nbr = nbr; P.mask16 = mask16; P.w = wpacked; P.bias = bias; P.out = out; P.res = res;
    P.n_out = (uint32_t)n_out;
    P.in_bytes = (uint32_t)((n_in - 1) * (int64_t)ld_in * 4 + (int64_t)cin * 4);
    P.ld_in = ld_in; P.cin = cin; P.K = K; P.ld_out = ld_out; P.cout = cout; P.ld_res = ld_res; P.res_mode = res_mode;
    P.relu_pre = relu_pre; P.relu_post = relu_post;
    chunking(cin, P.n16, P.has8, P.has4);
    P.nblk = P.n16 + P.has8 + P.has4;
    P.ntile_co = (cout + 15) / 16;
    P.vec_store = (cout % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
                   (res_mode != 1 || (ld_res % 4 == 0 && ((uintptr_t)res & 15) == 0)))
                      ? 1
                      : 0;
    const int ck = (cin == 4 || cin == 8) ? cin : 0;
    if (!ck && (cin % 16 != 0)) return INSMOS_EINVAL;  // supported widths: 4, 8, or a multiple of 16
    // tile shape, from tools/conv_tune.py sweeps on MI355X: a 16-row gather is ~8x the cost of a coalesced
    // weight fragment, so generic layers always use 16-row tiles (JT = 1) and widen in channels instead:
    // 2 channel tiles per wave, 4 when the layer is large enough to still give >= 2 waves per SIMD (the
    // dense BEV convs).  Single-chunk small-C layers (Cin 4/8) use `ck_jt` row groups per wave.
    static int ck_jt = 0;
    if (!ck_jt) { const char* e = getenv("INSMOS_CK_JT"); ck_jt = e ? atoi(e) : 2; if (ck_jt != 1 && ck_jt != 4) ck_jt = 2; }
    Cfg best = {1, ck ? ck_jt : 1};
    {
        const long groups = (long)((n_out + 15) / 16);
        if (!ck) {
            if (P.ntile_co % 4 == 0 && groups * (P.ntile_co / 4) >= 2048) best.cot = 4;
            else if (P.ntile_co % 2 == 0) best.cot = 2;
        } else if (P.ntile_co % 2 == 0) {
            best.cot = 2;
        }
    }
    P.n_otiles = (int)((n_out + 16 * best.jt - 1) / (16 * best.jt));
    const bool ident = (nbr == nullptr);
    ConvKernel kern = nullptr;
    if (ck == 4) kern = ident ? pick_kernel<4, true>(best.cot, best.jt) : pick_kernel<4, false>(best.cot, best.jt);
    else if (ck == 8) kern = ident ? pick_kernel<8, true>(best.cot, best.jt) : pick_kernel<8, false>(best.cot, best.jt);
    else kern = ident ? pick_kernel<0, true>(best.cot, best.jt) : pick_kernel<0, false>(best.cot, best.jt);
    // tap-split tiles: all channel tiles in one wave, the block's waves share the row group and split the taps
    int split = 1;
    static int split_env = -1;
    if (split_env < 0) { const char* e = getenv("INSMOS_CONV_SPLIT"); split_env = e ? atoi(e) : 1; }
    // (measured: +20-30 % on the masked 27-tap C >= 64 layers; no gain on dense 9-tap BEV convs or 3-tap layers)
    if (split_env && !ck && !ident && mask16 && (P.ntile_co == 4 || P.ntile_co == 8) && K >= 16) {
        split = 4;
        best = {P.ntile_co, 1};
        P.n_otiles = (int)((n_out + 15) / 16);
        if (P.ntile_co == 8) kern = split == 4 ? k_sparse_conv<8, 1, 0, false, 2, 4> : k_sparse_conv<8, 1, 0, false, 2, 2>;
        else kern = split == 4 ? k_sparse_conv<4, 1, 0, false, 2, 4> : k_sparse_conv<4, 1, 0, false, 2, 2>;
    }
    if (g_force_cot && !ck && !ident && P.ntile_co % g_force_cot == 0) {
        split = 1;
        ConvKernel fk = pick_forced(g_force_cot, g_force_jt, g_force_ring);
        if (fk) {
            kern = fk;
            best = {g_force_cot, g_force_jt};
            P.n_otiles = (int)((n_out + 16 * best.jt - 1) / (16 * best.jt));
        }
    }
    if (!kern) return INSMOS_EINVAL;
    static int xcd_env = -1;
    if (xcd_env < 0) { const char* e = getenv("INSMOS_XCD_REMAP"); xcd_env = e ? atoi(e) : 0; }  // measured on S0: -4 % (weights are re-read per XCD, ranges imbalance) -> off by default
    P.xcd_remap = xcd_env;
    long waves = (long)P.n_otiles * (P.ntile_co / best.cot) * split;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    ProfScope ps(KK_SPARSE_CONV, s);
    hipLaunchKernelGGL(kern, grid, block, 0, s, P);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_debug_conv_force(int cot, int jt, int ring) {
    g_force_cot = cot; g_force_jt = jt; g_force_ring = ring;
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
