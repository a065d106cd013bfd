// insmos_amd/csrc/spconv_lds.hip -- LDS-staged gathers for the 81-tap (3 x 3 x 3 x 3) single-chunk layers of MotionNet
// (BasicBlock convolutions with Cin = 8 or 16: models/MinkowskiEngine/resnet.py:110-119, minkunet.py:52-137).
//
// Why.  These layers run at 12-25 TFLOP/s on the generic kernels (spconv.hip): a (16-row tile, tap) costs a vector-memory
// gather that touches 16 distinct cache lines to use 32-64 bytes of each, plus an index load and a weight fragment, for 2-4
// MFMAs -- the texture-address path is the bound, the matrix cores idle ~60 %.  But the gathers are LOCAL: rows are in
// (time, Morton) order, so the 27 spatial neighbours of 64 consecutive rows at one time offset sit, for ~88 % of the present
// (row, tap) pairs, inside ONE run of 256 consecutive input rows, and a 64-row block re-reads each distinct neighbour row ~4.5 x
// (tools/gather_locality.py, tools/gather_window_policy.py on the S0 window).
//
// What.  One wave owns 64 consecutive output rows (four 16-row tiles) x COT channel tiles.  Per time offset dt (a group of 27
// consecutive taps):
//   1. the block's neighbour indices (lane = row, one coalesced dword load per ACTIVE tap, all three groups up front);
//   2. a window of CAP = 256 input rows centred on the centre tap's neighbours is copied into LDS with coalesced 16-byte loads
//      (only the part of it that is referenced);
//   3. present neighbours outside the window (~12 %: the block's neighbourhood crosses a Morton boundary) are listed, and the
//      listed rows are fetched into an overflow area of OVF = 128 rows, one entry per lane, all loads independent;
//   4. every (row, tap) gets a byte offset into that LDS image (absent neighbours point at a zero row), and the contraction
//      reads its B fragments with ds_read_b64 / b128 -- no vector-memory gather in the loop.  A tap's weight fragment is fetched
//      once per 64 rows (prefetched one tap ahead).
// A (block, dt) with more than OVF misses (5-8 %) keeps raw indices in the table and gathers from memory like the generic kernel.
//
// Same bits as the generic kernels: per 16-row tile the taps are walked in ascending order through the tile's own active-tap
// mask, each step is the same v_mfma_f32_16x16x4_f32 sequence on the same operand values (tests/test_gpu_conv.py).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

namespace insmos {
namespace {

constexpr int LDS_CAP = 256;   // window rows
constexpr int LDS_OVF = 128;   // overflow rows

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
// the 27 taps [27 g, 27 g + 27) of a 128-bit tap set
__device__ __forceinline__ uint32_t group_bits(uint64_t lo, uint64_t hi, int g) {
    const uint64_t v = g == 0 ? lo : g == 1 ? (lo >> 27) : ((lo >> 54) | (hi << 10));
    return (uint32_t)v & 0x7FFFFFFu;
}

// CK = 8: Cin = 8 (32-byte rows, 2 MFMA steps per tap); CK = 0: Cin = 16 (64-byte rows, 4 steps)
template <int CK, int COT>
__global__ void __launch_bounds__(64, 2) k_conv_lds(ConvP P) {
    constexpr uint32_t LW = CK == 8 ? 8u : 16u;   // bytes per lane per B fragment
    constexpr uint32_t RB = 4u * LW;              // bytes per input row (rows are contiguous: ld_in == cin)
    constexpr uint32_t LWF = CK == 8 ? 2u : 4u;   // floats per lane per weight fragment
    constexpr uint32_t FR = 64u * LWF;
    constexpr int NS = CK == 8 ? 2 : 4;
    constexpr uint32_t OVF_OFF = LDS_CAP * RB, ZERO_OFF = (LDS_CAP + LDS_OVF) * RB;
    constexpr int NIT = (int)(LDS_CAP * RB / 1024u);   // 1 KiB slabs of the window: 8 (CK 8) / 16
    __shared__ uint32_t s_off[27 * 64];
    __shared__ uint32_t s_list[LDS_OVF];
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[(LDS_CAP + LDS_OVF + 1) * RB];

    const int lane = threadIdx.x;
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t n4 = n_out * 4u;
    const uint32_t nblk = (uint32_t)P.n_otiles;   // 64-row blocks
    const uint32_t cg = blockIdx.x / nblk, ob = blockIdx.x % nblk;
    const uint32_t r0 = P.row0 + ob * 64u;
    const uint32_t my_row = r0 + (uint32_t)lane;
    const bool row_ok = my_row < n_out;
    const uint32_t cout = P.cout;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb = __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n4), 0x00020000);
    const uint32_t tap_stride = (uint32_t)P.ntile_co * FR;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);

    // ---- the four tiles' active-tap sets (scalar) and their union
    uint64_t tlo[4], thi[4], ulo = 0, uhi = 0;
    {
        const uint32_t ngrp = (n_out + 15u) >> 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t grp = (r0 >> 4) + (uint32_t)t;
            const uint32_t* mp = P.mask16 + (size_t)(grp < ngrp ? grp : ngrp - 1) * 4;
            const uint32_t w0 = __builtin_amdgcn_readfirstlane(mp[0]), w1 = __builtin_amdgcn_readfirstlane(mp[1]);
            const uint32_t w2 = __builtin_amdgcn_readfirstlane(mp[2]), w3 = __builtin_amdgcn_readfirstlane(mp[3]);
            const bool okg = grp < ngrp;
            tlo[t] = okg ? (((uint64_t)w1 << 32) | w0) : 0ull;
            thi[t] = okg ? (((uint64_t)w3 << 32) | w2) : 0ull;
            ulo |= tlo[t];
            uhi |= thi[t];
        }
    }
    // ---- 1. neighbour indices of my row under every active tap of the block (one exposed memory latency for all three groups)
    int idx[3][27];
    {
        const uint32_t rowoff = (row_ok ? my_row : n_out - 1u) * 4u;
#pragma unroll
        for (int dg = 0; dg < 3; ++dg) {
            const uint32_t ug = group_bits(ulo, uhi, dg);
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                idx[dg][k] = -1;
                if ((ug >> k) & 1u)   // (wave-uniform)
                    idx[dg][k] = __builtin_amdgcn_raw_buffer_load_b32(rs_nb, rowoff, (uint32_t)(dg * 27 + k) * n4, 0);
            }
        }
    }
    // the zero row absent neighbours point at
    if (lane < (int)(RB / 4u)) *(uint32_t*)(s_rows + ZERO_OFF + 4u * (uint32_t)lane) = 0u;

    f32x4 acc[COT][4];
#pragma unroll
    for (int it = 0; it < COT; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[it][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t woffv[COT];  // lanes whose output channel lies beyond Cout read zeros from past the end of the buffer
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co = (cg * COT + it) * 16u + (uint32_t)j;
        woffv[it] = co < cout ? ((cg * COT + it) * FR + (uint32_t)lane * LWF) * 4u : 0x7FFFFFF0u;
    }
    auto load_w = [&](int ktap, f32x4 (&a)[COT]) {
        const uint32_t sw = (uint32_t)ktap * tap_stride * 4u;
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            if constexpr (CK == 8) {
                const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, woffv[it], sw, 0));
                a[it] = (f32x4){t2[0], t2[1], 0.f, 0.f};
            } else {
                a[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw, 0));
            }
        }
    };

#pragma unroll
    for (int dg = 0; dg < 3; ++dg) {
        const uint32_t ug = group_bits(ulo, uhi, dg);
        if (ug == 0u) continue;   // (wave-uniform)
        if (!row_ok) {
#pragma unroll
            for (int k = 0; k < 27; ++k) idx[dg][k] = -1;
        }
        // ---- 2. window placement: centred on the centre tap's neighbours (the voxels themselves at t + dt); a block none of
        // whose voxels exists there centres on its smallest neighbour index
        int lo;
        {
            const int c = idx[dg][13];
            int cmin = wave_min_i32(c >= 0 ? c : 0x7fffffff);
            const int cmax = wave_max_i32(c);
            if (cmax < 0) {
                int vmin = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 27; ++k) vmin = min(vmin, idx[dg][k] >= 0 ? idx[dg][k] : 0x7fffffff);
                cmin = wave_min_i32(vmin);
                lo = cmin - 32;
            } else {
                lo = (int)(((unsigned)cmin + (unsigned)cmax) >> 1) - LDS_CAP / 2;
            }
            lo = __builtin_amdgcn_readfirstlane(lo < 0 ? 0 : lo);
        }
        // ---- 3. classify (straight-line per active tap: selects, one ballot, two LDS stores): a byte offset into the LDS image for
        // every (tap, row) -- window hit / overflow slot / zero row -- and the list of missed rows
        int hmin = 0x7fffffff, hmax = -1;
        int n_ovf = 0;
        const uint32_t lane_lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
        const uint32_t lane_lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            if (!((ug >> k) & 1u)) continue;   // (wave-uniform)
            const int v = idx[dg][k];
            const bool hit = (uint32_t)(v - lo) < (uint32_t)LDS_CAP;   // (v = -1: v - lo wraps to a huge value)
            const bool miss = v >= 0 && !hit;
            const unsigned long long bal = __ballot(miss);
            const int slot = n_ovf + __builtin_popcount((uint32_t)bal & lane_lt_lo) + __builtin_popcount((uint32_t)(bal >> 32) & lane_lt_hi);
            const uint32_t off = hit ? (uint32_t)(v - lo) * RB : miss ? OVF_OFF + (uint32_t)slot * RB : ZERO_OFF;
            hmin = hit ? min(hmin, v) : hmin;
            hmax = hit ? max(hmax, v) : hmax;
            if (bal != 0ull) {   // (wave-uniform)
                if (miss && slot < LDS_OVF) s_list[slot] = (uint32_t)v;
                n_ovf += __builtin_popcountll(bal);
            }
            s_off[k * 64 + lane] = off;
        }
        const bool raw = n_ovf > LDS_OVF;   // (wave-uniform) more misses than the overflow area takes: this group gathers from memory
        if (raw) {
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                if (!((ug >> k) & 1u)) continue;
                const int v = idx[dg][k];
                s_off[k * 64 + lane] = v >= 0 ? (uint32_t)v : 0x7fffffffu;   // (absent: far beyond the buffer -> zeros)
            }
        } else {
            // ---- 4. the referenced part of the window, 1 KiB slabs (slab s = window bytes [1024 s, 1024 s + 1024))
            hmin = wave_min_i32(hmin);
            hmax = wave_max_i32(hmax);
            const int s_first = hmax >= 0 ? (int)(((uint32_t)(hmin - lo) * RB) >> 10) : 0;
            const int s_last = hmax >= 0 ? (int)(((uint32_t)(hmax - lo) * RB + RB - 1u) >> 10) : -1;
            const uint32_t wbase = (uint32_t)lo * RB + (uint32_t)lane * 16u;
            // ---- overflow rows: entry e of the list -> lane e (and e + 64), whole row, independent loads
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the list is complete (one wave, in-order LDS)
            f32x4 orow[2][RB / 16u];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = lane + 64 * h;
                if (h * 64 < n_ovf) {   // (wave-uniform)
                    const uint32_t gr = e < n_ovf ? s_list[e] : 0x7fffffffu;
                    const uint32_t ro = gr * RB;   // (entries past the end read beyond the buffer: zeros, never stored)
#pragma unroll
                    for (int q = 0; q < (int)(RB / 16u); ++q)
                        orow[h][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ro, (uint32_t)q * 16u, 0));
                }
            }
            // the window in batches of 8 slabs (8 KiB in flight per wave)
#pragma unroll
            for (int sb = 0; sb < NIT; sb += 8) {
                if (sb > s_last || sb + 7 < s_first) continue;   // (wave-uniform)
                f32x4 slab[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (sb + q >= s_first && sb + q <= s_last)   // (wave-uniform)
                        slab[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, wbase, (uint32_t)(sb + q) * 1024u, 0));
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (sb + q >= s_first && sb + q <= s_last)
                        *(f32x4*)(s_rows + (uint32_t)(sb + q) * 1024u + (uint32_t)lane * 16u) = slab[q];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = lane + 64 * h;
                if (h * 64 < n_ovf && e < n_ovf) {
#pragma unroll
                    for (int q = 0; q < (int)(RB / 16u); ++q) *(f32x4*)(s_rows + OVF_OFF + (uint32_t)e * RB + (uint32_t)q * 16u) = orow[h][q];
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): offsets and rows are in LDS

        // ---- 5. contraction: the group's active taps in ascending order.  Every memory operation of the loop is unconditional
        // and issued ahead of its use -- a tap's offsets two taps ahead, its B fragments (and weight fragment) one tap ahead --
        // so the waits are counted; only the MFMAs sit under (wave-uniform) per-tile branches.  A tile that does not have the tap
        // reads the zero row and is skipped.
        auto tile_has = [&](int kt, int t) -> bool {
            return kt < 64 ? ((tlo[t] >> kt) & 1ull) != 0ull : ((thi[t] >> (kt - 64)) & 1ull) != 0ull;
        };
        auto next_tap = [&](uint32_t& set, int keep) -> int {   // pops the lowest tap of `set`, or repeats `keep` when it is empty
            const int r = set ? __builtin_ctz(set) : keep;
            set &= set - 1u;
            return r;
        };
        auto read_offs = [&](int kk, uint32_t (&o)[4]) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = s_off[kk * 64 + t * 16 + j];
        };
        auto read_b = [&](auto rawc, const uint32_t (&o)[4], f32x4 (&bb)[4]) {
            constexpr bool RAW = decltype(rawc)::value;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (CK == 8) {
                    f32x2 t2;
                    if constexpr (RAW) t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, o[t] * RB + (uint32_t)g * LW, 0, 0));
                    else t2 = *(const f32x2*)(s_rows + o[t] + (uint32_t)g * LW);
                    bb[t] = (f32x4){t2[0], t2[1], 0.f, 0.f};
                } else {
                    if constexpr (RAW) bb[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, o[t] * RB + (uint32_t)g * LW, 0, 0));
                    else bb[t] = *(const f32x4*)(s_rows + o[t] + (uint32_t)g * LW);
                }
            }
        };
        auto contract = [&](auto rawc) {
            uint32_t rem = ug;
            const int n_taps = __builtin_popcount(ug);
            int k0 = next_tap(rem, 0);
            int k1 = next_tap(rem, k0);
            int k2 = next_tap(rem, k1);
            f32x4 a0[COT], a1[COT], b0[4];
            uint32_t o1[4];
            {
                uint32_t o0[4];
                read_offs(k0, o0);
                read_offs(k1, o1);
                load_w(dg * 27 + k0, a0);
                load_w(dg * 27 + k1, a1);
                read_b(rawc, o0, b0);
            }
            for (int it_tap = 0; it_tap < n_taps; ++it_tap) {
                // requests for the taps behind the current one
                uint32_t o2[4];
                f32x4 a2[COT], b1[4];
                read_offs(k2, o2);
                load_w(dg * 27 + k2, a2);
                read_b(rawc, o1, b1);
                // the current tap's MFMAs
                const int kt = dg * 27 + k0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!tile_has(kt, t)) continue;   // (wave-uniform)
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                        for (int it = 0; it < COT; ++it) acc[it][t] = MFMA(a0[it][s2], b0[t][s2], acc[it][t]);
                }
                // rotate
                k0 = k1; k1 = k2; k2 = next_tap(rem, k2);
#pragma unroll
                for (int t = 0; t < 4; ++t) { b0[t] = b1[t]; o1[t] = o2[t]; }
#pragma unroll
                for (int it = 0; it < COT; ++it) { a0[it] = a1[it]; a1[it] = a2[it]; }
            }
        };
        if (raw) contract(std::true_type{});
        else contract(std::false_type{});
        // (the next group rewrites s_off / s_rows: every LDS read above has returned -- its value fed an MFMA operand)
    }

    // ---- epilogue: lane (g, j) holds channels co0..co0+3 of row r0 + 16 t + j (the generic kernels' epilogue)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t orow_t = r0 + (uint32_t)t * 16u + (uint32_t)j;
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            const uint32_t co0 = (cg * COT + it) * 16 + 4 * g;
            if (orow_t >= n_out || co0 >= cout) continue;
            f32x4 v = acc[it][t];
            v += *(const f32x4*)(P.bias + co0);
            if (P.relu_pre) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (P.res_mode == 1) {
                const float* rp = P.res + (size_t)orow_t * P.ld_res + co0;
                if (P.vec_store && co0 + 3 < cout) {
                    v += *(const f32x4*)rp;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co0 + r < cout) v[r] += rp[r];
                }
            } else if (P.res_mode == 2) {
                const float* rp = P.res + (size_t)orow_t * P.ld_res + 2 * co0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < cout) v[r] += rp[2 * r] + rp[2 * r + 1];
            }
            if (P.relu_post) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            float* op = P.out + (size_t)orow_t * P.ld_out + co0;
            if (P.vec_store && co0 + 3 < cout) {
                *(f32x4*)op = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < cout) op[r] = v[r];
            }
        }
    }
}

int g_lds = -1;   // -1 = read INSMOS_CONV_LDS (default off); insmos_debug_conv_lds

}  // namespace

bool conv_lds_ok(const ConvP& P, int ck, int cot) {
    if (g_lds < 0) {
        // OFF by default: correct (bit-identical, tested) but measured 0.4-0.6 x the generic kernels' speed in this first form --
        // a wave runs seven dependent memory phases per block at two waves per SIMD
        const char* e = getenv("INSMOS_CONV_LDS");
        g_lds = e ? atoi(e) : 0;
    }
    // 81 taps in three groups of 27 (3^3 x 3: the time offset is the slowest tap digit), one chunk per tap, rows contiguous
    return g_lds && P.nbr && P.mask16 && P.K == 81 && ((ck == 8) || (ck == 0 && P.n16 == 1)) && P.ld_in == P.cin &&
           (cot == 1 || cot == 2) && P.ntile_co == cot && !(P.row0 & 15u);
}

bool conv_lds_try(const ConvP& Pin, int ck, int cot, long n_rows, hipStream_t s, int* rc) {
    if (!conv_lds_ok(Pin, ck, cot)) return false;
    ConvP P = Pin;
    P.n_otiles = (int)((n_rows + 63) / 64);
    const long blocks = (long)P.n_otiles * (P.ntile_co / cot);
    void (*kern)(ConvP) = ck == 8 ? (cot == 1 ? k_conv_lds<8, 1> : k_conv_lds<8, 2>) : (cot == 1 ? k_conv_lds<0, 1> : k_conv_lds<0, 2>);
    INSMOS_LAUNCH(kern, dim3((unsigned)blocks), dim3(64), 0, s, P);
    *rc = hipGetLastError() == hipSuccess ? INSMOS_OK : INSMOS_EHIP;
    return true;
}

}  // namespace insmos

extern "C" int insmos_debug_conv_lds(int on) {
    insmos::g_lds = on ? 1 : 0;
    return INSMOS_OK;
}
