// insmos_amd/csrc/spconv_lds.hip -- LDS-staged gathers for the 81-tap (3 x 3 x 3 x 3) single-chunk layers of MotionNet
// (BasicBlock convolutions with Cin = 8 or 16: models/MinkowskiEngine/resnet.py:110-119, minkunet.py:52-137).
//
// Why.  These layers run at 12-25 TFLOP/s on the generic kernels (spconv.hip): a (16-row tile, tap) costs a vector-memory
// gather that touches 16 distinct cache lines to use 32-64 bytes of each, plus an index load and a weight fragment, for 2-4
// MFMAs -- the texture-address path is the bound, the matrix cores idle ~60 %.  But the gathers are LOCAL: rows are in
// (time, Morton) order, so the 27 spatial neighbours of a run of consecutive rows at one time offset sit, for ~90 % of the present
// (row, tap) pairs, inside ONE run of a few hundred consecutive input rows, and a block re-reads each distinct neighbour row
// ~4.5 x (tools/gather_locality.py, tools/gather_window_policy.py on the S0 window).
//
// What (second form).  A 256-thread workgroup owns 256 consecutive output rows -- wave w the four 16-row tiles of rows
// [64 w, 64 w + 64) x all channel tiles.  Per time offset dt (a group of 27 consecutive taps), in lock step:
//   A. every wave loads its rows' neighbour indices (lane = row, one coalesced dword load per tap its tiles use);
//   B. the workgroup places a window of CAPW consecutive input rows around the centre tap's neighbours (the voxels themselves at
//      t + dt; min / max over the four waves through LDS);
//   C. every wave classifies its (tap, row) pairs: inside the window -> its row slot; present but outside (~10 %: the
//      neighbourhood crosses a Morton boundary) -> a slot of the shared overflow area (LDS counter) and an entry in the miss list;
//      absent -> the zero row.  Slots are 16-bit, [27][64] per wave;
//   D. the 256 threads copy the referenced part of the window with coalesced 16-byte loads and fetch the listed rows, one
//      list entry per thread -- all loads independent -- into LDS;
//   E. every wave contracts: B fragments by ds_read_b64 / b128 from its slots, no vector-memory gather in the loop; a tap's weight
//      fragment is fetched once per 64 rows (two taps ahead).
// A workgroup whose misses exceed the overflow area (rare) writes raw indices where the window would be and gathers from memory.
// The first form (one wave per 64 rows, private window: commit 1a98267) measured 0.4-0.6 x the generic kernels: a wave ran seven
// dependent memory phases per block at two waves per SIMD.  Sharing the window halves the staging traffic and the LDS per wave,
// so 8-12 waves fit a CU and one workgroup's phases hide behind the others' contractions.
//
// Same bits as the generic kernels: per 16-row tile the taps are walked in ascending order through the tile's own active-tap
// mask, each step is the same v_mfma_f32_16x16x4_f32 sequence on the same operand values (tests/test_gpu_conv.py).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

namespace insmos {
namespace {

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
// DBG (probe build, INSMOS_CONV_LDS=2): wave 0 of every workgroup adds its phase clocks and counters to g_lds_stats
}  // namespace
__device__ unsigned long long g_lds_stats[16];
namespace {
template <int CK, int COT, int DBG = 0>
__global__ void __launch_bounds__(256, CK == 8 ? 3 : 2) k_conv_ldsw(ConvP P) {
    constexpr uint32_t LW = CK == 8 ? 8u : 16u;   // bytes per lane per B fragment
    constexpr uint32_t RB = 4u * LW;              // bytes per input row (rows are contiguous: ld_in == cin)
    constexpr uint32_t LWF = CK == 8 ? 2u : 4u;   // floats per lane per weight fragment
    constexpr uint32_t FR = 64u * LWF;
    constexpr int NS = CK == 8 ? 2 : 4;
    constexpr int CAPW = CK == 8 ? 512 : 384;     // window rows
    constexpr int OVFW = CK == 8 ? 512 : 320;     // overflow rows
    // LDS row pitch = the row size.  (A padded pitch -- 40 / 80 bytes, so that a 16-row gather spreads over the bank groups -- was
    // measured with the probe build: no change, the contraction is not LDS-bank bound)
    constexpr uint32_t RP = RB;
    constexpr uint32_t ZERO_SLOT = CAPW + OVFW;
    constexpr int NSLAB = (int)(CAPW * RB / 4096u);   // 4 KiB steps of the window copy (256 threads x 16 bytes)
    static_assert((CAPW * RB) % 4096u == 0 && 27 * 64 * 4 * 4 <= (CAPW + OVFW) * (int)RP, "window steps / raw-index table fit");
    __shared__ uint16_t s_slot[4][27 * 64];
    __shared__ uint32_t s_list[OVFW];
    __shared__ int s_red[4][4];
    __shared__ int s_cnt;
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[(CAPW + OVFW + 1) * RP];

    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t n4 = n_out * 4u;
    const uint32_t r0 = P.row0 + blockIdx.x * 256u + wv * 64u;   // this wave's first row
    const uint32_t my_row = r0 + (uint32_t)lane;
    const bool row_ok = my_row < n_out;
    const uint32_t cout = P.cout;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb = __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n4), 0x00020000);
    const uint32_t tap_stride = (uint32_t)P.ntile_co * FR;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);

    // ---- this wave's four tiles' active-tap sets (scalar) and their union
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
    if (tid < (int)(RB / 4u)) *(uint32_t*)(s_rows + ZERO_SLOT * RP + 4u * (uint32_t)tid) = 0u;   // the zero row
    if (tid == 0) s_cnt = 0;

    f32x4 acc[COT][4];
#pragma unroll
    for (int it = 0; it < COT; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[it][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t woffv[COT];  // lanes whose output channel lies beyond Cout read zeros from past the end of the buffer
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co = (uint32_t)it * 16u + (uint32_t)j;
        woffv[it] = co < cout ? ((uint32_t)it * FR + (uint32_t)lane * LWF) * 4u : 0x7FFFFFF0u;
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
    const uint32_t rowoff = (row_ok ? my_row : n_out - 1u) * 4u;
    const uint32_t lane_lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xFFFFFFFFu;
    const uint32_t lane_lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);

    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int slot, unsigned long long& t0) {
        if constexpr (DBG) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            tk[slot] += t1 - t0;
            t0 = t1;
        }
    };
#pragma unroll 1
    for (int dg = 0; dg < 3; ++dg) {
        unsigned long long t0 = 0;
        if constexpr (DBG) t0 = __builtin_readcyclecounter();
        const uint32_t ug = group_bits(ulo, uhi, dg);   // (wave-uniform; the other waves of the workgroup have their own)
        // ---- A. my row's neighbour index under every tap this wave's tiles use
        int idx[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            idx[k] = -1;
            if ((ug >> k) & 1u) idx[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_nb, rowoff, (uint32_t)(dg * 27 + k) * n4, 0);
        }
        {
            // tables written with sparse stores (k_resolve_taps<.., 2>, insmos_build_nbr_rank_sparse) hold NOTHING in the slots of a
            // 16-row group that lacks the tap: a lane keeps only the entries of its OWN tile's taps (garbage there would steer the
            // window placement, the miss list and the raw fallback -- correct results through the MFMA gating, but a random path)
            const uint64_t mlo = g == 0 ? tlo[0] : g == 1 ? tlo[1] : g == 2 ? tlo[2] : tlo[3];
            const uint64_t mhi = g == 0 ? thi[0] : g == 1 ? thi[1] : g == 2 ? thi[2] : thi[3];
            const uint32_t mg = row_ok ? group_bits(mlo, mhi, dg) : 0u;
#pragma unroll
            for (int k = 0; k < 27; ++k)
                if (!((mg >> k) & 1u)) idx[k] = -1;
        }
        // ---- B. window placement: centred on the centre tap's neighbours; a workgroup none of whose voxels exists at t + dt
        // centres on its smallest neighbour index.  (The barrier also closes the previous group: every wave is done with s_rows.)
        {
            const int c = idx[13];
            const int cmin = wave_min_i32(c >= 0 ? c : 0x7fffffff), cmax = wave_max_i32(c);
            int vmin = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 27; ++k) vmin = min(vmin, idx[k] >= 0 ? idx[k] : 0x7fffffff);
            vmin = wave_min_i32(vmin);
            if (lane == 0) { s_red[wv][0] = cmin; s_red[wv][1] = cmax; s_red[wv][2] = vmin; }
        }
        tick(0, t0);   // A: index loads + wave reductions
        __syncthreads();
        tick(1, t0);   // barrier 1 (also waits for the slowest wave's previous contraction)
        int lo;
        bool any;
        {
            const int cmin = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
            const int cmax = max(max(s_red[0][1], s_red[1][1]), max(s_red[2][1], s_red[3][1]));
            const int vmin = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
            any = vmin != 0x7fffffff;
            lo = cmax >= 0 ? (int)(((unsigned)cmin + (unsigned)cmax) >> 1) - CAPW / 2 : vmin - 32;
            lo = __builtin_amdgcn_readfirstlane(lo < 0 ? 0 : lo);
        }
        __syncthreads();   // (s_red is rewritten below)
        if (!any) continue;   // (workgroup-uniform) no neighbour at this time offset anywhere in the block
        // ---- C. classify (straight-line per tap: selects, one ballot, LDS stores); misses take overflow slots from the shared counter
        int hmin = 0x7fffffff, hmax = -1;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            if (!((ug >> k) & 1u)) continue;   // (wave-uniform)
            const int v = idx[k];
            const bool hit = (uint32_t)(v - lo) < (uint32_t)CAPW;   // (v = -1: v - lo wraps to a huge value)
            const bool miss = v >= 0 && !hit;
            const unsigned long long bal = __ballot(miss);
            uint32_t slot = hit ? (uint32_t)(v - lo) : ZERO_SLOT;
            hmin = hit ? min(hmin, v) : hmin;
            hmax = hit ? max(hmax, v) : hmax;
            if (bal != 0ull) {   // (wave-uniform)
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_cnt, __builtin_popcountll(bal));
                base = __builtin_amdgcn_readfirstlane(base);
                const int e = base + __builtin_popcount((uint32_t)bal & lane_lt_lo) + __builtin_popcount((uint32_t)(bal >> 32) & lane_lt_hi);
                if (miss) {
                    if (e < OVFW) s_list[e] = (uint32_t)v;
                    slot = (uint32_t)(CAPW + (e < OVFW ? e : 0));   // (past the end: the whole group falls back below)
                }
            }
            s_slot[wv][k * 64 + lane] = (uint16_t)slot;
        }
        hmin = wave_min_i32(hmin);
        hmax = wave_max_i32(hmax);
        if (lane == 0) { s_red[wv][0] = hmin; s_red[wv][1] = hmax; }
        tick(2, t0);   // C: classify
        __syncthreads();
        const int n_ovf = s_cnt;
        const bool raw = n_ovf > OVFW;   // (workgroup-uniform) more misses than the overflow area takes
        if (raw) {
            // raw neighbour indices where the window would be ([wave][27][64] dwords): this group gathers from memory
            uint32_t* s_raw = (uint32_t*)s_rows + wv * (27 * 64);
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                if (!((ug >> k) & 1u)) continue;
                s_raw[k * 64 + lane] = idx[k] >= 0 ? (uint32_t)idx[k] : 0x7fffffffu;   // (absent: far beyond the buffer -> zeros)
            }
        } else {
            // ---- D. the referenced part of the window (4 KiB per step over the 256 threads) + the listed rows
            hmin = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
            hmax = max(max(s_red[0][1], s_red[1][1]), max(s_red[2][1], s_red[3][1]));
            const int s_first = hmax >= 0 ? (int)(((uint32_t)(hmin - lo) * RB) >> 12) : 0;
            const int s_last = hmax >= 0 ? (int)(((uint32_t)(hmax - lo) * RB + RB - 1u) >> 12) : -1;
            const uint32_t wbase = (uint32_t)lo * RB + (uint32_t)tid * 16u;
            f32x4 slab[NSLAB];
#pragma unroll
            for (int sI = 0; sI < NSLAB; ++sI)
                if (sI >= s_first && sI <= s_last)   // (workgroup-uniform)
                    slab[sI] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, wbase, (uint32_t)sI * 4096u, 0));
            // (list entries beyond the first 256 are rare: a second round trip, one buffer)
            constexpr int NE = (OVFW + 255) / 256;
            f32x4 orow[RB / 16u];
            {
                const uint32_t gr = tid < n_ovf ? s_list[tid] : 0x7fffffffu;
                const uint32_t ro = gr * RB;   // (entries past the end read beyond the buffer: zeros, never stored)
                if (n_ovf > 0) {   // (workgroup-uniform)
#pragma unroll
                    for (int q = 0; q < (int)(RB / 16u); ++q)
                        orow[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ro, (uint32_t)q * 16u, 0));
                }
            }
#pragma unroll
            for (int sI = 0; sI < NSLAB; ++sI)
                if (sI >= s_first && sI <= s_last) {
                    const uint32_t wb = (uint32_t)sI * 4096u + (uint32_t)tid * 16u;          // byte inside the window
                    unsigned char* dst = s_rows + (wb / RB) * RP + (wb % RB);
                    if constexpr (CK == 8) {   // (40-byte pitch: 8-byte aligned only)
                        *(f32x2*)dst = (f32x2){slab[sI][0], slab[sI][1]};
                        *(f32x2*)(dst + 8) = (f32x2){slab[sI][2], slab[sI][3]};
                    } else {
                        *(f32x4*)dst = slab[sI];
                    }
                }
            if (tid < n_ovf) {
#pragma unroll
                for (int q = 0; q < (int)(RB / 16u); ++q) *(f32x4*)(s_rows + (uint32_t)(CAPW + tid) * RB + (uint32_t)q * 16u) = orow[q];
            }
#pragma unroll 1
            for (int h = 1; h < NE; ++h) {
                if (h * 256 >= n_ovf) break;   // (workgroup-uniform)
                const int e = tid + 256 * h;
                const uint32_t gr = e < n_ovf ? s_list[e] : 0x7fffffffu;
                const uint32_t ro = gr * RB;
#pragma unroll
                for (int q = 0; q < (int)(RB / 16u); ++q)
                    orow[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ro, (uint32_t)q * 16u, 0));
                if (e < n_ovf) {
#pragma unroll
                    for (int q = 0; q < (int)(RB / 16u); ++q) {
                        unsigned char* dst = s_rows + (uint32_t)(CAPW + e) * RP + (uint32_t)q * 16u;
                        if constexpr (CK == 8) {
                            *(f32x2*)dst = (f32x2){orow[q][0], orow[q][1]};
                            *(f32x2*)(dst + 8) = (f32x2){orow[q][2], orow[q][3]};
                        } else {
                            *(f32x4*)dst = orow[q];
                        }
                    }
                }
            }
        }
        tick(3, t0);   // D: staging loads + LDS writes (incl. barrier 3)
        __syncthreads();
        if (tid == 0) s_cnt = 0;   // (everyone has read it; the next group's counting starts behind its barrier B)
        tick(4, t0);   // barrier 4
        if constexpr (DBG) {
            if (tid == 0) {
                atomicAdd(&g_lds_stats[8], 1ull);
                atomicAdd(&g_lds_stats[9], raw ? 1ull : 0ull);
                atomicAdd(&g_lds_stats[10], (unsigned long long)n_ovf);
            }
        }

        // ---- E. contraction: this wave's active taps of the group in ascending order.  Every memory operation of the loop is
        // unconditional and issued ahead of its use -- a tap's slots two taps ahead, its B fragments one tap ahead, its weight
        // fragment two taps ahead -- so the waits are counted; only the MFMAs sit under (wave-uniform) per-tile branches.  A tile
        // that does not have the tap reads the zero row and is skipped.
        if (ug == 0u) continue;   // (wave-uniform; the barriers above were taken)
        auto tile_has = [&](int kt, int t) -> bool {
            return kt < 64 ? ((tlo[t] >> kt) & 1ull) != 0ull : ((thi[t] >> (kt - 64)) & 1ull) != 0ull;
        };
        auto next_tap = [&](uint32_t& set, int keep) -> int {   // pops the lowest tap of `set`, or repeats `keep` when it is empty
            const int r = set ? __builtin_ctz(set) : keep;
            set &= set - 1u;
            return r;
        };
        auto contract = [&](auto rawc) {
            constexpr bool RAW = decltype(rawc)::value;
            auto read_offs = [&](int kk, uint32_t (&o)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (RAW) o[t] = ((const uint32_t*)s_rows)[wv * (27 * 64) + kk * 64 + t * 16 + j] * RB + (uint32_t)g * LW;
                    else o[t] = (uint32_t)s_slot[wv][kk * 64 + t * 16 + j] * RP + (uint32_t)g * LW;
                }
            };
            auto read_b = [&](const uint32_t (&o)[4], f32x4 (&bb)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (CK == 8) {
                        f32x2 t2;
                        if constexpr (RAW) t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, o[t], 0, 0));
                        else t2 = *(const f32x2*)(s_rows + o[t]);
                        bb[t] = (f32x4){t2[0], t2[1], 0.f, 0.f};
                    } else {
                        if constexpr (RAW) bb[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, o[t], 0, 0));
                        else bb[t] = *(const f32x4*)(s_rows + o[t]);
                    }
                }
            };
            // Operand rings of depth 3 with STATIC slot indices (the loop is unrolled by three): rotating the slots by copying
            // registers would make every iteration wait for the load it has just issued -- the copy needs the value (measured
            // on the first two forms of this kernel: one exposed memory latency per tap, 0.5-0.65 x the generic kernels).
            uint32_t rem = ug;
            const int n_taps = __builtin_popcount(ug);
            int kq[3];
            kq[0] = next_tap(rem, 0);
            kq[1] = next_tap(rem, kq[0]);
            kq[2] = next_tap(rem, kq[1]);
            f32x4 ar[3][COT], br[3][4];
            uint32_t orr[3][4];
            read_offs(kq[0], orr[0]);
            read_offs(kq[1], orr[1]);
            load_w(dg * 27 + kq[0], ar[0]);
            load_w(dg * 27 + kq[1], ar[1]);
            read_b(orr[0], br[0]);
            for (int i = 0; i < n_taps; i += 3) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    // requests for the taps behind the current one: tap i + r + 2 -> slot (r + 2) % 3, B of tap i + r + 1 -> slot (r + 1) % 3
                    read_offs(kq[(r + 2) % 3], orr[(r + 2) % 3]);
                    load_w(dg * 27 + kq[(r + 2) % 3], ar[(r + 2) % 3]);
                    read_b(orr[(r + 1) % 3], br[(r + 1) % 3]);
                    const int kt = dg * 27 + kq[r];
                    if (i + r < n_taps) {   // (wave-uniform; past the end the ring re-requests the last tap and nothing is added)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (!tile_has(kt, t)) continue;   // (wave-uniform)
#pragma unroll
                            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                                for (int it = 0; it < COT; ++it) acc[it][t] = MFMA(ar[r][it][s2], br[r][t][s2], acc[it][t]);
                        }
                    }
                    kq[r] = next_tap(rem, kq[(r + 2) % 3]);   // tap i + r + 3 (scalar)
                }
            }
        };
        if (raw) contract(std::true_type{});
        else contract(std::false_type{});
        tick(5, t0);   // E: contraction
        if constexpr (DBG) {
            if (lane == 0) atomicAdd(&g_lds_stats[11], (unsigned long long)__builtin_popcount(ug));
        }
    }
    if constexpr (DBG) {
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) atomicAdd(&g_lds_stats[q], tk[q]);
            atomicAdd(&g_lds_stats[12], 1ull);
        }
    }

    // ---- epilogue: lane (g, j) holds channels co0..co0+3 of row r0 + 16 t + j (the generic kernels' epilogue)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t orow_t = r0 + (uint32_t)t * 16u + (uint32_t)j;
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            const uint32_t co0 = (uint32_t)it * 16 + 4 * g;
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

int g_lds = -1;   // -1 = read INSMOS_CONV_LDS; insmos_debug_conv_lds

}  // namespace

bool conv_lds_ok(const ConvP& P, int ck, int cot) {
    if (g_lds < 0) {
        const char* e = getenv("INSMOS_CONV_LDS");
        g_lds = e ? atoi(e) : 0;   // (off until measured faster: DESIGN.md 3.1b)
    }
    // 81 taps in three groups of 27 (3^3 x 3: the time offset is the slowest tap digit), one chunk per tap, rows contiguous
    return g_lds && P.nbr && P.mask16 && P.K == 81 && ((ck == 8) || (ck == 0 && P.n16 == 1)) && P.ld_in == P.cin &&
           (cot == 1 || cot == 2) && P.ntile_co == cot && !(P.row0 & 15u);
}

bool conv_lds_try(const ConvP& Pin, int ck, int cot, long n_rows, hipStream_t s, int* rc) {
    if (!conv_lds_ok(Pin, ck, cot)) return false;
    ConvP P = Pin;
    const long blocks = (n_rows + 255) / 256;
    void (*kern)(ConvP) = ck == 8 ? (cot == 1 ? k_conv_ldsw<8, 1> : k_conv_ldsw<8, 2>) : (cot == 1 ? k_conv_ldsw<0, 1> : k_conv_ldsw<0, 2>);
    if (g_lds == 2) kern = ck == 8 ? (cot == 1 ? k_conv_ldsw<8, 1, 1> : k_conv_ldsw<8, 2, 1>) : (cot == 1 ? k_conv_ldsw<0, 1, 1> : k_conv_ldsw<0, 2, 1>);
    INSMOS_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), 0, s, P);
    *rc = hipGetLastError() == hipSuccess ? INSMOS_OK : INSMOS_EHIP;
    return true;
}

}  // namespace insmos

extern "C" int insmos_debug_conv_lds(int on) {
    insmos::g_lds = on;   // 0 off, 1 on, 2 on with the probe build (insmos_debug_conv_lds_stats)
    return INSMOS_OK;
}

// probe build counters: [0..5] cycles of wave 0 per workgroup in phases A, barrier 1, C, D, barrier 4, E; [8] groups, [9] raw groups,
// [10] overflow rows, [11] active taps (summed over waves), [12] workgroups.  reset != 0 clears them after the read.
extern "C" int insmos_debug_conv_lds_stats(unsigned long long* out16_host, int reset) {
    if (!out16_host) return INSMOS_EINVAL;
    HIP_TRY(hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(insmos::g_lds_stats), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(insmos::g_lds_stats), z, sizeof(z)));
    }
    return INSMOS_OK;
}
