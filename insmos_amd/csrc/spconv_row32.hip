// insmos_amd/csrc/spconv_row32.hip -- the Cin = 32 sparse convolutions with WHOLE-ROW gathers (round 5).
//
// Why (tools/probes/gather_rate_probe.hip on the MI355X, profiles/r05_gather_rate_probe.txt): the vector L1 of a CU serves one 128-byte
// line per two clocks whatever part of the line is asked for.  A B-fragment gather of the MFMA tiles -- lane (g, j) reads 16 B of row j:
// 16 rows x 64 B per instruction -- therefore costs 16 line slots for 1 KiB, and at Cin = 32 (a row IS one 128-byte line) the two chunk
// gathers of a tap walk the same 16 lines twice: 2 x 27 ns per CU when the rows sit in L1 / L2, against 8 ns for a weight fragment of
// the same size.  An instruction whose lane octets each read one whole row (8 rows x 128 B) costs 8.3 ns -- the rate of a coalesced
// stream.  With four SIMDs sharing that L1 the Cin = 32, Cout = 16 layers of spconv.hip (one channel tile per gathered row) were
// L1-bound by 2.5x.
//
// What changes against k_sparse_conv<COT, 1, 0, false, 3, SPLIT, false>:
//   * per tap TWO gathers: lane (r = lane >> 3, q = lane & 7) reads bytes [16 q, 16 q + 16) of the neighbour row of output row r, then
//     of output row 8 + r (its index comes from a dword load of its own: 8 distinct table entries per instruction, one line);
//   * the rows land in the wrong lanes for the 16 x 16 x 4 MFMA (row j must sit in the four lanes with lane & 15 == j), so they pass
//     through a wave-private LDS buffer (2.5 KiB): two ds_write_b128 (row-major, pitch 160 B), two ds_read_b128 -- lane (g, j) reads bytes
//     [64 c + 16 g, + 16) of row j for chunk c.  Pitch 160 B is conflict-free for the four hardware lane groups of ds_read_b128
//     (brute-forced like bev.hip's 96 B); the LDS pipe is otherwise idle in these kernels;
//   * a three-stage software pipeline over TAPS: gathers of tap t + 2 in flight, weight fragments of tap t + 1 requested and its rows
//     going through LDS, MFMAs of tap t.
// Per output element the MFMA chain is unchanged (tap ascending, chunk 0 then 1, steps 0..3; tap-split waves own taps k % 4 == ws and
// meet in LDS in wave order): the SAME BITS as spconv.hip (tests/test_gpu_conv.py::test_whole_row_gather_kernel_is_bitwise_the_generic_one).
#include <atomic>
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace insmos {
namespace {

__device__ __forceinline__ int r32_pop_or_keep(uint64_t& lo, uint64_t& hi, int keep) {
    const bool use_lo = lo != 0;
    const uint64_t w = use_lo ? lo : hi;
    const int k = (w ? __builtin_ctzll(w) : 0) + (use_lo ? 0 : 64);
    const uint64_t cleared = w & (w - 1);
    const bool any = w != 0;
    lo = use_lo ? cleared : lo;
    hi = use_lo ? hi : cleared;
    return any ? k : keep;
}

constexpr int kPitch = 160;               // bytes between rows of a staging buffer
constexpr int kStageBytes = 16 * kPitch;  // one tap: 16 rows

// SWZ (second form, no LDS): the gathers stay 16 rows x 64 B, but the two of a tap are HALF-SWIZZLED -- in the first one even rows
// read chunk 0 and odd rows chunk 1, in the second the other way round -- and a v_cndmask per register puts the chunks back.  The
// same probe: 16 rows x 64 B that all take the SAME half of their lines cost 27 ns per CU (a conflict on address bit 6), with the
// halves alternating 14 (L1) / 19 (L2) ns.  No staging, no second index load, the generic tiles' register footprint.
template <int COT, int SPLIT, bool SWZ>
__global__ void __launch_bounds__(SPLIT == 1 ? 64 : 256) k_conv_row32(ConvP P) {
    static_assert(SPLIT == 1 || SPLIT == 4, "one wave per tile, or four waves sharing a tile's taps");
    constexpr int NW = SPLIT == 1 ? 1 : 4;
    // ONE staging buffer per wave: the LDS executes a wave's instructions in order, so tap t + 1's ds_writes cannot pass tap t's ds_reads
    __shared__ __attribute__((aligned(16))) unsigned char stage[SWZ ? 1 : NW][SWZ ? 16 : kStageBytes];
    const int lane = threadIdx.x & 63;
    const uint32_t wib = SPLIT == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ws = wib;
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t n_cg = P.ntile_co / COT;
    const uint32_t n_tiles = (uint32_t)P.n_otiles * n_cg;
    const uint32_t tile_raw = blockIdx.x;
    const bool live = tile_raw < n_tiles;
    const uint32_t tile = live ? tile_raw : n_tiles - 1;
    const uint32_t cg = tile / P.n_otiles;
    const uint32_t ot = tile % P.n_otiles;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n_out * 4u), 0x00020000);
    constexpr uint32_t FR = 256u;                                // floats per weight fragment
    const uint32_t blk_stride = (uint32_t)P.ntile_co * FR;       // floats between the two chunk blocks of a tap
    const uint32_t tap_stride = 2u * blk_stride;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    const uint32_t cout = P.cout;

    // output rows: j for the epilogue; (lane >> 3) and 8 + (lane >> 3) for the whole-row gathers
    const uint32_t row_base = P.row0 + ot * 16u;
    const uint32_t orow = row_base + (uint32_t)j;
    // (SWZ: both gathers read the neighbour of the lane's own row j, 16 B at 16 g inside the half picked by the row's parity)
    const uint32_t ra = SWZ ? orow : row_base + (uint32_t)(lane >> 3), rb = SWZ ? orow : ra + 8u;
    const uint32_t roa = (ra < n_out ? ra : n_out - 1) * 4u, rob = (rb < n_out ? rb : n_out - 1) * 4u;
    const bool odd = (j & 1) != 0;
    const uint32_t qoff = SWZ ? (uint32_t)g * 16u + (odd ? 64u : 0u) : (uint32_t)(lane & 7) * 16u;
    const uint32_t qoff_b = SWZ ? (uint32_t)g * 16u + (odd ? 0u : 64u) : qoff;

    // ---- active taps of the tile (SGPRs), the tap-split residue class of this wave
    uint64_t tlo, thi;
    {
        const int K = P.K;
        const uint32_t ngrp = (n_out + 15) >> 4;
        const uint32_t grp = (P.row0 >> 4) + ot;
        if (P.mask16) {
            const uint32_t* mp = P.mask16 + (size_t)(grp < ngrp ? grp : ngrp - 1) * 4;
            const uint32_t w0 = __builtin_amdgcn_readfirstlane(mp[0]), w1 = __builtin_amdgcn_readfirstlane(mp[1]);
            const uint32_t w2 = __builtin_amdgcn_readfirstlane(mp[2]), w3 = __builtin_amdgcn_readfirstlane(mp[3]);
            tlo = grp < ngrp ? (((uint64_t)w1 << 32) | w0) : 0ull;
            thi = grp < ngrp ? (((uint64_t)w3 << 32) | w2) : 0ull;
        } else {
            tlo = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
            thi = K > 64 ? (K >= 128 ? ~0ull : ((1ull << (K - 64)) - 1ull)) : 0ull;
        }
    }
    int nt = __builtin_popcountll(tlo) + __builtin_popcountll(thi);
    if constexpr (SPLIT == 4) {
        if (P.tap_mod) {
            const uint64_t mine = 0x1111111111111111ull << ws;   // taps k with k % 4 == ws
            tlo &= mine;
            thi &= mine;
            nt = live ? __builtin_popcountll(tlo) + __builtin_popcountll(thi) : 0;
        } else {
            for (uint32_t q = 0; q < ws; ++q) (void)r32_pop_or_keep(tlo, thi, 0);
            nt = (live && nt > (int)ws) ? (nt - (int)ws + 3) / 4 : 0;
        }
    }
    auto next_tap = [&](int keep) {
        const int k = r32_pop_or_keep(tlo, thi, keep);
        if constexpr (SPLIT == 4) {
            if (!P.tap_mod) {
#pragma unroll
                for (int q = 1; q < 4; ++q) (void)r32_pop_or_keep(tlo, thi, 0);
            }
        }
        return k;
    };

    f32x4 acc[COT];
#pragma unroll
    for (int it = 0; it < COT; ++it) acc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t woffv[COT];
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co = (cg * COT + it) * 16u + (uint32_t)(lane & 15);
        woffv[it] = co < cout ? ((cg * COT + it) * FR + lane * 4u) * 4u : 0x7FFFFFF0u;
    }

    unsigned char* const my_stage = &stage[SWZ ? 0 : wib][0];
    const uint32_t wr_a = (uint32_t)(lane >> 3) * kPitch + qoff;             // bytes: row (lane >> 3), piece q
    const uint32_t wr_b = wr_a + 8u * kPitch;                                // row 8 + (lane >> 3)
    const uint32_t rd_0 = (uint32_t)j * kPitch + (uint32_t)g * 16u;          // chunk 0 of row j; chunk 1 is 64 B further

    if (nt > 0) {
        // tap ring of the load cursor: row byte offsets of the tap to request next (L), of the one after it (N), raw indices of the
        // tap after that in flight (NN) -- scaled offsets are what rotates, never a just-loaded register (see spconv.hip)
        int kL = next_tap(0);
        int kN = next_tap(kL);
        int kNN = next_tap(kN);
        auto load_idx = [&](int k, uint32_t& ia, uint32_t& ib) {
            ia = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, roa, (uint32_t)k * n_out * 4u, 0);
            if constexpr (SWZ) ib = ia;
            else ib = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, rob, (uint32_t)k * n_out * 4u, 0);
        };
        uint32_t offLa, offLb, offNa, offNb, idxNNa, idxNNb;
        {
            uint32_t i0a, i0b, i1a, i1b;
            load_idx(kL, i0a, i0b);
            load_idx(kN, i1a, i1b);
            load_idx(kNN, idxNNa, idxNNb);
            offLa = i0a * 128u + qoff; offLb = i0b * 128u + qoff_b;   // (-1 wraps past the end of the buffer: the load returns 0)
            offNa = i1a * 128u + qoff; offNb = i1b * 128u + qoff_b;
        }
        f32x4 ga[3], gb[3];            // gathered row pieces of taps in flight
        f32x4 as[3][2][COT];           // weight fragments [slot][chunk][channel tile]
        f32x4 fr[3][2];                // B fragments [slot][chunk], after the trip through LDS
        int kW = kL;   // tap whose weight fragments are requested next (one tap behind the gather cursor)
#define R32_REQD(slot)                                                                                                      \
    {                                                                                                                       \
        ga[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, offLa, 0, 0));                    \
        gb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, offLb, 0, 0));                    \
        kW = kL;                                                                                                            \
        kL = kN; kN = kNN;                                                                                                  \
        offLa = offNa; offLb = offNb;                                                                                       \
        offNa = idxNNa * 128u + qoff; offNb = idxNNb * 128u + qoff_b;                                                       \
        kNN = next_tap(kNN);                                                                                                \
        load_idx(kNN, idxNNa, idxNNb);                                                                                      \
    }
#define R32_REQW(slot)                                                                                                      \
    {                                                                                                                       \
        const uint32_t sw = (uint32_t)kW * tap_stride * 4u;                                                                 \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                       \
            _Pragma("unroll") for (int it = 0; it < COT; ++it)                                                              \
                as[slot][c][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw + (uint32_t)c * blk_stride * 4u, 0)); \
    }
#define R32_STAGE(slot)                                                                                                     \
    if constexpr (!SWZ) {                                                                                                   \
        unsigned char* sb = my_stage;                                                                                       \
        __builtin_amdgcn_wave_barrier();   /* (scheduling only: the reads below are other lanes' writes) */                 \
        *(f32x4*)(sb + wr_a) = ga[slot];                                                                                    \
        *(f32x4*)(sb + wr_b) = gb[slot];                                                                                    \
        __builtin_amdgcn_wave_barrier();                                                                                    \
        fr[slot][0] = *(const f32x4*)(sb + rd_0);                                                                           \
        fr[slot][1] = *(const f32x4*)(sb + rd_0 + 64);                                                                      \
        __builtin_amdgcn_wave_barrier();                                                                                    \
    }
#define R32_MMA(slot)                                                                                                       \
    {                                                                                                                       \
        if constexpr (SWZ) {   /* un-swizzle: even rows hold (chunk 0, chunk 1) in (ga, gb), odd rows the other way round */ \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                 \
                fr[slot][0][e] = odd ? gb[slot][e] : ga[slot][e];                                                           \
                fr[slot][1][e] = odd ? ga[slot][e] : gb[slot][e];                                                           \
            }                                                                                                               \
        }                                                                                                                   \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                       \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                   \
                _Pragma("unroll") for (int it = 0; it < COT; ++it)                                                          \
                    acc[it] = MFMA(as[slot][c][it][s], fr[slot][c][s], acc[it]);                                            \
    }
        // prologue: gathers of taps 0 and 1 and the weights of tap 0 requested, tap 0 staged (running off the end of the tap list
        // re-requests the last tap: clamp, no guard).  Weight fragments (L1 / L2 hits) are requested ONE tap ahead, gathers two.
        R32_REQD(0)
        R32_REQW(0)
        R32_REQD(1)
        R32_STAGE(0)
        int t = 0;
        for (; t + 3 <= nt; t += 3) {
            R32_REQW(1) R32_REQD(2) R32_STAGE(1) R32_MMA(0)
            R32_REQW(2) R32_REQD(0) R32_STAGE(2) R32_MMA(1)
            R32_REQW(0) R32_REQD(1) R32_STAGE(0) R32_MMA(2)
        }
        // tail (wave-uniform): taps t, t + 1 -- their gathers are in flight (slots 0, 1), tap t is staged and has its weights
        if (t < nt) {
            R32_REQW(1)
            R32_STAGE(1)
            R32_MMA(0)
            if (t + 1 < nt) R32_MMA(1)
        }
#undef R32_REQD
#undef R32_REQW
#undef R32_STAGE
#undef R32_MMA
    }

    // ---- epilogue: lane (g, j) holds channels co0 .. co0 + 3 of row orow (spconv.hip's, unchanged)
    auto finish = [&](int it, f32x4 v) {
        const uint32_t co0 = (cg * COT + it) * 16 + 4 * g;
        const uint32_t o = orow;
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
        for (int it = 0; it < COT; ++it) finish(it, acc[it]);
    } else {
        __shared__ f32x4 red[4][COT][64];  // [wave][channel tile][lane]
#pragma unroll
        for (int it = 0; it < COT; ++it) red[wib][it][lane] = acc[it];
        __syncthreads();
        if (live && (COT >= 4 || ws < (uint32_t)COT)) {   // (fewer channel tiles than waves: the first COT waves finish)
            constexpr int PER = COT >= 4 ? COT / 4 : 1;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int it = (int)ws * PER + q;
                f32x4 v = red[0][it][lane];
#pragma unroll
                for (int p = 1; p < 4; ++p) v += red[p][it][lane];   // fixed order -> deterministic, and the tap-split tiles' order
                finish(it, v);
            }
        }
    }
}

typedef void (*R32Kernel)(ConvP);
template <bool SWZ>
R32Kernel pick_row32(int cot, int split) {
    if (split == 1) {
        if (cot == 1) return k_conv_row32<1, 1, SWZ>;
        if (cot == 2) return k_conv_row32<2, 1, SWZ>;
        if (cot == 4) return k_conv_row32<4, 1, SWZ>;
    } else if (split == 4) {
        if (cot == 1) return k_conv_row32<1, 4, SWZ>;
        if (cot == 2) return k_conv_row32<2, 4, SWZ>;
        if (cot == 4) return k_conv_row32<4, 4, SWZ>;
    }
    return nullptr;
}

std::atomic<int> g_row32_dbg{-1};   // insmos_debug_conv_row32 (test hook; atomic: may be flipped while another host thread launches): -1 = none
}  // namespace

// the kernel for this launch shape, or null: Cin = 32 with rows that ARE 128-byte lines (pitch 32 floats, 128-byte aligned base), a
// neighbour table, 16-row tiles (one wave, or four tap-split waves) of 1 / 2 / 4 channel tiles
ConvKernelFn conv_row32_pick(const ConvP& P, int cot, int jt, int split, bool by_chunk) {
    static const int row32_env = [] { const char* e = getenv("INSMOS_CONV_ROW32"); const int v = e ? atoi(e) : 1; return (v < 0 || v > 3) ? 1 : v; }();
    const int dbg = g_row32_dbg.load(std::memory_order_relaxed);
    const int g_row32 = dbg >= 0 ? dbg : row32_env;
    if (!g_row32 || !P.nbr || P.cin != 32 || P.n16 != 2 || P.has8 || P.has4 || P.ld_in != 32 || ((uintptr_t)P.in & 127) || jt != 1 || by_chunk)
        return nullptr;
    // Where it pays (per layer on a launch set of 8, profiles/r05_row32_layers.txt): the Cout = 16 layers with a real tap list --
    // block7.0.conv1 284 -> 241 us, conv_up_instance_block_up2 / up1 71 / 60 -> 57 / 47, conv_up_m1.0 68 -> 52, inv_conv2.0 53 -> 49:
    // one channel tile per gathered row, the L1 was their bound.  With two channel tiles (Cout 32) the MFMA time per tap equals the
    // L1 time even on the generic tiles and the kernel's larger register footprint buys nothing (+-2 %); the 8-tap k2s2 maps (one
    // or two active taps per group: all prologue) and Cout 64 lose 10-25 %.
    // The half-swizzled form (no LDS, the generic footprint) on the same table: Cout 16 layers 250 / 59 / 57 / 50 -- behind the staged
    // form --, the 27-tap Cout 32 layers 86.7 / 80.3 / 82.5 / 81.5 -> 81.2 / 76.2 / 78.7 / 76.7 (-5 %), the 81-tap ones +-2 %.
    // g_row32: 1 = this rule, 2 = the staged form on every shape, 3 = the half-swizzled form on every shape (test hooks), 0 = off.
    if (g_row32 == 3) return pick_row32<true>(cot, split);
    if (g_row32 == 2) return pick_row32<false>(cot, split);
    if (cot == 1 && P.K >= 16) return pick_row32<false>(cot, split);
    if (cot == 2 && P.K >= 16 && P.K <= 32) return pick_row32<true>(cot, split);
    return nullptr;
}

}  // namespace insmos

extern "C" int insmos_debug_conv_row32(int on) {
    if (on < -1 || on > 3) return INSMOS_EINVAL;   // (2 / 3: the staged / the half-swizzled form on every shape the kernel is built for)
    insmos::g_row32_dbg.store(on, std::memory_order_relaxed);
    return INSMOS_OK;
}
