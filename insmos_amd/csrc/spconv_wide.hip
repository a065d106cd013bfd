// insmos_amd/csrc/spconv_wide.hip -- the WIDE 3D layers (Cin 64 / 128 / 256, Cout 64 / 128: spconv_unet.py:146-160, 181-200) on 32-row
// tiles with the gathered rows staged ONCE per tap in LDS (round 6).
//
// Why.  spconv.hip runs these layers on chunk-split tiles: four waves share a 16-row group, wave ws contracts the 16-channel chunks
// c % 4 == ws over all taps, every wave fetches its own chunk pieces of the 16 rows (16 rows x 64 B: 16 lines, 27 ns per CU) and
// the weight fragments of all its (chunk, channel tile) pairs (8 ns each) -- per (16-row group, tap) of a 128 -> 128 layer that is
// 8 gathers + 64 fragments = 0.73 us of vector-L1 time against 0.98 us of MFMA time: both pipes ~70 % busy, neither hidden behind
// the other (0.53 of the fp32 MFMA peak; the dense BEV kernel, whose operands come out of LDS, holds 0.81).  Here
//   * a workgroup of four waves owns 32 output rows (two 16-row groups) and ALL output channels; wave w owns the channel tiles
//     {COTW w .. COTW w + COTW - 1}: a weight fragment is fetched once per (tap, chunk, tile) of the WORKGROUP and feeds both row
//     groups (half the fragment traffic per MFMA), and no partial sums have to meet: every wave ends with whole output elements;
//   * per tap the 32 neighbour rows are gathered WHOLE (lanes of a wave read consecutive 16-byte pieces of a row: 1-4 rows x
//     1024-256 B per instruction -- the L1 charges lines, 8 ns per KiB instead of 27), written to a double-buffered LDS stage (rows
//     of 512 B with the 16-byte pieces XOR-swizzled by the row: conflict-free writes and B-fragment reads) and read from there by
//     all four waves; one barrier per tap;
//   * the summation order is the chunk-split tiles': four partial chains over the chunks c % 4 == 0..3 (taps ascending, chunks
//     ascending, MFMA steps 0..3), summed ((c0 + c1) + c2) + c3 -- a wave keeps FOUR accumulators per (channel tile, row group) and
//     adds them in that order in its epilogue: THE SAME BITS (tests/test_gpu_conv.py::test_wide_staged_kernel_is_bitwise_the_split_tiles).
// Cin = 256 rows (1 KiB) are staged in two halves (chunks 0..7, then 8..15: the class chains' chunk order is kept).
#include <atomic>
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace insmos {
namespace {

__device__ __forceinline__ int wide_pop_or_keep(uint64_t& lo, uint64_t& hi, int keep) {
    const bool use_lo = lo != 0;
    const uint64_t w = use_lo ? lo : hi;
    const int k = (w ? __builtin_ctzll(w) : 0) + (use_lo ? 0 : 64);
    const uint64_t cleared = w & (w - 1);
    const bool any = w != 0;
    lo = use_lo ? cleared : lo;
    hi = use_lo ? hi : cleared;
    return any ? k : keep;
}

// NCH: 16-channel chunks staged per step (4 = 256-byte row pieces, 8 = 512-byte); NH: steps per tap (Cin = 16 NCH NH);
// COTW: channel tiles per wave (Cout = 64 COTW)
template <int NCH, int NH, int COTW>
__global__ void __launch_bounds__(256) k_conv_wide(ConvP P) {
    static_assert(NCH == 4 || NCH == 8, "staged row pieces of 256 or 512 bytes");
    constexpr int RB = NCH * 64;               // staged bytes per row
    constexpr int RPI = 1024 / RB;             // rows per gather instruction (2 or 4)
    constexpr int NG = 8 / RPI;                // gather instructions per wave and step (the wave stages 8 of the 32 rows)
    constexpr int LPR = 64 / RPI;              // lanes per row
    __shared__ __attribute__((aligned(128))) unsigned char stage[2][32 * RB];
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t cout = P.cout;
    const uint32_t tile = blockIdx.x;
    const uint32_t row_base = P.row0 + tile * 32u;
    const uint32_t ngrp = (n_out + 15) >> 4;
    const uint32_t grp0 = (P.row0 >> 4) + tile * 2u;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n_out * 4u), 0x00020000);
    constexpr uint32_t FR = 256u;
    const uint32_t ntile_co = (uint32_t)P.ntile_co;
    const uint32_t blk_stride = ntile_co * FR;                       // floats between chunk blocks of one tap
    const uint32_t tap_stride = (uint32_t)(NCH * NH) * blk_stride;   // floats between taps
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;

    // ---- active taps of the tile = union of its two groups' masks (SGPRs); per group: which taps ITS rows may read (sparse tables)
    uint64_t m0lo = 0, m0hi = 0, m1lo = 0, m1hi = 0;
    {
        const int K = P.K;
        const uint64_t alo = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
        const uint64_t ahi = K > 64 ? (K >= 128 ? ~0ull : ((1ull << (K - 64)) - 1ull)) : 0ull;
        if (P.mask16) {
            const uint32_t* mp = P.mask16 + (size_t)(grp0 < ngrp ? grp0 : ngrp - 1) * 4;
            const uint32_t a0 = __builtin_amdgcn_readfirstlane(mp[0]), a1 = __builtin_amdgcn_readfirstlane(mp[1]);
            const uint32_t a2 = __builtin_amdgcn_readfirstlane(mp[2]), a3 = __builtin_amdgcn_readfirstlane(mp[3]);
            const uint32_t* mq = P.mask16 + (size_t)(grp0 + 1 < ngrp ? grp0 + 1 : ngrp - 1) * 4;
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(mq[0]), b1 = __builtin_amdgcn_readfirstlane(mq[1]);
            const uint32_t b2 = __builtin_amdgcn_readfirstlane(mq[2]), b3 = __builtin_amdgcn_readfirstlane(mq[3]);
            if (grp0 < ngrp) { m0lo = ((uint64_t)a1 << 32) | a0; m0hi = ((uint64_t)a3 << 32) | a2; }
            if (grp0 + 1 < ngrp) { m1lo = ((uint64_t)b1 << 32) | b0; m1hi = ((uint64_t)b3 << 32) | b2; }
        } else {
            if (grp0 < ngrp) { m0lo = alo; m0hi = ahi; }
            if (grp0 + 1 < ngrp) { m1lo = alo; m1hi = ahi; }
        }
    }
    uint64_t tlo = m0lo | m1lo, thi = m0hi | m1hi;
    const int nt = __builtin_popcountll(tlo) + __builtin_popcountll(thi);

    // this wave stages rows 8 w .. 8 w + 7 of the tile: instruction i covers rows 8 w + RPI i .. + RPI - 1, lane = (row in it, piece)
    const uint32_t sub = (uint32_t)lane / LPR, piece = (uint32_t)lane % LPR;   // piece: 16-byte piece of the staged row part
    // index vector: lane r < 32 holds the neighbour of tile row r under a tap (-1: none; entries outside the row's group mask are unwritten)
    const uint32_t my_row = row_base + (uint32_t)(lane & 31);
    const uint32_t idx_off = (my_row < n_out ? my_row : n_out - 1) * 4u;
    const bool upper = (lane & 31) >= 16;
    auto load_idx = [&](int k) -> uint32_t {
        // (readfirstlane: carried around the loop hipcc keeps the tap id in a VGPR and wraps the load in a waterfall loop)
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, idx_off, (uint32_t)__builtin_amdgcn_readfirstlane(k) * n_out * 4u, 0);
    };
    auto fix_idx = [&](uint32_t raw, int k) -> uint32_t {   // rows beyond the table / of a group that lacks the tap: no neighbour
        const uint64_t wlo = upper ? m1lo : m0lo, whi = upper ? m1hi : m0hi;
        const bool has = k < 64 ? ((wlo >> k) & 1ull) != 0 : ((whi >> (k - 64)) & 1ull) != 0;
        return (has && my_row < n_out) ? raw : 0xFFFFFFFFu;
    };

    // accumulators: [chunk class][channel tile of the wave][row group]
    f32x4 acc[4][COTW][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int it = 0; it < COTW; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) acc[q][it][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t woffv[COTW];   // lanes whose output channel lies beyond Cout read zeros from past the end of the buffer (spconv.hip)
#pragma unroll
    for (int it = 0; it < COTW; ++it) {
        const uint32_t t_ = w * COTW + it;
        const uint32_t co = t_ * 16u + (uint32_t)(lane & 15);
        woffv[it] = co < cout ? (t_ * FR + (uint32_t)lane * 4u) * 4u : 0x7FFFFFF0u;
    }
    // stage addresses: write = (row, piece) of the gathered lanes; read = B fragment of (row group jt, chunk c): row 16 jt + j, piece 4 c + g
    auto swz = [](uint32_t row, uint32_t p) -> uint32_t { return row * (uint32_t)RB + ((p ^ (row & 15u)) << 4); };

    if (nt > 0) {
        // step = (tap, half); the tap ring: kA = tap being computed, kB = next, kC = the one after (indices in flight)
        int kA = wide_pop_or_keep(tlo, thi, 0);
        int kB = wide_pop_or_keep(tlo, thi, kA);
        int kC = wide_pop_or_keep(tlo, thi, kB);
        uint32_t ixA = fix_idx(load_idx(kA), kA);
        uint32_t ixB = load_idx(kB);          // (raw: fixed when it becomes the current tap -- a just-loaded register is not touched)
        uint32_t ixC = load_idx(kC);
        f32x4 gr[NG];                         // gathered row pieces of the NEXT step
        auto gather = [&](uint32_t ix, int half) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const uint32_t r = 8u * w + (uint32_t)(RPI * i) + sub;                      // tile row of this lane
                const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r * 4u), (int)ix);
                const uint32_t off = nb == 0xFFFFFFFFu ? 0x7FFFFFF0u : nb * ld4 + (uint32_t)half * (uint32_t)RB + piece * 16u;
                gr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            }
        };
        gather(ixA, 0);
        const int nsteps = nt * NH;
        int half = 0;
        f32x4 wr[4][COTW];                    // weight fragments of four consecutive chunks (ring slot = chunk % 4)
        auto load_w = [&](int slot, int k, int c) {
            const uint32_t sw = ((uint32_t)__builtin_amdgcn_readfirstlane(k) * tap_stride + (uint32_t)c * blk_stride) * 4u;
#pragma unroll
            for (int it = 0; it < COTW; ++it)
                wr[slot][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw, 0));
        };
        // weights of the first three chunks of the first step
        load_w(0, kA, 0); load_w(1, kA, 1); load_w(2, kA, 2);
        for (int st = 0; st < nsteps; ++st) {
            unsigned char* const sb = &stage[st & 1][0];
            // ---- a. the step's rows (gathered during the step before) go to the stage buffer
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const uint32_t r = 8u * w + (uint32_t)(RPI * i) + sub;
                *(f32x4*)(sb + swz(r, piece)) = gr[i];
            }
            // ---- b. gathers of the next step (the second half of this tap, or the next tap); its successor's indices
            const bool last_half = half == NH - 1;
            const int cbase = half * NCH;                 // first chunk of this step inside the tap's contraction
            const int kcur = kA;
            if (last_half) {
                ixA = fix_idx(ixB, kB);
                kA = kB; kB = kC; ixB = ixC;
                kC = wide_pop_or_keep(tlo, thi, kC);
                ixC = load_idx(kC);
                gather(ixA, 0);
            } else {
                gather(ixA, half + 1);
            }
            const int knext = kA;                          // tap of the next step (== kcur inside a tap)
            const int cnext = last_half ? 0 : cbase + NCH; // its first chunk
            __syncthreads();
            // ---- c. the chunk loop: B fragments from the stage, weights through the ring (three chunks ahead, across the step boundary)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + 3 < NCH) load_w((c + 3) & 3, kcur, cbase + c + 3);
                else load_w((c + 3) & 3, knext, cnext + (c + 3 - NCH));
                f32x4 b0 = *(const f32x4*)(sb + swz((uint32_t)j, (uint32_t)(4 * c + g)));
                f32x4 b1 = *(const f32x4*)(sb + swz(16u + (uint32_t)j, (uint32_t)(4 * c + g)));
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int it = 0; it < COTW; ++it) {
                        // (chunk class = chunk index inside the tap's contraction mod 4: cbase is a multiple of 4)
                        acc[c & 3][it][0] = MFMA(wr[c & 3][it][s], b0[s], acc[c & 3][it][0]);
                        acc[c & 3][it][1] = MFMA(wr[c & 3][it][s], b1[s], acc[c & 3][it][1]);
                    }
            }
            half = last_half ? 0 : half + 1;
        }
    }

    // ---- epilogue: the four class chains meet in the chunk-split tiles' order; lane (g, j) holds channels co0 .. co0 + 3 of its row
#pragma unroll
    for (int it = 0; it < COTW; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const uint32_t co0 = (w * COTW + it) * 16u + 4u * (uint32_t)g;
            const uint32_t o = row_base + 16u * jt + (uint32_t)j;
            if (o >= n_out || co0 >= cout) continue;
            f32x4 v = acc[0][it][jt];
            v += acc[1][it][jt];
            v += acc[2][it][jt];
            v += acc[3][it][jt];
            v += *(const f32x4*)(P.bias + co0);
            if (P.relu_pre) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (P.res_mode == 1) {
                const float* rp = P.res + (size_t)o * P.ld_res + co0;
                if (P.vec_store && co0 + 3 < cout) {
                    v += *(const f32x4*)rp;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co0 + e < cout) v[e] += rp[e];
                }
            } else if (P.res_mode == 2) {
                const float* rp = P.res + (size_t)o * P.ld_res + 2 * co0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co0 + e < cout) v[e] += rp[2 * e] + rp[2 * e + 1];
            }
            if (P.relu_post) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            float* op = P.out + (size_t)o * P.ld_out + co0;
            if (P.vec_store && co0 + 3 < cout) {
                *(f32x4*)op = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co0 + e < cout) op[e] = v[e];
            }
        }
}

std::atomic<int> g_wide_dbg{-1};   // insmos_debug_conv_wide (test hook): -1 = INSMOS_CONV_WIDE (default on), 0 = off, 1 = on

}  // namespace

// the staged 32-row kernel for a layer the dispatcher would run on chunk-split tiles, or null: Cin in {64, 128, 256} read as whole
// rows (the input view starts a row: 16-byte aligned pieces), Cout in {64, 128}, a neighbour table
ConvKernelFn conv_wide_pick(const ConvP& P, long* blocks) {
    static const int env = [] { const char* e = getenv("INSMOS_CONV_WIDE"); return e ? atoi(e) : 1; }();
    const int dbg = g_wide_dbg.load(std::memory_order_relaxed);
    if (!(dbg >= 0 ? dbg : env)) return nullptr;
    if (!P.nbr || P.has8 || P.has4 || (P.n16 != 4 && P.n16 != 8 && P.n16 != 16) || (P.ntile_co != 4 && P.ntile_co != 8) || P.K > 128 ||
        (P.row0 & 15u))
        return nullptr;
    const long rows = (long)P.n_out - (long)P.row0;
    *blocks = (rows + 31) / 32;
    const bool c2 = P.ntile_co == 8;
    if (P.n16 == 4) return c2 ? k_conv_wide<4, 1, 2> : k_conv_wide<4, 1, 1>;
    if (P.n16 == 8) return c2 ? k_conv_wide<8, 1, 2> : k_conv_wide<8, 1, 1>;
    return c2 ? k_conv_wide<8, 2, 2> : k_conv_wide<8, 2, 1>;
}

}  // namespace insmos

extern "C" int insmos_debug_conv_wide(int on) {
    if (on < -1 || on > 1) return INSMOS_EINVAL;
    insmos::g_wide_dbg.store(on, std::memory_order_relaxed);
    return INSMOS_OK;
}
