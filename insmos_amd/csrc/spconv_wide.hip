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
//
// STATUS (round 6, measured: profiles/r06_wide_layers_ab.csv, r06_wide_probe_layers.csv): OPT-IN (INSMOS_CONV_WIDE=1), the chunk-split
// tiles stay the default.  Same bits, but 1.2-1.6x SLOWER on the 64 / 128-channel layers (conv4.1.0: 217 us split, 303 us here).  The
// probe builds (-DINSMOS_WIDE_PROBE_BUILD, INSMOS_WIDE_PROBE) say where the time is: without gathers AND without weight loads the kernel still needs 230 us, and
// without MFMAs 16 us -- the load side this kernel improves (half the fragment traffic, whole-line gathers) was not the limit of these
// layers; what the kernel adds -- one barrier per tap for five waves, 1 296 heavy workgroups (three resident per CU) for 256 CUs
// where the split tiles have 2 592 light ones -- costs MFMA issue slots.  The split tiles already run these layers at 94 TFLOP/s
// useful x 1.14 issued / useful (regrouped row order, DESIGN.md 3.5: 17.5 active slots per group for 15.3 valid taps per row)
// = 108 TFLOP/s executed = 0.78 of the fp32 MFMA rate the chip sustains (139 TFLOP/s).  (tools/wide_stats.py prints the issued /
// useful ratios of the GENERATED row order -- the step path does not regroup: 1.41 at level 4, 1.54 at level 3.)
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
// COTW: channel tiles per CONSUMER wave (Cout = 64 COTW)
//
// FIVE waves: waves 0..3 CONSUME (weight fragments + MFMAs), wave 4 PRODUCES (neighbour indices, whole-row gathers, stage writes).
// The first build had every wave do both, and lost 30 % to the chunk-split tiles: a wave's vector-memory results return IN ORDER, so a
// weight fragment requested behind the next tap's gathers (HBM latency) cannot arrive before them -- the chunk loop stalled on every
// tap's gathers (ISA: s_waitcnt vmcnt(0) in front of the MFMAs).  With the gathers in a wave of their own the consumers' queues hold
// weight fragments only (L1 / L2 hits), and the producer has a whole step (~4 000 cycles of MFMA work per consumer) to cover the
// gather latency.  (The four consumers of a workgroup land on the four SIMDs, the producer -- the LAST wave -- beside one of them.)
template <int NCH, int NH, int COTW, int PROBE = 0>
__global__ void __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(4, 4))) k_conv_wide(ConvP P) {   // (<= 128 VGPRs: three workgroups per CU)
    static_assert(NCH == 4 || NCH == 8, "staged row pieces of 256 or 512 bytes");
    constexpr int RB = NCH * 64;               // staged bytes per row
    constexpr int RPI = 1024 / RB;             // rows per gather instruction (2 or 4)
    constexpr int NG = 32 / RPI;               // gather instructions per step (the producer stages all 32 rows)
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
    const int nsteps = nt * NH;
    auto swz = [](uint32_t row, uint32_t p) -> uint32_t { return row * (uint32_t)RB + ((p ^ (row & 15u)) << 4); };

    if (w == 4) {
        // ================================ producer ================================
        if (nt == 0) return;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_nb =
            __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n_out * 4u), 0x00020000);
        const uint32_t ld4 = (uint32_t)P.ld_in * 4u;
        const uint32_t sub = (uint32_t)lane / LPR, piece = (uint32_t)lane % LPR;   // instruction i: tile rows RPI i .. RPI i + RPI - 1
        // index vector: lane r < 32 holds the neighbour of tile row r under a tap
        const uint32_t my_row = row_base + (uint32_t)(lane & 31);
        const uint32_t idx_off = (my_row < n_out ? my_row : n_out - 1) * 4u;
        const bool upper = (lane & 31) >= 16;
        auto load_idx = [&](int k) -> uint32_t {
            // (readfirstlane: carried around the loop hipcc keeps the tap id in a VGPR and wraps the load in a waterfall loop)
            return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, idx_off, (uint32_t)__builtin_amdgcn_readfirstlane(k) * n_out * 4u, 0);
        };
        // rows beyond the table / of a group that lacks the tap (entries outside a group's mask are unwritten memory): no neighbour.
        // (scalar bit tests, then ONE per-lane select: a per-lane select between the 64-bit masks made hipcc build a table in scratch,
        // and every scratch access shares vmcnt with the gathers -- the loop waited for its own prefetch)
        auto fix_idx = [&](uint32_t raw, int k) -> uint32_t {
            const uint32_t ks = (uint32_t)__builtin_amdgcn_readfirstlane(k);
            const uint32_t sh = ks & 63u;
            const uint32_t h0 = (uint32_t)((ks < 64u ? m0lo : m0hi) >> sh) & 1u;
            const uint32_t h1 = (uint32_t)((ks < 64u ? m1lo : m1hi) >> sh) & 1u;
            const uint32_t has = upper ? h1 : h0;
            return (has != 0u && my_row < n_out) ? raw : 0xFFFFFFFFu;
        };
        int kA = wide_pop_or_keep(tlo, thi, 0);      // tap of the step whose rows are gathered next
        int kB = wide_pop_or_keep(tlo, thi, kA);
        uint32_t ixA = fix_idx(load_idx(kA), kA);
        uint32_t ixB = load_idx(kB);                 // (raw: masked when it becomes the current tap)
        int half = 0;
        f32x4 gr[NG];
        auto gather = [&]() {                        // the rows of (kA, half), then on to the next step
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const uint32_t r = (uint32_t)(RPI * i) + sub;
                const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r * 4u), (int)ixA);
                const uint32_t off = (nb == 0xFFFFFFFFu || (PROBE & 1)) ? 0x7FFFFFF0u : nb * ld4 + (uint32_t)half * (uint32_t)RB + piece * 16u;
                gr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            }
            if (half == NH - 1) {
                half = 0;
                ixA = fix_idx(ixB, kB);
                kA = kB;
                kB = wide_pop_or_keep(tlo, thi, kB);
                ixB = load_idx(kB);
            } else {
                ++half;
            }
        };
        auto put = [&](int buf) {
            unsigned char* const sb = &stage[buf][0];
#pragma unroll
            for (int i = 0; i < NG; ++i) *(f32x4*)(sb + swz((uint32_t)(RPI * i) + sub, piece)) = gr[i];
        };
        gather();            // step 0
        put(0);
        gather();            // step 1 (past the end: the last tap again, harmless)
        __syncthreads();     // B_0: stage[0] holds step 0
        for (int st = 0; st < nsteps; ++st) {
            put((st + 1) & 1);   // (the consumers read stage[st & 1] now; they left the other buffer before B_st)
            gather();            // step st + 2
            __syncthreads();     // B_{st + 1}
        }
        return;
    }

    // ================================ consumers ================================
    constexpr uint32_t FR = 256u;
    const uint32_t ntile_co = (uint32_t)P.ntile_co;
    const uint32_t blk_stride = ntile_co * FR;                       // floats between chunk blocks of one tap
    const uint32_t tap_stride = (uint32_t)(NCH * NH) * blk_stride;   // floats between taps
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
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
    if (nt > 0) {
        int kA = wide_pop_or_keep(tlo, thi, 0);      // tap of the current step, kB of the next tap
        int kB = wide_pop_or_keep(tlo, thi, kA);
        int half = 0;
        f32x4 wr[4][COTW];                           // weight fragments of four consecutive chunks (ring slot = chunk % 4)
        if (PROBE & 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int it = 0; it < COTW; ++it) wr[q][it] = (f32x4){(float)lane, 1.f, 2.f, (float)w};
        }
        auto load_w = [&](int slot, int k, int c) {
            const uint32_t sw = ((uint32_t)__builtin_amdgcn_readfirstlane(k) * tap_stride + (uint32_t)c * blk_stride) * 4u;
#pragma unroll
            for (int it = 0; it < COTW; ++it)
                if (!(PROBE & 4)) wr[slot][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw, 0));
        };
        load_w(0, kA, 0); load_w(1, kA, 1); load_w(2, kA, 2);
        __syncthreads();     // B_0
        for (int st = 0; st < nsteps; ++st) {
            const unsigned char* const sb = &stage[st & 1][0];
            const bool last_half = half == NH - 1;
            const int cbase = half * NCH;                 // first chunk of this step inside the tap's contraction
            const int kcur = kA;
            if (last_half) { kA = kB; kB = wide_pop_or_keep(tlo, thi, kB); }
            const int knext = kA;                          // tap of the next step (== kcur inside a tap)
            const int cnext = last_half ? 0 : cbase + NCH; // its first chunk
            // which of the two row groups has the tap: the other one's MFMAs would add W x 0 (the chunk-split tiles skip them, and a
            // 32-row tile's tap union is up to 1.5x one group's list) -- scalar tests, one of three loop bodies
            const uint32_t ks = (uint32_t)__builtin_amdgcn_readfirstlane(kcur), sh = ks & 63u;
            const bool do0 = !(PROBE & 2) && (((ks < 64u ? m0lo : m0hi) >> sh) & 1ull) != 0;
            const bool do1 = !(PROBE & 2) && (((ks < 64u ? m1lo : m1hi) >> sh) & 1ull) != 0;
            // the chunk loop: B fragments from the stage, weights through the ring (three chunks ahead, across the step boundary);
            // ONE body with uniform branches around each group's MFMAs (three specialised bodies cost 70 more VGPRs)
            // (B fragments one chunk ahead and unconditionally: a read behind the branch is waited for on the spot)
            f32x4 b0 = *(const f32x4*)(sb + swz((uint32_t)j, (uint32_t)g));
            f32x4 b1 = *(const f32x4*)(sb + swz(16u + (uint32_t)j, (uint32_t)g));
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + 3 < NCH) load_w((c + 3) & 3, kcur, cbase + c + 3);
                else load_w((c + 3) & 3, knext, cnext + (c + 3 - NCH));
                f32x4 n0 = b0, n1 = b1;
                if (c + 1 < NCH) {
                    n0 = *(const f32x4*)(sb + swz((uint32_t)j, (uint32_t)(4 * (c + 1) + g)));
                    n1 = *(const f32x4*)(sb + swz(16u + (uint32_t)j, (uint32_t)(4 * (c + 1) + g)));
                }
                if (do0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int it = 0; it < COTW; ++it)   // (chunk class = chunk index inside the tap's contraction mod 4)
                            acc[c & 3][it][0] = MFMA(wr[c & 3][it][s], b0[s], acc[c & 3][it][0]);
                }
                if (do1) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int it = 0; it < COTW; ++it)
                            acc[c & 3][it][1] = MFMA(wr[c & 3][it][s], b1[s], acc[c & 3][it][1]);
                }
                b0 = n0; b1 = n1;
            }
            half = last_half ? 0 : half + 1;
            __syncthreads();     // B_{st + 1}
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

std::atomic<int> g_wide_dbg{-1};   // insmos_debug_conv_wide (test hook): -1 = INSMOS_CONV_WIDE (default off), 0 = off, 1 = on

}  // namespace

// the staged 32-row kernel for a layer the dispatcher would run on chunk-split tiles, or null: Cin in {64, 128, 256} read as whole
// rows (the input view starts a row: 16-byte aligned pieces), Cout in {64, 128}, a neighbour table
ConvKernelFn conv_wide_pick(const ConvP& P, long* blocks) {
    const int env = [] { const char* e = getenv("INSMOS_CONV_WIDE"); return e ? atoi(e) : 0; }();   // (per call: A/B tools flip it) default OFF: see the header
    const int dbg = g_wide_dbg.load(std::memory_order_relaxed);
    if (!(dbg >= 0 ? dbg : env)) return nullptr;
    if (!P.nbr || P.has8 || P.has4 || (P.n16 != 4 && P.n16 != 8 && P.n16 != 16) || (P.ntile_co != 4 && P.ntile_co != 8) || P.K > 128 ||
        (P.row0 & 15u))
        return nullptr;
    const long rows = (long)P.n_out - (long)P.row0;
    *blocks = (rows + 31) / 32;
    const bool c2 = P.ntile_co == 8;
    if (P.n16 == 4) return c2 ? k_conv_wide<4, 1, 2> : k_conv_wide<4, 1, 1>;
#ifdef INSMOS_WIDE_PROBE_BUILD   // (not in the product library: add -DINSMOS_WIDE_PROBE_BUILD to FLAGS of __graft_entry__.py for tools/r06_wide_probe.sh)
    if (P.n16 == 8 && c2) {   // PROBE builds (timing only, wrong results): 1 = no gathers, 2 = no MFMAs, 4 = no weight loads
        const char* e = getenv("INSMOS_WIDE_PROBE");
        switch (e ? atoi(e) : 0) {
            case 1: return k_conv_wide<8, 1, 2, 1>;
            case 2: return k_conv_wide<8, 1, 2, 2>;
            case 3: return k_conv_wide<8, 1, 2, 3>;
            case 4: return k_conv_wide<8, 1, 2, 4>;
            case 5: return k_conv_wide<8, 1, 2, 5>;
            case 6: return k_conv_wide<8, 1, 2, 6>;
            default: break;
        }
    }
#endif
    if (P.n16 == 8) return c2 ? k_conv_wide<8, 1, 2> : k_conv_wide<8, 1, 1>;
    return c2 ? k_conv_wide<8, 2, 2> : k_conv_wide<8, 2, 1>;
}

}  // namespace insmos

extern "C" int insmos_debug_conv_wide(int on) {
    if (on < -1 || on > 1) return INSMOS_EINVAL;
    insmos::g_wide_dbg.store(on, std::memory_order_relaxed);
    return INSMOS_OK;
}
