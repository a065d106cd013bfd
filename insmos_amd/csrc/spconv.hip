// insmos_amd/csrc/spconv.hip -- output-stationary sparse convolution on the CDNA4 matrix cores.
//
//   out[o, :] = epilogue( sum_k  in[nbr[k][o], :] @ W[k]  + bias )
//
// One wave owns 16*JT consecutive output rows x 16*COT output channels (tile shape picked per layer, see
// insmos_sparse_conv).  Taps are walked through a per-16-row-group ACTIVE-TAP BITMASK produced when the neighbour
// table is built: a tap none of the tile's rows uses is never touched (rows are Morton-/raster-ordered, so occupancy is
// spatially coherent: ~60 % of (16-row group, tap) slots are empty on LiDAR data).  Because the active-tap list is known
// up front the loop is software-pipelined: neighbour indices are fetched two taps ahead, the gathered rows (B
// fragments: one 16-byte load per lane per 16-channel chunk; four 16-lane groups cover one 64-byte sector of a row)
// and the pre-packed weight A fragments (one coalesced 16-byte load per lane, L1/L2 resident) one or two work items
// ahead of the MFMAs that consume them.  All offsets are 32-bit so loads use SGPR-base + VGPR-offset addressing.  The
// contraction runs on v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fmaf chain), i = output channel, j = output row,
// so lane (g, j) ends up with 4 consecutive channels of row j and the epilogue (folded-BN bias, ReLU, residual /
// channel-pair residual) is one float4 store per tile.  No atomics, no scatter: results are deterministic.
//
// Fragment maps (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[i = 4*(lane>>4) + reg][j = lane&15].  The contraction index of MFMA
// step s inside a 16-channel chunk is channel c0 + 4*(lane>>4) + s (any assignment is legal as long
// as A and B agree), which is what makes the gather a contiguous float4 per lane.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "conv_common.h"
#include "prec.h"

namespace insmos {

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
//
// SPLIT > 1: the tile covers ALL its channel tiles in one wave and the SPLIT consecutive waves of a block
// share the tile's row group and split its contraction -- by 16-channel chunk when SPLITC (Cin/16 is a
// multiple of SPLIT: perfectly even), else by every SPLIT-th active tap -- so each input row chunk is
// gathered exactly once per tile (instead of once per channel group); the partial accumulators are summed
// through LDS in fixed wave order (deterministic) in the epilogue.
//
// LOAD BALANCE.  Active-tap counts vary 4x between tiles on LiDAR data (tools/conv_probe.py: the same launch with
// the same total work spread evenly over the tiles runs 1.2-1.6x faster), and mid-size layers have only a few tiles
// per SIMD.  Two launch-shape rules follow (measured; a device-side atomic tile queue and a residency cap that
// would let the dispatcher re-balance were both slower):
//  * unsplit tiles run as ONE-WAVE workgroups, so a finished wave's slot is refilled at once instead of waiting for
//    the slowest of four waves in its block;
//  * masked layers with enough work per tile are SPLIT (above): 4x the waves, each a quarter as long.
// DBG (probe builds only, tools/conv_probe.py): bit 0 = no weight loads, bit 1 = no gathers, bit 2 = no MFMAs
template <int COT, int JT, int CK, bool IDENT, int R, int SPLIT, bool SPLITC, int DBG = 0, int PREC = 0>
__global__ void __launch_bounds__(SPLIT == 1 ? 64 : 256) k_sparse_conv(ConvP P) {
    static_assert(PREC == 0 || (CK == 0 && DBG == 0), "reduced-precision variants exist for 16-channel chunks only");
    static_assert(SPLIT == 1 || (JT == 1 && (COT % SPLIT == 0 || SPLIT % COT == 0)), "split tiles are 16 rows x all channels");
    static_assert(!SPLITC || (SPLIT > 1 && CK == 0), "chunk split needs 16-channel chunks");
    const int lane = threadIdx.x & 63;
    // readfirstlane: tell the compiler the wave id (hence every tile-level quantity) is wave-uniform
    const uint32_t wib = SPLIT == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t TPB = 4 / SPLIT;  // tiles per block (SPLIT > 1); unsplit tiles are one-wave blocks
    const uint32_t n_cg = P.ntile_co / COT;
    const uint32_t n_tiles = (uint32_t)P.n_otiles * n_cg;
    const uint32_t ws = wib % SPLIT;
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;  // row pitch in bytes

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb =
        __builtin_amdgcn_make_buffer_rsrc((void*)(IDENT ? (const void*)P.in : (const void*)P.nbr), 0,
                                          IDENT ? 4 : (int)((uint32_t)P.K * n_out * 4u), 0x00020000);
    constexpr uint32_t LW = CK == 4 ? 4u : CK == 8 ? 8u : 16u;  // bytes per lane per gather
    const uint32_t goff = (uint32_t)g * LW;
    // A fragments: 4 floats per lane per 16-channel chunk; the single-chunk layers (Cin = 8 / 4) pack 2 / 1 floats per lane
    // (their other MFMA steps would multiply packed zeros: half / three quarters of the weight bytes these L2-bound layers fetch)
    constexpr uint32_t LWF = CK == 4 ? 1u : CK == 8 ? 2u : 4u;  // floats per lane per fragment
    constexpr uint32_t FR = 64u * LWF;                          // floats per fragment
    const uint32_t blk_stride = (uint32_t)P.ntile_co * FR;      // floats between chunk blocks of one tap
    const uint32_t tap_stride = (uint32_t)P.nblk * blk_stride;  // floats between taps
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    constexpr int NS = CK ? CK / 4 : 4;  // MFMA steps per chunk
    const int nchunk = CK ? 1 : P.n16;
    constexpr int CSTEP = SPLITC ? SPLIT : 1;
    const int c0 = SPLITC ? (int)ws : 0;
    const uint32_t cout = P.cout;
    {
        // (workgroup b runs on XCD b % 8: consecutive tiles are dealt round-robin over the eight L2s.  Handing each XCD a
        //  contiguous run of tiles instead was measured: +4...29 % per layer -- the work per tile varies smoothly along the
        //  row order, so contiguous runs unbalance the XCDs; round-robin is also the better load balancer)
        // (round 6: runs of S = 4 ... 256 consecutive units per XCD instead -- unit = ((b / 8 / S) * 8 + b % 8) * S + b / 8 % S, so that
        //  neighbouring tiles' gathers meet in ONE L2 while the XCDs stay balanced -- measured per layer, interleaved in one process:
        //  +-1 % on every layer, +2...30 % on the inverse maps at S = 16 / 64: the L2 is not what these kernels wait for.  Removed.)
        const uint32_t unit = blockIdx.x;
        const uint32_t tile_raw = SPLIT == 1 ? unit : unit * TPB + wib / SPLIT;
        const bool live = tile_raw < n_tiles;  // only split blocks can hold a dead tile (kept for the barriers)
        const uint32_t tile = live ? tile_raw : n_tiles - 1;
        const uint32_t cg = tile / P.n_otiles;
        const uint32_t ot = tile % P.n_otiles;

        uint32_t orow[JT], rowoff[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            orow[jt] = P.row0 + ot * (16 * JT) + jt * 16 + j;
            rowoff[jt] = (orow[jt] < n_out ? orow[jt] : n_out - 1) * 4u;  // byte offset inside one tap row of nbr
        }

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
                    const uint32_t grp = (P.row0 >> 4) + ot * JT + jt;
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
        if constexpr (SPLIT > 1 && !SPLITC) {
            static_assert(SPLIT == 4, "residue masks below are written for 4 waves per tile");
            if (P.tap_mod) {
                const uint64_t mine = 0x1111111111111111ull << ws;  // taps k with k % 4 == ws (64 % 4 == 0: same in both words)
                tlo &= mine;
                thi &= mine;
                nt = live ? __builtin_popcountll(tlo) + __builtin_popcountll(thi) : 0;
            } else {
                for (uint32_t q = 0; q < ws; ++q) (void)pop_or_keep(tlo, thi, 0);  // my taps: ranks ws, ws+SPLIT, ...
                nt = (live && nt > (int)ws) ? (nt - (int)ws + SPLIT - 1) / SPLIT : 0;
            }
        } else if constexpr (SPLIT > 1) {
            nt = live ? nt : 0;
        }
        // next tap of THIS wave (a tap-split wave skips the taps owned by the other waves of the tile)
        auto next_tap = [&](int keep) {
            const int k = pop_or_keep(tlo, thi, keep);
            if constexpr (SPLIT > 1 && !SPLITC) {
                if (!P.tap_mod) {
#pragma unroll
                    for (int q = 1; q < SPLIT; ++q) (void)pop_or_keep(tlo, thi, 0);
                }
            }
            return k;
        };

        f32x4 acc[COT][JT];
#pragma unroll
        for (int it = 0; it < COT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc[it][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // per-lane byte offset of this lane's A fragment in each channel tile.  Lanes whose output channel (lane & 15)
        // lies beyond Cout hold packed ZEROS: point them past the end of the buffer instead, the hardware returns the
        // same zeros without touching the L1 (Cout = 8 layers: half of every weight fragment)
        uint32_t woffv[COT];
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            const uint32_t co = (cg * COT + it) * 16u + (uint32_t)(lane & 15);
            woffv[it] = co < cout ? ((cg * COT + it) * FR + lane * LWF) * 4u : 0x7FFFFFF0u;
        }

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
                if constexpr (DBG & 2) {
                    if (k == 1000) b[jt] = (f32x4){1.f, 1.f, 1.f, 1.f};
                } else if constexpr (CK == 4) {
                    b[jt] = (f32x4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off[jt], so, 0)),
                                    0.f, 0.f, 0.f};
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
            for (int it = 0; it < COT; ++it) {
                if constexpr (DBG & 1) {
                    if (k == 1000) a[it] = (f32x4){1.f, 1.f, 1.f, 1.f};
                } else if constexpr (CK == 4) {
                    a[it] = (f32x4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, woffv[it], sw, 0)), 0.f, 0.f, 0.f};
                } else if constexpr (CK == 8) {
                    f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, woffv[it], sw, 0));
                    a[it] = (f32x4){t[0], t[1], 0.f, 0.f};
                } else {
                    a[it] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw, 0));
                }
            }
        };
        auto mma = [&](const f32x4 (&a)[COT], const f32x4 (&b)[JT]) {
            if constexpr (DBG & 4) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) asm volatile("" ::"v"(b[jt]));
#pragma unroll
                for (int it = 0; it < COT; ++it) asm volatile("" ::"v"(a[it]));
                return;
            }
            if constexpr (PREC == 1) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const s16x4 bh = round_bf16(b[jt]);
#pragma unroll
                    for (int it = 0; it < COT; ++it) acc[it][jt] = MFMA_BF16(round_bf16(a[it]), bh, acc[it][jt]);
                }
                return;
            } else if constexpr (PREC == 3) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    s16x4 bh, bl;
                    split_bf16(b[jt], bh, bl);
#pragma unroll
                    for (int it = 0; it < COT; ++it) {
                        s16x4 ah, al;
                        unpack_split(a[it], ah, al);
                        acc[it][jt] = MFMA_BF16(al, bh, acc[it][jt]);  // small terms first
                        acc[it][jt] = MFMA_BF16(ah, bl, acc[it][jt]);
                        acc[it][jt] = MFMA_BF16(ah, bh, acc[it][jt]);
                    }
                }
                return;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int it = 0; it < COT; ++it) acc[it][jt] = MFMA(a[it][s], b[jt][s], acc[it][jt]);
        };

        if (nt > 0) {
            // ---- load cursor: (kL, cL) = next item to request; tap ring: offL = row offsets of the current tap,
            // offN = those of the next one (already scaled), idxNN = raw indices of the tap after that, in flight.
            // (The ring holds SCALED offsets on purpose: rotating raw loaded indices between variables made hipcc
            // copy the just-issued load at the loop back-edge, i.e. wait vmcnt(0) every iteration.)
            int kL = next_tap(0);
            int kN = next_tap(kL);
            int kNN = next_tap(kN);
            int cL = c0;
            uint32_t offL[JT], offN[JT], idxNN[JT];
            {
                uint32_t idx0[JT], idx1[JT];
                load_idx(kL, idx0);
                load_idx(kN, idx1);
                load_idx(kNN, idxNN);
                row_offsets(idx0, offL);
                row_offsets(idx1, offN);
            }
            f32x4 bs[R][JT], as[R][COT];
            if constexpr (DBG != 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) bs[r][jt] = (f32x4){0.5f, 0.5f, 0.5f, 0.5f};
#pragma unroll
                    for (int it = 0; it < COT; ++it) as[r][it] = (f32x4){0.5f, 0.5f, 0.5f, 0.5f};
                }
            }
            const int nitems = nt * (nchunk / CSTEP);
#define REQUEST(slot)                                                                     \
    {                                                                                     \
        load_ab(kL, cL, offL, as[slot], bs[slot]);                                        \
        if (cL + CSTEP < nchunk) {                                                        \
            cL += CSTEP;                                                                  \
        } else {                                                                          \
            cL = c0;                                                                      \
            kL = kN; kN = kNN;                                                            \
            _Pragma("unroll") for (int jt = 0; jt < JT; ++jt) offL[jt] = offN[jt];        \
            row_offsets(idxNN, offN);                                                     \
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
            __shared__ f32x4 red[4][COT][64];  // [wave in block][channel tile][lane]
#pragma unroll
            for (int it = 0; it < COT; ++it) red[wib][it][lane] = acc[it][0];
            __syncthreads();
            if (live && (COT >= SPLIT || ws < (uint32_t)COT)) {  // (fewer channel tiles than waves: the first COT waves finish)
                const uint32_t w0 = wib - ws;  // first wave of this tile inside the block
                constexpr int PER = COT >= SPLIT ? COT / SPLIT : 1;
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
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-chunk layers (Cin = 8, or 16: one chunk per tap; 16 rows x COT channel tiles per wave) with QUAD INDEX LOADS.
// These layers are bound by the texture-address unit: ~16 cycles per wave-wide vector memory instruction whatever it
// fetches, and k_sparse_conv issues three per tap here (neighbour indices, gather, weight fragment).  The index load is
// the wasteful one: 64 lanes fetch 16 distinct dwords (the four lane groups g read the same 16 rows).  This kernel
// fetches the indices of FOUR taps with one instruction -- lane (g, j) reads row j under the g-th tap of a quad -- and
// hands each tap's 16 indices to all four lane groups with a ds_bpermute (LDS crossbar, no memory traffic): 2.25 vector
// memory instructions per tap instead of 3.
//
// Pipeline (per wave, item = tap): operand ring of 4 slots requested 3 taps ahead; the bpermute of tap n+4 is issued at
// step n and consumed at step n+1; the index quad Q+2 is requested at the first step of quad Q and rotated into place
// after its last (a load issued four steps earlier).  Taps past the end of the tile's list re-request the last tap (clamp,
// no guard).  Per output row the operation order is that of k_sparse_conv<COT, 1, CK, false, *, 1, false>: same bits.
template <int COT, int CK>
__global__ void __launch_bounds__(64) k_sparse_conv_q(ConvP P) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;
    const uint32_t n4 = n_out * 4u;  // bytes per tap row of the neighbour table
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n4), 0x00020000);
    constexpr uint32_t LW = CK == 8 ? 8u : 16u;  // bytes per lane per gather
    constexpr uint32_t LWF = CK == 8 ? 2u : 4u;  // floats per lane per weight fragment
    constexpr uint32_t FR = 64u * LWF;
    const uint32_t goff = (uint32_t)g * LW;
    const uint32_t tap_stride = (uint32_t)P.ntile_co * FR;  // one chunk block per tap
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    constexpr int NS = CK ? CK / 4 : 4;
    const uint32_t cout = P.cout;

    const uint32_t tile = blockIdx.x;
    const uint32_t cg = tile / P.n_otiles;
    const uint32_t ot = tile % P.n_otiles;
    const uint32_t orow = P.row0 + ot * 16 + j;
    const uint32_t rowoff = (orow < n_out ? orow : n_out - 1) * 4u;

    uint64_t tlo, thi;
    if (P.mask16) {
        const uint32_t* mp = P.mask16 + (size_t)((P.row0 >> 4) + ot) * 4;
        const uint32_t w0 = __builtin_amdgcn_readfirstlane(mp[0]), w1 = __builtin_amdgcn_readfirstlane(mp[1]);
        const uint32_t w2 = __builtin_amdgcn_readfirstlane(mp[2]), w3 = __builtin_amdgcn_readfirstlane(mp[3]);
        tlo = ((uint64_t)w1 << 32) | w0;
        thi = ((uint64_t)w3 << 32) | w2;
    } else {
        const int K = P.K;
        tlo = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
        thi = K > 64 ? (K >= 128 ? ~0ull : ((1ull << (K - 64)) - 1ull)) : 0ull;
    }
    const int nt = __builtin_popcountll(tlo) + __builtin_popcountll(thi);

    f32x4 acc[COT];
#pragma unroll
    for (int it = 0; it < COT; ++it) acc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t woffv[COT];  // lanes whose output channel lies beyond Cout read zeros from past the end of the buffer
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co = (cg * COT + it) * 16u + (uint32_t)(lane & 15);
        woffv[it] = co < cout ? ((cg * COT + it) * FR + lane * LWF) * 4u : 0x7FFFFFF0u;
    }

    const uint32_t gm[4] = {g == 0 ? ~0u : 0u, g == 1 ? ~0u : 0u, g == 2 ? ~0u : 0u, g == 3 ? ~0u : 0u};
    auto pop4 = [&](int (&k)[4], int keep) {
        k[0] = pop_or_keep(tlo, thi, keep);
        k[1] = pop_or_keep(tlo, thi, k[0]);
        k[2] = pop_or_keep(tlo, thi, k[1]);
        k[3] = pop_or_keep(tlo, thi, k[2]);
    };
    // lane (g, j): neighbour index of row j under the g-th tap of the quad
    auto load_quad = [&](const int (&k)[4]) -> uint32_t {
        // (bitwise select on lane masks: a ?: chain over k[] made hipcc spill the tap ids to a scratch array indexed by g)
        const uint32_t ksel = ((uint32_t)k[0] & gm[0]) | ((uint32_t)k[1] & gm[1]) | ((uint32_t)k[2] & gm[2]) | ((uint32_t)k[3] & gm[3]);
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, rowoff + ksel * n4, 0, 0);
    };
    const int bp = j * 4;  // ds_bpermute byte address of lane (0, j)
    auto quad_tap = [&](uint32_t iq, int p) -> uint32_t {  // the 16 indices of the quad's p-th tap, to every lane group
        return (uint32_t)__builtin_amdgcn_ds_bpermute(bp + 64 * p, (int)iq);
    };
    auto load_ab = [&](int k, uint32_t idx, f32x4 (&a)[COT], f32x4& b) {
        const uint32_t off = idx * ld4 + goff;  // index -1 wraps past the end of the buffer -> the load returns 0
        if constexpr (CK == 8) {
            f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, off, 0, 0));
            b = (f32x4){t[0], t[1], 0.f, 0.f};
        } else {
            b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
        }
        const uint32_t sw = (uint32_t)k * tap_stride * 4u;
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            if constexpr (CK == 8) {
                f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, woffv[it], sw, 0));
                a[it] = (f32x4){t[0], t[1], 0.f, 0.f};
            } else {
                a[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw, 0));
            }
        }
    };
    auto mma = [&](const f32x4 (&a)[COT], const f32x4& b) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int it = 0; it < COT; ++it) acc[it] = MFMA(a[it][s], b[s], acc[it]);
    };

    if (nt > 0) {
        int kqA[4], kqB[4], kqC[4];  // tap ids of the current quad and the next two (SGPRs)
        pop4(kqA, 0);
        pop4(kqB, kqA[3]);
        pop4(kqC, kqB[3]);
        uint32_t iqB, iqC, pidx;
        f32x4 as[4][COT], bs[4];
        {
            const uint32_t iqA = load_quad(kqA);
            iqB = load_quad(kqB);
            const uint32_t p0 = quad_tap(iqA, 0), p1 = quad_tap(iqA, 1), p2 = quad_tap(iqA, 2);
            pidx = quad_tap(iqA, 3);
            load_ab(kqA[0], p0, as[0], bs[0]);
            load_ab(kqA[1], p1, as[1], bs[1]);
            load_ab(kqA[2], p2, as[2], bs[2]);
        }
        // one quad of steps: bpermutes read `src` (quad Q+1), the load of quad Q+2 lands in `dst`.  The two index registers
        // swap roles every quad (no copy of a just-loaded register: that would wait for the load at the loop back-edge)
        auto quad = [&](const uint32_t& src, uint32_t& dst) {
            dst = load_quad(kqC);  // (tap ids popped one quad earlier: no scalar chain in front of the load)
            __builtin_amdgcn_sched_barrier(0);
            int kqD[4];
            pop4(kqD, kqC[3]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                load_ab(s == 0 ? kqA[3] : kqB[s - 1], pidx, as[(s + 3) & 3], bs[(s + 3) & 3]);  // tap i + s + 3
                pidx = quad_tap(src, s);                                                          // tap i + s + 4
                mma(as[s], bs[s]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { kqA[q] = kqB[q]; kqB[q] = kqC[q]; kqC[q] = kqD[q]; }
        };
        int i = 0;
        for (; i + 8 <= nt; i += 8) {
            quad(iqB, iqC);
            quad(iqC, iqB);
        }
        if (i + 4 <= nt) {
            quad(iqB, iqC);
            i += 4;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (i + s < nt) mma(as[s], bs[s]);  // wave-uniform tail, at most 3 taps (operands already in flight)
        }
    }

    // ---- epilogue: lane (g, j) holds channels co0..co0+3 of row orow (same as k_sparse_conv's)
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co0 = (cg * COT + it) * 16 + 4 * g;
        if (orow >= n_out || co0 >= cout) continue;
        f32x4 v = acc[it];
        v += *(const f32x4*)(P.bias + co0);
        if (P.relu_pre) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (P.res_mode == 1) {
            const float* rp = P.res + (size_t)orow * P.ld_res + co0;
            if (P.vec_store && co0 + 3 < cout) {
                v += *(const f32x4*)rp;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < cout) v[r] += rp[r];
            }
        } else if (P.res_mode == 2) {
            const float* rp = P.res + (size_t)orow * P.ld_res + 2 * co0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co0 + r < cout) v[r] += rp[2 * r] + rp[2 * r + 1];
        }
        if (P.relu_post) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        float* op = P.out + (size_t)orow * P.ld_out + co0;
        if (P.vec_store && co0 + 3 < cout) {
            *(f32x4*)op = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co0 + r < cout) op[r] = v[r];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused ConvTranspose2d(k=2,s=2)+BN+ReLU and the 1x1 heads (base_bev_backbone.py:104-115 deblocks, center_head.py:65-72).
// The deconv is a 1x1 conv to 4 sub-sites x CUP channels; its output is read ONLY by the heads, so it never has to exist
// in memory: a wave owns 16 BEV sites x one sub-site, keeps the CUP = 16*NT deconv channels of those rows in its
// accumulators, applies bias + ReLU, and -- because lane (g, j) of a D fragment holds channels 4g..4g+3 of row j, exactly
// the B-fragment layout of a contraction over those channels -- feeds them straight into the head's MFMAs.
// Same operation order as the two separate launches (chunks ascending, 4 steps each): identical bits.
// SKIP (round 4): constant-region skipping carried past the 3x3 stack (csrc/bev.hip).  A site whose last-layer output is the
// stack's constant c_last (dist > reach and further than breach from the image border: the predicate of k_bev_conv3x3<SKIP> for a
// layer behind the last one) has ONE deconv + head result per sub-site, `chead` (4 x 16 floats, evaluated by this kernel on a
// constant input: insmos_deconv_head_constant); a wave whose 16 sites are all such stores it and leaves.  Same bits.
template <int NT, bool SKIP = false>  // NT: channel tiles of the deconv output per sub-site (CUP / 16)
__global__ void __launch_bounds__(64) k_deconv_head(const float* __restrict__ x, uint32_t n_site, int ld_x, int n16_in,
                                                     const float* __restrict__ wd, const float* __restrict__ bd,
                                                     const float* __restrict__ wh, const float* __restrict__ bh,
                                                     float* __restrict__ head, int ld_head, int head_cout,
                                                     const uint8_t* __restrict__ dist, int H, int W, int reach, int breach,
                                                     const float* __restrict__ chead) {
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    const uint32_t sub = blockIdx.x & 3u, rg = blockIdx.x >> 2;
    const uint32_t row = rg * 16u + (uint32_t)j;
    const uint32_t rowc = row < n_site ? row : n_site - 1;
    auto store_head = [&](f32x4 o) {   // o = the lane's four head channels of its site, bias included
        const uint32_t co0 = 4u * g;
        if (row >= n_site || (int)co0 >= head_cout) return;
        float* op = head + ((size_t)row * 4 + sub) * ld_head + co0;
        if ((int)co0 + 3 < head_cout && (ld_head & 3) == 0) {
            *(f32x4*)op = o;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((int)co0 + r < head_cout) op[r] = o[r];
        }
    };
    if constexpr (SKIP) {
        bool on = false;
        if (row < n_site) {
            const int gx = (int)(row % (uint32_t)W), gy = (int)((row / (uint32_t)W) % (uint32_t)H);
            const int bd_ = min(min(gy, H - 1 - gy), min(gx, W - 1 - gx));
            on = (int)dist[row] <= reach || bd_ <= breach;
        }
        if (__ballot(on) == 0ull) {   // (wave-uniform)
            store_head(*(const f32x4*)(chead + sub * 16u + 4u * g));
            return;
        }
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)n_site * ld_x * 4), 0x00020000);
    const uint32_t ntile_d = 4u * NT;  // channel tiles of the packed deconv layer
    const __amdgpu_buffer_rsrc_t rs_wd =
        __builtin_amdgcn_make_buffer_rsrc((void*)wd, 0, (int)((size_t)n16_in * ntile_d * 1024u), 0x00020000);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t xoff = rowc * (uint32_t)ld_x * 4u + (uint32_t)g * 16u;
    const uint32_t woff = (sub * NT * 256u + (uint32_t)lane * 4u) * 4u;
    for (int c = 0; c < n16_in; ++c) {
        const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff, (uint32_t)c * 64u, 0));
        const uint32_t sw = (uint32_t)c * ntile_d * 1024u;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wd, woff + (uint32_t)t * 1024u, sw, 0));
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[t] = MFMA(a[s2], b[s2], acc[t]);
        }
    }
    // deconv epilogue (folded BN shift, ReLU) in registers
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 bv = *(const f32x4*)(bd + (sub * NT + t) * 16u + 4u * g);
        acc[t] += bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
    }
    // heads: contraction over the NT*16 channels just computed; wh packed [chunk = NT][tile = 1][lane][4]
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 a2 = *(const f32x4*)(wh + ((size_t)t * 64 + lane) * 4);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) o = MFMA(a2[s2], acc[t][s2], o);
    }
    if ((int)(4u * g) >= head_cout) return;   // (bh holds head_cout floats)
    o += *(const f32x4*)(bh + 4u * g);
    store_head(o);
}

// active sites of k_deconv_head<.., SKIP> (accounting, bench.py): sites of the 16-site groups that hold a non-constant site
__global__ void k_deconv_head_active_sites(const uint8_t* __restrict__ dist, uint32_t n_site, int H, int W, int reach, int breach,
                                           unsigned long long* __restrict__ out) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;   // (block = 256 = four waves of four 16-site groups)
    bool on = false;
    if (row < n_site) {
        const int gx = (int)(row % (uint32_t)W), gy = (int)((row / (uint32_t)W) % (uint32_t)H);
        const int bd_ = min(min(gy, H - 1 - gy), min(gx, W - 1 - gx));
        on = (int)dist[row] <= reach || bd_ <= breach;
    }
    const unsigned long long bal = __ballot(on);
    const int sh = (threadIdx.x & 63) & ~15;
    const unsigned long long counted = __ballot(((bal >> sh) & 0xFFFFull) != 0 && row < n_site);
    if ((threadIdx.x & 63) == 0 && counted) atomicAdd(out, (unsigned long long)__popcll(counted));
}

// B images stacked along the row axis (site = (b*H + y)*W + x): a tap never leaves its image
__global__ void k_dense_nbr2d(int H, int W, int B, int32_t* __restrict__ nbr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = (int64_t)H * W * B;
    if (t >= n * 9) return;
    int k = (int)(t / n);
    int site = (int)(t % n);
    int img = site / (H * W), loc = site % (H * W);
    int y = loc / W + k / 3 - 1, x = loc % W + k % 3 - 1;
    nbr[t] = (y >= 0 && y < H && x >= 0 && x < W) ? (img * H + y) * W + x : -1;
}

__global__ void k_sparse_to_bev(const float* __restrict__ feat, int ld_feat, int C, const int32_t* __restrict__ coords,
                                int64_t n, int D, int H, int W, float* __restrict__ bev) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int64_t i = t / C;
    int c = (int)(t % C);
    int4 q = *(const int4*)(coords + i * 4);  // [b, d, y, x]
    bev[(((int64_t)q.x * H + q.z) * W + q.w) * ((int64_t)C * D) + (int64_t)c * D + q.y] = feat[i * ld_feat + c];
}

}  // namespace insmos

using namespace insmos;

static void chunking(int cin, int& n16, int& has8, int& has4) {
    n16 = cin / 16;
    int rem = cin % 16;
    has8 = (rem & 8) ? 1 : 0;
    has4 = (rem & 4) ? 1 : 0;
}

// floats per lane of an A fragment: 4 per 16-channel chunk; the single-chunk layers Cin = 8 / 4 pack 2 / 1 (see k_sparse_conv)
static int frag_lane_floats(int cin) { return cin == 8 ? 2 : cin == 4 ? 1 : 4; }

extern "C" size_t insmos_packed_weight_floats(int K, int cin, int cout) {
    int n16, h8, h4;
    chunking(cin, n16, h8, h4);
    int ntile = (cout + 15) / 16;
    return (size_t)K * (size_t)(n16 + h8 + h4) * (size_t)ntile * 64 * (size_t)frag_lane_floats(cin) + rowlane_tail_floats(K, cin, cout);
}
// floats of the MFMA fragment part alone (the row-lane tail of the small-channel layers starts here)
static size_t packed_fragment_floats(int K, int cin, int cout) {
    return (size_t)insmos_packed_weight_floats(K, cin, cout) - rowlane_tail_floats(K, cin, cout);
}

extern "C" int insmos_pack_weights_host(const float* taps, int K, int cin_real, int cout_real, int cin, int cout,
                                        float* packed) {
    if (!taps || !packed || K <= 0 || cin % 4 != 0 || cin < cin_real || cout < cout_real) return INSMOS_EINVAL;
    int n16, h8, h4;
    chunking(cin, n16, h8, h4);
    const int nblk = n16 + h8 + h4, ntile = (cout + 15) / 16;
    const int lf = frag_lane_floats(cin);  // floats stored per lane (4; 2 / 1 for the single-chunk layers Cin = 8 / 4)
    auto W = [&](int k, int ci, int co) -> float {
        return (ci < cin_real && co < cout_real) ? taps[((size_t)k * cin_real + ci) * cout_real + co] : 0.f;
    };
    for (int k = 0; k < K; ++k) {
        int blk = 0;
        auto emit = [&](int c0, int width) {  // width = channels per lane group: 4, 2 or 1
            for (int t = 0; t < ntile; ++t)
                for (int l = 0; l < 64; ++l) {
                    int g = l >> 4, i = l & 15;
                    float* dst = packed + ((((size_t)k * nblk + blk) * ntile + t) * 64 + l) * lf;
                    for (int s = 0; s < lf; ++s) dst[s] = (s < width) ? W(k, c0 + width * g + s, t * 16 + i) : 0.f;
                }
            ++blk;
        };
        for (int c = 0; c < n16; ++c) emit(c * 16, 4);
        int c0 = n16 * 16;
        if (h8) { emit(c0, 2); c0 += 8; }
        if (h4) emit(c0, 1);
    }
    if (rowlane_tail_floats(K, cin, cout)) {   // [tap][p][co], spconv_rowlane.hip
        float* tail = packed + packed_fragment_floats(K, cin, cout);
        for (int k = 0; k < K; ++k)
            for (int p = 0; p < cin; ++p)
                for (int co = 0; co < cout; ++co) tail[((size_t)k * cin + p) * cout + co] = W(k, rowlane_ci(cin, p), co);
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
#define CASE(C, J) if (cot == C && jt == J) return k_sparse_conv<C, J, CK, IDENT, ring_depth<C, J>(), 1, false>;
    if constexpr (CK == 0) {
        CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(4, 1) CASE(4, 2) CASE(4, 4)
        CASE(8, 1) CASE(8, 2) CASE(8, 4)
    } else {
        CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4)
    }
#undef CASE
    return nullptr;
}

// split tiles: 4 waves per 16-row tile, all channel tiles per wave; contraction split by chunk (even) or by tap
ConvKernel pick_split(int cot, int ck, bool by_chunk) {
    if (ck == 8) {
        if (cot == 1) return k_sparse_conv<1, 1, 8, false, 3, 4, false>;
        if (cot == 2) return k_sparse_conv<2, 1, 8, false, 3, 4, false>;
        return nullptr;
    }
    if (ck) return nullptr;
#define CASE(C, RR) if (cot == C) return by_chunk ? k_sparse_conv<C, 1, 0, false, RR, 4, true> : k_sparse_conv<C, 1, 0, false, RR, 4, false>;
    CASE(8, 2) CASE(4, 2) CASE(2, 3) CASE(1, 3)
#undef CASE
    return nullptr;
}

// reduced-precision variants (insmos_conv_precision) of the kernels the 16-channel-chunk layers run on: split tiles and the
// unsplit 16-row tile
template <int PREC>
ConvKernel pick_prec(int cot, int split, bool by_chunk) {
    if (split == 4) {
#define CASE(C, RR)                                                                                         \
    if (cot == C)                                                                                           \
        return by_chunk ? k_sparse_conv<C, 1, 0, false, RR, 4, true, 0, PREC> : k_sparse_conv<C, 1, 0, false, RR, 4, false, 0, PREC>;
        CASE(8, 2) CASE(4, 2) CASE(2, 3) CASE(1, 3)
#undef CASE
        return nullptr;
    }
    if (cot == 1) return k_sparse_conv<1, 1, 0, false, 3, 1, false, 0, PREC>;
    if (cot == 2) return k_sparse_conv<2, 1, 0, false, 3, 1, false, 0, PREC>;
    if (cot == 4) return k_sparse_conv<4, 1, 0, false, 2, 1, false, 0, PREC>;
    return nullptr;
}
int g_prec = 0;  // 0 = exact fp32 (the product default), 1 = bf16 operands, 3 = split-bf16 x 3 (see insmos_conv_precision)
// per-HOST-THREAD override (-1 = none): the training convolutions switch precision for their own launches only, so an
// inference forward issued from another host thread meanwhile keeps exact fp32 (insmos_conv_precision_thread)
thread_local int tl_prec = -1;
inline int cur_prec() { return tl_prec >= 0 ? tl_prec : g_prec; }
std::unordered_map<const float*, const void*> g_split_weights;  // packed fp32 weights -> their (hi | lo) bf16 split
std::mutex g_split_mu;

// tuning hooks (insmos_debug_conv_force): generic non-identity layers at an explicit (COT, JT, ring); probe builds
int g_force_cot = 0, g_force_jt = 0, g_force_ring = 0, g_dbg = 0;
int g_quad = -1;  // quad-index kernel for single-chunk layers: -1 = read INSMOS_CONV_QUAD (default on)
// (atomics: the debug hooks may flip them while another host thread launches; every variant gives the same bits)
std::atomic<int> g_half_wide{-1}, g_half_c64{-1};  // insmos_debug_conv_split_half: -1 = environment / defaults
// probe variants of the kernel configurations the heavy S0 layers use (ring >= 16 selects dbg = ring / 16)
template <int DBG>
ConvKernel pick_probe(int cot, int jt, int ck, int split, bool by_chunk) {
    if (ck == 8 && cot == 1 && jt == 2) return k_sparse_conv<1, 2, 8, false, 3, 1, false, DBG>;
    if (ck == 0 && split == 1 && cot == 1 && jt == 1) return k_sparse_conv<1, 1, 0, false, 3, 1, false, DBG>;
    if (ck == 0 && split == 1 && cot == 2 && jt == 1) return k_sparse_conv<2, 1, 0, false, 3, 1, false, DBG>;
    if (ck == 0 && split == 1 && cot == 4 && jt == 1) return k_sparse_conv<4, 1, 0, false, 2, 1, false, DBG>;
    if (ck == 0 && split == 4 && cot == 8 && by_chunk) return k_sparse_conv<8, 1, 0, false, 2, 4, true, DBG>;
    if (ck == 0 && split == 4 && cot == 4 && by_chunk) return k_sparse_conv<4, 1, 0, false, 2, 4, true, DBG>;    // C = 64 layers
    if (ck == 0 && split == 4 && cot == 2 && !by_chunk) return k_sparse_conv<2, 1, 0, false, 3, 4, false, DBG>;  // C = 32, tap split
    if (ck == 0 && split == 4 && cot == 2 && by_chunk) return k_sparse_conv<2, 1, 0, false, 3, 4, true, DBG>;    // 64 -> 32, chunk split
    if (ck == 0 && split == 4 && cot == 1 && !by_chunk) return k_sparse_conv<1, 1, 0, false, 3, 4, false, DBG>;  // 81 taps 32 -> 16
    return nullptr;
}
ConvKernel pick_forced(int cot, int jt, int ring) {
#define CASE(C, J, RR) if (cot == C && jt == J && ring == RR) return k_sparse_conv<C, J, 0, false, RR, 1, false>;
#define CASES(C, J) CASE(C, J, 2) CASE(C, J, 3) CASE(C, J, 4)
    CASES(1, 1) CASES(1, 2) CASES(1, 4) CASES(2, 1) CASES(2, 2) CASES(2, 4) CASES(4, 1) CASES(4, 2) CASES(4, 4)
    CASE(8, 1, 2) CASE(8, 1, 3) CASE(8, 2, 2) CASE(8, 2, 3) CASE(8, 4, 2)
#undef CASES
#undef CASE
    return nullptr;
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// WHICH layers run on split tiles (four waves per 16-row tile) -- the rule is part of a layer's summation order, so it is a
// function of the layer's shape alone (never of its row count): insmos_conv_tap_classes hands it to the tap-compacted kernel.
// (measured on S0: splitting pays when a tile carries >= ~100 (tap, chunk, channel-tile) MFMA groups -- the
// 81-tap C >= 32 4D layers gain 20-30 % -- and costs 30-50 % on the small-C layers that already have 10k+ tiles)
struct SplitRule { int split_env, split_dense, split_work, tap_mod; };
const SplitRule& split_rule() {
    static const SplitRule r = {env_int("INSMOS_CONV_SPLIT", 1), env_int("INSMOS_CONV_SPLIT_DENSE", 1), env_int("INSMOS_CONV_SPLIT_WORK", 100),
                                env_int("INSMOS_SPLIT_TAP_MOD", 1)};
    return r;
}
bool wants_split(int K, int ck, int n16, int ntile_co, bool masked) {
    const SplitRule& r = split_rule();
    const bool co_ok = ntile_co == 1 || ntile_co == 2 || ntile_co == 4 || ntile_co == 8;
    const bool wide = ntile_co >= 4;
    const long tile_work = (long)K * (ck ? 1 : n16) * ntile_co;
    return r.split_env && co_ok && (ck == 0 || ck == 8) &&
           ((masked && K >= 16 && (wide || tile_work >= r.split_work)) || (!ck && wide && r.split_dense && n16 % 4 == 0 && K >= 3));
}

}  // namespace

namespace insmos {
int conv_precision() { return cur_prec(); }
int conv_precision_thread_get() { return tl_prec; }
const void* split_weights_of(const float* wpacked) {
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split_weights.find(wpacked);
    return it == g_split_weights.end() ? nullptr : it->second;
}
}  // namespace insmos

static int sparse_conv_impl(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr,
                            const uint32_t* mask16, int K, int64_t n_out, int64_t row0, const float* wpacked,
                            const float* bias, float* out, int ld_out, int cout, const float* res, int ld_res, int res_mode,
                            int relu_pre, int relu_post, void* stream) {
    if (n_out <= 0 || row0 >= n_out) return INSMOS_OK;
    if (row0 < 0) return INSMOS_EINVAL;
    row0 &= ~(int64_t)15;  // whole 16-row groups: a few rows below the requested start are computed too
    const int64_t n_rows = n_out - row0;
    if (!in || n_in <= 0 || !wpacked || !bias || !out || cin <= 0 || ld_in % 4 != 0 || ld_in < cin || K <= 0 || K > 128 ||
        (!nbr && (K != 1 || n_in < n_out)) || cout <= 0 || ld_out < cout || (res_mode != 0 && !res) ||
        ((uintptr_t)in & 15) || n_out * (int64_t)K * 4 >= (1ll << 31) || n_in * (int64_t)ld_in * 4 >= (1ll << 31))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ConvP P;
    P.in = in; P.nbr = nbr; P.mask16 = mask16; P.w = wpacked; P.bias = bias; P.out = out; P.res = res;
    P.w_rl = rowlane_tail_floats(K, cin, cout) ? wpacked + packed_fragment_floats(K, cin, cout) : nullptr;
    P.n_out = (uint32_t)n_out;
    P.row0 = (uint32_t)row0;
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
    // the 81-tap single-chunk layers (MotionNet's BasicBlocks with Cin 8 / 16): LDS-staged gathers (spconv_lds.hip)
    if (!g_force_cot && !g_dbg && conv_lds_ok(P, ck, P.ntile_co)) {
        int rc_lds = INSMOS_OK;
        ProfScope ps(KK_SPARSE_CONV, s);
        ps.meta[0] = K; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = n_rows;
        if (conv_lds_try(P, ck, P.ntile_co, (long)n_rows, s, &rc_lds)) return rc_lds;
    }
    // the small-channel layers (Cin, Cout in {8, 16}): one lane per output row on the vector ALUs (spconv_rowlane.hip)
    if (!g_force_cot && !g_dbg && nbr && conv_rowlane_ok(P)) {
        int rc_rl = INSMOS_OK;
        ProfScope ps(KK_SPARSE_CONV, s);
        ps.meta[0] = K; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = n_rows;
        if (conv_rowlane_try(P, (long)n_rows, s, &rc_rl)) return rc_rl;
    }
    // tile shape, from tools/conv_tune.py sweeps on MI355X: a 16-row gather is ~8x the cost of a coalesced
    // weight fragment, so generic layers always use 16-row tiles (JT = 1) and widen in channels instead:
    // 2 channel tiles per wave, 4 when the layer is large enough to still give >= 2 waves per SIMD.
    // Single-chunk small-C layers (Cin 4/8) use `ck_jt` row groups per wave.
    static int ck_jt = 0;
    if (!ck_jt) { ck_jt = env_int("INSMOS_CK_JT", 1); if (ck_jt != 2 && ck_jt != 4) ck_jt = 1; }
    Cfg best = {1, ck ? ck_jt : 1};
    const long groups = (long)((n_rows + 15) / 16);
    if (!ck) {
        if (P.ntile_co % 4 == 0 && groups * (P.ntile_co / 4) >= 2048) best.cot = 4;
        else if (P.ntile_co % 2 == 0) best.cot = 2;
    } else if (P.ntile_co % 2 == 0) {
        best.cot = 2;
    }
    P.n_otiles = (int)((n_rows + 16 * best.jt - 1) / (16 * best.jt));
    const bool ident = (nbr == nullptr);
    ConvKernel kern = nullptr;
    if (ck == 4) kern = ident ? pick_kernel<4, true>(best.cot, best.jt) : pick_kernel<4, false>(best.cot, best.jt);
    else if (ck == 8) kern = ident ? pick_kernel<8, true>(best.cot, best.jt) : pick_kernel<8, false>(best.cot, best.jt);
    else kern = ident ? pick_kernel<0, true>(best.cot, best.jt) : pick_kernel<0, false>(best.cot, best.jt);
    // split tiles: all channel tiles in one wave, the 4 waves of a block share the row group and split the
    // contraction (by chunk when Cin/16 is a multiple of 4 -- even work --, else by active tap).  Each input row
    // chunk is gathered once per tile instead of once per channel group, and the layer gets 4x the waves.
    int split = 1;
    bool by_chunk = false;
    P.tap_mod = split_rule().tap_mod;
    if (!ident && wants_split(K, ck, P.n16, P.ntile_co, mask16 != nullptr)) {
        // (probes of round 5, measured per layer on a launch set of 8 and REMOVED from the product dispatcher -- profiles/r05_knob_ab_layers.txt:
        //  operand ring 4 / 5 on the tap-split tiles +0.7 / +1.9 %, ring 3 on the chunk-split tiles +12...17 % on the Cin = 128 layers,
        //  four blocks per tile for C = 128 +3.7 %)
        ConvKernel sk = pick_split(P.ntile_co, ck, !ck && P.n16 % 4 == 0);
        int cot_split = P.ntile_co;
        // Few row groups (one window alone: the level-4 layers have ~420): the chunk-split tiles of a wide layer leave most CUs with
        // one or two 4-wave blocks, each latency-bound on its operand loads.  Two blocks per tile, half of the channel tiles each,
        // double the waves (the tile's rows are gathered twice; every output channel keeps its summation order: same bits).
        // (round 5: the threshold was 1536 -- single windows only; measured per layer on a launch set of 8 S0 windows, the C = 128
        //  level-4 layers, 2 592 row groups: conv_up_m4.0 501 -> 467 us, conv_up_t4.* 243 / 256 -> 226 / 228, all convolutions
        //  8 569 -> 8 474 us per set; the C = 64 level-3 layers, 5 056 groups, LOSE 18 % at half width: profiles/r05_knob_ab_layers.txt)
        static const int half_wide_env = env_int("INSMOS_CONV_SPLIT_HALF", 4096), half_c64_env = env_int("INSMOS_CONV_SPLIT_HALF_C64", 1536);
        const int hw_ = g_half_wide.load(std::memory_order_relaxed), hc_ = g_half_c64.load(std::memory_order_relaxed);
        const int half_wide = hw_ >= 0 ? hw_ : half_wide_env, half_c64 = hc_ >= 0 ? hc_ : half_c64_env;
        const int half_below = P.ntile_co >= 8 ? half_wide : half_c64;   // (Cout 64 keeps the single-window threshold)
        if (sk && !ck && P.n16 % 4 == 0 && P.ntile_co >= 4 && groups < half_below) {
            ConvKernel hk = pick_split(P.ntile_co / 2, ck, true);
            if (hk) { sk = hk; cot_split = P.ntile_co / 2; }
        }
        if (sk) {
            split = 4;
            by_chunk = (!ck && P.n16 % 4 == 0);
            best = {cot_split, 1};
            P.n_otiles = (int)groups;
            kern = sk;
        }
    }
    // single-chunk unsplit layers: quad index loads (same bits, fewer vector memory instructions per tap)
    if (g_quad < 0) g_quad = env_int("INSMOS_CONV_QUAD", 1);
    // (measured per layer on a launch set of 8 S0 windows: -7...-13 % on the 27- and 81-tap layers and on the 8-tap 16-channel
    //  ones, +3...5 % on the 8-tap Cin = 8 layers, whose whole tap list is two quads: those stay on the generic kernel)
    if (g_quad && !ident && split == 1 && best.jt == 1 && best.cot <= 2 && ((ck == 8 && K >= 16) || (ck == 0 && P.n16 == 1))) {
        if (ck == 8) kern = best.cot == 1 ? k_sparse_conv_q<1, 8> : k_sparse_conv_q<2, 8>;
        else kern = best.cot == 1 ? k_sparse_conv_q<1, 0> : k_sparse_conv_q<2, 0>;
    }
    // reduced precision (opt-in, never the default): 16-channel-chunk layers with a neighbour table
    // (single-chunk layers stay fp32: they are vector-memory bound, measured 0.91x under the experiment)
    if (cur_prec() && !ck && P.n16 >= 2 && !ident && best.jt == 1) {
        int prec = cur_prec();
        if (prec == 3) {
            const void* ws = split_weights_of(wpacked);
            if (!ws) prec = 0;  // no split weights registered for this layer: exact fp32
            else P.w = (const float*)ws;
        }
        ConvKernel pk = prec == 1 ? pick_prec<1>(best.cot, split, by_chunk) : prec == 3 ? pick_prec<3>(best.cot, split, by_chunk) : nullptr;
        if (pk) kern = pk;
        else P.w = wpacked;
    }
    if (g_force_cot && !ck && !ident && P.ntile_co % g_force_cot == 0) {
        ConvKernel fk = pick_forced(g_force_cot, g_force_jt, g_force_ring);
        if (fk) {
            split = 1;
            by_chunk = false;
            kern = fk;
            best = {g_force_cot, g_force_jt};
            P.n_otiles = (int)((n_rows + 16 * best.jt - 1) / (16 * best.jt));
        }
    }
    if (g_dbg && !ident) {
        ConvKernel pk = nullptr;
        switch (g_dbg) {
            case 1: pk = pick_probe<1>(best.cot, best.jt, ck, split, by_chunk); break;
            case 2: pk = pick_probe<2>(best.cot, best.jt, ck, split, by_chunk); break;
            case 3: pk = pick_probe<3>(best.cot, best.jt, ck, split, by_chunk); break;
            case 4: pk = pick_probe<4>(best.cot, best.jt, ck, split, by_chunk); break;
            case 7: pk = pick_probe<7>(best.cot, best.jt, ck, split, by_chunk); break;
        }
        if (pk) kern = pk;
    }
    // Cin = 32 with rows that are whole 128-byte lines: the same tile on the whole-row gather kernel (spconv_row32.hip; same bits)
    if (kern && !ident && !ck && !cur_prec() && !g_force_cot && !g_dbg) {
        ConvKernelFn rk = conv_row32_pick(P, best.cot, best.jt, split, by_chunk);
        if (rk) kern = rk;
    }
    if (!kern) return INSMOS_EINVAL;
    // chunk-split layers with whole input rows: 32-row tiles with LDS-staged rows (spconv_wide.hip; same bits)
    if (split == 4 && by_chunk && !ident && !cur_prec() && !g_force_cot && !g_dbg) {
        long wb = 0;
        ConvKernelFn wk = conv_wide_pick(P, &wb);
        if (wk) {
            ProfScope ps(KK_SPARSE_CONV, s);
            ps.meta[0] = K; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = n_rows;
            INSMOS_LAUNCH(wk, dim3((unsigned)wb), dim3(320), 0, s, P);   // (four consumer waves + the producer)
            HIP_TRY(hipGetLastError());
            return INSMOS_OK;
        }
    }

    // ---- launch shape: one ONE-WAVE block per tile (split: one 4-wave block per tile)
    const long tiles = (long)P.n_otiles * (P.ntile_co / best.cot);
    const int wpb = split == 1 ? 1 : 4;  // waves per block
    const long nblocks = split == 1 ? tiles : (tiles + (4 / split) - 1) / (4 / split);
    dim3 grid((unsigned)nblocks), block(64 * wpb);
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = K; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = n_rows;
    INSMOS_LAUNCH(kern, grid, block, 0, s, P);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_sparse_conv(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr,
                                  const uint32_t* mask16, int K, int64_t n_out, const float* wpacked, const float* bias,
                                  float* out, int ld_out, int cout, const float* res, int ld_res, int res_mode,
                                  int relu_pre, int relu_post, void* stream) {
    return sparse_conv_impl(in, n_in, ld_in, cin, nbr, mask16, K, n_out, 0, wpacked, bias, out, ld_out, cout, res, ld_res,
                            res_mode, relu_pre, relu_post, stream);
}

extern "C" int insmos_sparse_conv_rows(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr,
                                       const uint32_t* mask16, int K, int64_t n_out, int64_t row0, const float* wpacked,
                                       const float* bias, float* out, int ld_out, int cout, const float* res, int ld_res,
                                       int res_mode, int relu_pre, int relu_post, void* stream) {
    return sparse_conv_impl(in, n_in, ld_in, cin, nbr, mask16, K, n_out, row0, wpacked, bias, out, ld_out, cout, res,
                            ld_res, res_mode, relu_pre, relu_post, stream);
}

extern "C" int insmos_deconv_head(const float* x, int64_t n_site, int ld_x, int cin, const float* wd_packed, const float* bd,
                                  int cup, const float* wh_packed, const float* bh, int head_cout, float* head, int ld_head,
                                  void* stream) {
    if (n_site <= 0) return INSMOS_OK;
    if (!x || !wd_packed || !bd || !wh_packed || !bh || !head || cin % 16 != 0 || ld_x < cin || (ld_x & 3) || cup != 256 ||
        head_cout <= 0 || head_cout > 16 || ld_head < head_cout || ((uintptr_t)x & 15) || n_site * (int64_t)ld_x * 4 >= (1ll << 31))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long groups = (long)((n_site + 15) / 16);
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = 1; ps.meta[1] = cin; ps.meta[2] = 4 * cup; ps.meta[3] = n_site;
    INSMOS_LAUNCH(k_deconv_head<16>, dim3((unsigned)(groups * 4)), dim3(64), 0, s, x, (uint32_t)n_site, ld_x, cin / 16, wd_packed, bd,
                  wh_packed, bh, head, ld_head, head_cout, nullptr, 1, 1, 0, 0, nullptr);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// insmos_deconv_head behind a 3x3 stack of `n_stack` layers run on a map with distance map `dist` (insmos_bev_distance_map, cap >=
// n_stack): 16-site groups whose sites all carry the stack's constant store `chead` (insmos_deconv_head_constant) instead of
// computing.  x = B images of H x W sites (n_site = B * H * W).  Output bits == insmos_deconv_head's.
extern "C" int insmos_deconv_head_skip(const float* x, int64_t n_site, int ld_x, int cin, const float* wd_packed, const float* bd,
                                       int cup, const float* wh_packed, const float* bh, int head_cout, float* head, int ld_head,
                                       const uint8_t* dist, int H, int W, int n_stack, const float* chead, void* stream) {
    if (n_site <= 0) return INSMOS_OK;
    if (!x || !wd_packed || !bd || !wh_packed || !bh || !head || cin % 16 != 0 || ld_x < cin || (ld_x & 3) || cup != 256 ||
        head_cout <= 0 || head_cout > 16 || ld_head < head_cout || ((uintptr_t)x & 15) || n_site * (int64_t)ld_x * 4 >= (1ll << 31) ||
        !dist || !chead || H <= 0 || W <= 0 || n_stack < 1 || n_stack > 200 || n_site % ((int64_t)H * W) != 0 || ((uintptr_t)chead & 15))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long groups = (long)((n_site + 15) / 16);
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = 1; ps.meta[1] = cin; ps.meta[2] = 4 * cup; ps.meta[3] = n_site;
    INSMOS_LAUNCH((k_deconv_head<16, true>), dim3((unsigned)(groups * 4)), dim3(64), 0, s, x, (uint32_t)n_site, ld_x, cin / 16, wd_packed,
                  bd, wh_packed, bh, head, ld_head, head_cout, dist, H, W, n_stack, n_stack - 2, chead);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" size_t insmos_deconv_head_constant_ws_floats(int cin) { return (size_t)16 * (size_t)cin + (size_t)16 * 4 * 16 + 64; }

// chead (4 x 16 floats: [sub-site][head channel], zero beyond head_cout) = what insmos_deconv_head produces at a site whose input
// row is c_in (cin floats), evaluated by the kernel itself on 16 such sites (same bits).  Depends on the weights only.
extern "C" int insmos_deconv_head_constant(const float* wd_packed, const float* bd, int cin, int cup, const float* wh_packed,
                                           const float* bh, int head_cout, const float* c_in, float* chead, float* ws, void* stream) {
    if (!wd_packed || !bd || !wh_packed || !bh || !c_in || !chead || !ws || cin <= 0 || cin % 16 != 0 || cup != 256 || head_cout <= 0 ||
        head_cout > 16 || ((uintptr_t)ws & 15))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* img = ws;                                            // (16, cin)
    float* res = ws + (((size_t)16 * cin + 63) & ~(size_t)63);  // (16 sites x 4 sub-sites, 16)
    for (int i = 0; i < 16; ++i)
        HIP_TRY(hipMemcpyAsync(img + (size_t)i * cin, c_in, (size_t)cin * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(res, 0, (size_t)16 * 4 * 16 * sizeof(float), s));
    int rc = insmos_deconv_head(img, 16, cin, cin, wd_packed, bd, cup, wh_packed, bh, head_cout, res, 16, stream);
    if (rc != INSMOS_OK) return rc;
    HIP_TRY(hipMemcpyAsync(chead, res, (size_t)4 * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));   // site 0: rows 0..3
    return INSMOS_OK;
}

// Accounting (bench.py): sites insmos_deconv_head_skip computes (16-site groups with a non-constant site); *sites_dev: 8 bytes.
extern "C" int insmos_deconv_head_skip_active_sites(const uint8_t* dist, int64_t n_site, int H, int W, int n_stack,
                                                    unsigned long long* sites_dev, void* stream) {
    if (!dist || !sites_dev || n_site <= 0 || n_site >= (1ll << 31) || H <= 0 || W <= 0 || n_stack < 1) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(sites_dev, 0, sizeof(unsigned long long), s));
    INSMOS_LAUNCH(k_deconv_head_active_sites, dim3((unsigned)((n_site + 255) / 256)), dim3(256), 0, s, dist, (uint32_t)n_site, H, W, n_stack,
                  n_stack - 2, sites_dev);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_debug_conv_force(int cot, int jt, int ring) {
    g_dbg = ring / 16;
    g_force_cot = cot; g_force_jt = jt; g_force_ring = ring % 16;
    return INSMOS_OK;
}

// fp32 packed weights -> (hi4 | lo4) bf16 per 16-byte lane slot; same indexing as the fp32 buffer
__global__ void k_split_weights_bf16(const f32x4* __restrict__ w, int64_t n4, u32x2* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    s16x4 hi, lo;
    insmos::split_bf16(w[i], hi, lo);
    out[2 * i] = __builtin_bit_cast(u32x2, hi);
    out[2 * i + 1] = __builtin_bit_cast(u32x2, lo);
}

extern "C" int insmos_split_weights_bf16(const float* wpacked, int64_t n_floats, void* out, void* stream) {
    if (!wpacked || !out || n_floats <= 0 || n_floats % 4 != 0) return INSMOS_EINVAL;
    const int64_t n4 = n_floats / 4;
    INSMOS_LAUNCH(k_split_weights_bf16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                  (const insmos::f32x4*)wpacked, n4, (insmos::u32x2*)out);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_register_split_weights(const float* wpacked, const void* wsplit) {
    if (!wpacked) return INSMOS_EINVAL;
    std::lock_guard<std::mutex> lk(g_split_mu);
    if (wsplit) g_split_weights[wpacked] = wsplit;
    else g_split_weights.erase(wpacked);
    return INSMOS_OK;
}

extern "C" int insmos_conv_precision(int mode) {
    if (mode != 0 && mode != 1 && mode != 3) return INSMOS_EINVAL;
    g_prec = mode;
    return INSMOS_OK;
}

extern "C" int insmos_conv_precision_thread(int mode) {
    if (mode != -1 && mode != 0 && mode != 1 && mode != 3) return INSMOS_EINVAL;
    tl_prec = mode;
    return INSMOS_OK;
}

extern "C" int insmos_debug_conv_split_half(int wide, int c64) {
    if (wide < -1 || c64 < -1) return INSMOS_EINVAL;
    g_half_wide = wide;
    g_half_c64 = c64;
    return INSMOS_OK;
}

// Partial chains a layer's output sums are made of on the tile kernels: 1 = one chain over the taps in ascending order (unsplit
// tiles, the quad-index / row-lane / whole-row forms of them), 4 = tap-split tiles (chains over the taps k % 4 == 0..3, summed
// ((c0 + c1) + c2) + c3), 0 = chunk-split tiles or a probe setting (no tap-compacted form).  `masked`: the table has a mask16 array.
extern "C" int insmos_conv_tap_classes(int K, int cin, int cout, int masked) {
    if (K <= 0 || cin <= 0 || cout <= 0) return 0;
    const int ck = (cin == 4 || cin == 8) ? cin : 0;
    if (!ck && cin % 16 != 0) return 0;
    const int n16 = cin / 16, ntile_co = (cout + 15) / 16;
    if (!wants_split(K, ck, n16, ntile_co, masked != 0)) return 1;
    if (!ck && n16 % 4 == 0) return 0;                  // chunk-split
    if (!pick_split(ntile_co, ck, false)) return 1;     // (no split kernel for the shape: the dispatcher stays unsplit)
    return split_rule().tap_mod ? 4 : 0;
}

extern "C" int insmos_debug_conv_quad(int on) {
    g_quad = on ? 1 : 0;
    return INSMOS_OK;
}

extern "C" int insmos_dense_nbr2d_b(int H, int W, int B, int32_t* nbr, void* stream) {
    if (H <= 0 || W <= 0 || B < 1 || !nbr || (int64_t)H * W * B * 9 * 4 >= (1ll << 31)) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_DENSE_NBR, s);
    int64_t n = (int64_t)H * W * B * 9;
    INSMOS_LAUNCH(k_dense_nbr2d, dim3(cdiv(n, 256)), dim3(256), 0, s, H, W, B, nbr);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
extern "C" int insmos_dense_nbr2d(int H, int W, int32_t* nbr, void* stream) { return insmos_dense_nbr2d_b(H, W, 1, nbr, stream); }

// B stacked images: coords[:, 0] = image of the voxel, bev is (B, H, W, C*D)
extern "C" int insmos_sparse_to_bev_b(const float* feat, int ld_feat, int C, const int32_t* coords, int64_t n, int D,
                                      int H, int W, int B, float* bev, void* stream) {
    if (!feat || !coords || !bev || C <= 0 || D <= 0 || B < 1) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps(KK_MEMSET, s);
        HIP_TRY(hipMemsetAsync(bev, 0, (size_t)B * H * W * C * D * sizeof(float), s));
    }
    if (n > 0) {
        ProfScope ps(KK_TO_BEV, s);
        INSMOS_LAUNCH(k_sparse_to_bev, dim3(cdiv(n * C, 256)), dim3(256), 0, s, feat, ld_feat, C, coords, n, D, H, W,
                           bev);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_sparse_to_bev(const float* feat, int ld_feat, int C, const int32_t* coords, int64_t n, int D,
                                    int H, int W, float* bev, void* stream) {
    return insmos_sparse_to_bev_b(feat, ld_feat, C, coords, n, D, H, W, 1, bev, stream);
}
