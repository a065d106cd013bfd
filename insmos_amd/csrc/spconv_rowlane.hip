// insmos_amd/csrc/spconv_rowlane.hip -- the small-channel sparse convolutions (Cin x Cout <= 16 x 16: MotionNet's 81-tap
// BasicBlocks at 8 / 16 channels, minkunet.py:55-69 + resnet.py:110-119, and the k2s2 maps between them) as a ROW-PER-LANE
// kernel on the vector ALUs.
//
// Why not the matrix cores here (round-3 profile, profiles/r03_layers_b8.csv): the 16-row MFMA tiles of spconv.hip run these
// layers at 12-37 TFLOP/s and at a quarter of the HBM rate -- neither roof.  Their limiter is vector-memory ISSUE: per (16-row
// group, tap) a wave spends an index load, a 16-row gather and a weight-fragment load (2.25-3 instructions) to feed two to four
// MFMAs, half of whose output columns are padding at Cout = 8.  The fp32 VALU peak of the part equals its fp32 MFMA peak
// (v_pk_fma_f32: 2 x 64 FMA per 4 cycles), so the contraction moves to the VALU and the memory side is rebuilt around it:
//
//   * one lane = one output row (RPL rows per lane): per (64 rows, tap) ONE coalesced index load (256 B) and Cin/4 b128 gathers
//     (a lane reads its neighbour's whole row) -- 3 vector-memory instructions per 64 rows and tap instead of 9-12;
//   * the tap's weights are wave-uniform: they come through the SCALAR cache (s_load_dwordx16 from the constant address space)
//     into SGPRs and enter v_pk_fma_f32 as the scalar operand; no weight fragments through the vector memory path at all, and
//     no Cout 8 -> 16 padding (a lane holds exactly Cout accumulators);
//   * taps are walked through the union of the tile's active-tap masks; a lane whose own 16-row group lacks the tap replaces the
//     (possibly unwritten: sparse table stores) entry by -1, whose gather the buffer descriptor answers with zeros.
//
// Bits: per output row and channel the sum is the SAME fmaf chain as the MFMA kernels' (taps ascending; inside a tap MFMA step s,
// lane group g = channel width * g + s, cdna_hip_programming.md: "bit-for-bit a k-ordered f32 fmaf chain"); rows of groups
// without a tap add fma(0, w, acc) = acc.  tests/test_gpu_conv.py::test_rowlane_kernel_is_bitwise_the_mfma_one.
//
// Weights: the "row-lane tail" appended to a layer's packed weights by insmos_pack_weights_host / _device when the layer
// qualifies (rowlane_tail_floats): [tap][p = 4 s + g][co] floats, co contiguous, so that a (co, co + 1) pair is one aligned
// SGPR pair.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace insmos {
namespace {

#define CONSTAS __attribute__((address_space(4)))

__device__ __forceinline__ int rl_pop_or_keep(uint64_t& lo, uint64_t& hi, int keep) {
    const bool use_lo = lo != 0;
    const uint64_t w = use_lo ? lo : hi;
    const int k = (w ? __builtin_ctzll(w) : 0) + (use_lo ? 0 : 64);
    const uint64_t cleared = w & (w - 1);
    const bool any = w != 0;
    lo = use_lo ? cleared : lo;
    hi = use_lo ? hi : cleared;
    return any ? k : keep;
}

// acc(lo, hi) += x * (w.lo, w.hi) with x = the low (HI = 0) or high (HI = 1) half of the register pair `xp`, broadcast by op_sel;
// w is an SGPR pair.  (Inline asm: from plain vector code hipcc broadcasts an odd element by copying it into the low half of a
// fresh register pair first -- and hoists those copies to the end of the previous loop iteration, where they wait for the loads.)
template <int HI>
__device__ __forceinline__ void pk_fma_bcast(f32x2& acc, const f32x2 xp, const f32x2 w) {
    if constexpr (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(xp), "s"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(xp), "s"(w));
}

// CIN in {8, 16}; CO in {8, 16} (== the layer's cout); RPL rows per lane
// DBG (probe builds, insmos_debug_conv_rowlane(mode | dbg << 4, .)): bit 0 = no FMAs, bit 1 = no gathers, bit 2 = gathers of the
// lane's OWN row (same instructions and bytes, consecutive lines), bit 3 = no weight loads
template <int CIN, int CO, int RPL, int DBG = 0>
__global__ void __launch_bounds__(64) k_conv_rowlane(ConvP P) {
    constexpr int NX = CIN / 4;   // b128 loads per gathered row
    constexpr int NC2 = CO / 2;   // accumulator pairs per row
    constexpr int WIDTH = CIN / 4;  // channels per MFMA lane group (the chain order below)
    const int lane = threadIdx.x;
    const uint32_t n_out = P.n_out;
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;
    const uint32_t n4 = n_out * 4u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nb = __builtin_amdgcn_make_buffer_rsrc((void*)P.nbr, 0, (int)((uint32_t)P.K * n4), 0x00020000);
    const CONSTAS f32x2* wc = (const CONSTAS f32x2*)(const CONSTAS void*)P.w_rl;

    const uint32_t r0 = P.row0 + blockIdx.x * (64u * RPL);
    uint32_t row[RPL], rowoff[RPL];
    uint32_t own[RPL][4];   // the active-tap mask of my row's 16-row group
    uint64_t tlo = 0, thi = 0;
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        row[q] = r0 + (uint32_t)q * 64u + (uint32_t)lane;
        const bool ok = row[q] < n_out;
        rowoff[q] = (ok ? row[q] : n_out - 1u) * 4u;
        uint4 m;
        if (P.mask16) {
            m = *(const uint4*)(P.mask16 + (size_t)(rowoff[q] >> 6) * 4);
        } else {
            const int K = P.K;
            m.x = K >= 32 ? ~0u : ((1u << K) - 1u);
            m.y = K >= 64 ? ~0u : K > 32 ? ((1u << (K - 32)) - 1u) : 0u;
            m.z = K >= 96 ? ~0u : K > 64 ? ((1u << (K - 64)) - 1u) : 0u;
            m.w = K >= 128 ? ~0u : K > 96 ? ((1u << (K - 96)) - 1u) : 0u;
        }
        own[q][0] = ok ? m.x : 0u; own[q][1] = ok ? m.y : 0u; own[q][2] = ok ? m.z : 0u; own[q][3] = ok ? m.w : 0u;
#pragma unroll
        for (int gl = 0; gl < 64; gl += 16) {
            tlo |= ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)own[q][1], gl) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)own[q][0], gl);
            thi |= ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)own[q][3], gl) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)own[q][2], gl);
        }
    }
    const int nt = __builtin_popcountll(tlo) + __builtin_popcountll(thi);

    f32x2 acc[RPL][NC2];
#pragma unroll
    for (int q = 0; q < RPL; ++q)
#pragma unroll
        for (int c = 0; c < NC2; ++c) acc[q][c] = (f32x2){0.f, 0.f};

    // my row's neighbour under tap k; -1 where my group lacks the tap (the entry may be unwritten there)
    auto load_idx = [&](int k, uint32_t (&idx)[RPL]) {
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            if constexpr (DBG & 16)   // probe: the address pattern of a TILE-major table ([tile][tap][64 rows]); values are garbage
                idx[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, (uint32_t)lane * 4u,
                                                                        ((blockIdx.x * RPL + q) * (uint32_t)P.K + (uint32_t)k) * 256u, 0);
            else
                idx[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_nb, rowoff[q], (uint32_t)k * n4, 0);
        }
    };
    // `live` (wave-uniform): false for the clamped taps past the end of the tile's list -- every lane gathers zeros
    auto gather = [&](int k, bool live, const uint32_t (&idx)[RPL], f32x4 (&x)[RPL][NX]) {
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const uint32_t wsel = k < 64 ? (k < 32 ? own[q][0] : own[q][1]) : (k < 96 ? own[q][2] : own[q][3]);
            const bool has = ((wsel >> (k & 31)) & 1u) && live;
            uint32_t off = (has ? idx[q] : 0xFFFFFFFFu) * ld4;   // -1 wraps past the end of the buffer: the loads return 0
            if constexpr (DBG & 4) off = has ? (rowoff[q] >> 2) % (P.in_bytes / ld4) * ld4 : 0xFFFFFFF0u;
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                if constexpr (DBG & 2) {
                    if (k == 1000) x[q][c] = (f32x4){1.f, 1.f, 1.f, 1.f};
                    asm volatile("" ::"v"(off));
                } else {
                    x[q][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, (uint32_t)c * 16u, 0));
                }
            }
        }
    };
    auto fma_tap = [&](int k, const f32x4 (&x)[RPL][NX]) {
        if constexpr (DBG & 1) {
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int c = 0; c < NX; ++c) asm volatile("" ::"v"(x[q][c]));
            return;
        }
        const CONSTAS f32x2* wk = wc + ((DBG & 8) ? (size_t)0 : (size_t)k * (CIN * NC2));
#pragma unroll
        for (int p = 0; p < CIN; ++p) {
            const int ci = WIDTH * (p & 3) + (p >> 2);   // MFMA step s = p >> 2, lane group g = p & 3
#pragma unroll
            for (int c = 0; c < NC2; ++c) {
                const f32x2 w2 = wk[p * NC2 + c];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const f32x4 xr = x[q][ci >> 2];
                    const f32x2 xp = (ci & 2) ? (f32x2){xr[2], xr[3]} : (f32x2){xr[0], xr[1]};
                    if (ci & 1) pk_fma_bcast<1>(acc[q][c], xp, w2);
                    else pk_fma_bcast<0>(acc[q][c], xp, w2);
                }
            }
        }
    };

    if (nt > 0) {
        // Software pipeline over the tile's taps, three operand slots and three index registers per row: at step t the FMAs of tap
        // t run on x[t % 3], the gathers of tap t + 2 are issued from indices requested two steps earlier, and the indices of tap
        // t + 4 are requested.  The loop is COUNTED and branch-free inside (every FMA unconditional: a guarded FMA block lets hipcc
        // sink the gathers that feed it into the guard, i.e. issue-and-wait -- seen in the first build's ISA); taps past the end of
        // the list are clamped to the last one and gather zeros, so the up to two extra steps add fma(0, w, acc) = acc.
        int kq[5];
        kq[0] = rl_pop_or_keep(tlo, thi, 0);
#pragma unroll
        for (int i = 1; i < 5; ++i) kq[i] = rl_pop_or_keep(tlo, thi, kq[i - 1]);
        uint32_t ia[RPL], ib[RPL], ic[RPL];
        f32x4 x0[RPL][NX], x1[RPL][NX], x2[RPL][NX];
        if constexpr (DBG & 2) {
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int c = 0; c < NX; ++c) x0[q][c] = x1[q][c] = x2[q][c] = (f32x4){0.5f, 0.5f, 0.5f, 0.5f};
        }
        load_idx(kq[0], ia);
        load_idx(kq[1], ib);
        load_idx(kq[2], ic);
        gather(kq[0], true, ia, x0);
        load_idx(kq[3], ia);
        gather(kq[1], 1 < nt, ib, x1);
        // one step: request the indices of tap kq[4] into IF (consumed by the previous step's gather), gather tap kq[2] from IN
        // into XN, FMAs of tap kq[0] on XC
#define RL_STEP(XC, XN, IN, IF, T)                                                  \
    {                                                                               \
        load_idx(kq[4], IF);                                                        \
        gather(kq[2], (T) + 2 < nt, IN, XN);                                        \
        fma_tap(kq[0], XC);                                                         \
        kq[0] = kq[1]; kq[1] = kq[2]; kq[2] = kq[3]; kq[3] = kq[4];                 \
        kq[4] = rl_pop_or_keep(tlo, thi, kq[4]);                                    \
    }
        for (int t = 0; t < nt; t += 3) {
            RL_STEP(x0, x2, ic, ib, t)
            RL_STEP(x1, x0, ia, ic, t + 1)
            RL_STEP(x2, x1, ib, ia, t + 2)
        }
#undef RL_STEP
    }

    // ---- epilogue: the operation order of k_sparse_conv's finish(): + bias, ReLU, residual / channel-pair residual, ReLU
    const CONSTAS f32x2* bc = (const CONSTAS f32x2*)(const CONSTAS void*)P.bias;
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        if (row[q] >= n_out) continue;
        float v[CO];
#pragma unroll
        for (int c = 0; c < NC2; ++c) {
            const f32x2 b2 = bc[c];
            v[2 * c] = acc[q][c][0] + b2[0];
            v[2 * c + 1] = acc[q][c][1] + b2[1];
        }
        if (P.relu_pre) {
#pragma unroll
            for (int c = 0; c < CO; ++c) v[c] = fmaxf(v[c], 0.f);
        }
        if (P.res_mode == 1) {
            const float* rp = P.res + (size_t)row[q] * P.ld_res;
            if (P.vec_store) {
#pragma unroll
                for (int c = 0; c < CO; c += 4) {
                    const f32x4 r4 = *(const f32x4*)(rp + c);
                    v[c] += r4[0]; v[c + 1] += r4[1]; v[c + 2] += r4[2]; v[c + 3] += r4[3];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CO; ++c) v[c] += rp[c];
            }
        } else if (P.res_mode == 2) {
            const float* rp = P.res + (size_t)row[q] * P.ld_res;
#pragma unroll
            for (int c = 0; c < CO; ++c) v[c] += rp[2 * c] + rp[2 * c + 1];
        }
        if (P.relu_post) {
#pragma unroll
            for (int c = 0; c < CO; ++c) v[c] = fmaxf(v[c], 0.f);
        }
        float* op = P.out + (size_t)row[q] * P.ld_out;
        if (P.vec_store) {
#pragma unroll
            for (int c = 0; c < CO; c += 4) *(f32x4*)(op + c) = (f32x4){v[c], v[c + 1], v[c + 2], v[c + 3]};
        } else {
#pragma unroll
            for (int c = 0; c < CO; ++c) op[c] = v[c];
        }
    }
}

typedef void (*RlKernel)(ConvP);
int g_rowlane_dbg = 0;
RlKernel pick_rowlane(int cin, int co, int rpl) {
    if (g_rowlane_dbg && rpl == 1) {
#define PROBE(CI, C)                                                              \
    if (cin == CI && co == C) switch (g_rowlane_dbg) {                            \
        case 1: return k_conv_rowlane<CI, C, 1, 1>;                               \
        case 2: return k_conv_rowlane<CI, C, 1, 2>;                               \
        case 3: return k_conv_rowlane<CI, C, 1, 3>;                               \
        case 4: return k_conv_rowlane<CI, C, 1, 4>;                               \
        case 8: return k_conv_rowlane<CI, C, 1, 8>;                               \
        case 5: return k_conv_rowlane<CI, C, 1, 5>;                               \
        case 21: return k_conv_rowlane<CI, C, 1, 21>;                             \
        case 20: return k_conv_rowlane<CI, C, 1, 20>;                             \
    }
        PROBE(8, 8) PROBE(16, 8) PROBE(8, 16)
#undef PROBE
    }
#define CASE(CI, C, R) if (cin == CI && co == C && rpl == R) return k_conv_rowlane<CI, C, R>;
    CASE(8, 8, 1) CASE(8, 8, 2) CASE(8, 16, 1) CASE(8, 16, 2) CASE(16, 8, 1) CASE(16, 8, 2) CASE(16, 16, 1) CASE(16, 16, 2)
#undef CASE
    return nullptr;
}

// which layers take the row-per-lane kernel: bit 0 = 8 x 8 layers with K >= 16 (block1, block8.conv2), bit 1 = K < 16 with
// Cin x Cout <= 128 (the k2s2 maps), bit 2 = 8 x 16 / 16 x 8 with K >= 16, bit 3 = 16 x 16; -1 = read INSMOS_CONV_ROWLANE (default 3:
// measured per layer on a launch set of 8 windows, profiles/r04_layers_rowlane_variants.csv -- 8 x 8 x 81 taps 1.35-1.45x faster
// than the MFMA tiles, the 8-tap maps 1.1-1.2x, 8 x 16 / 16 x 8 slower (0.8-0.9x), 16 x 16 much slower (0.5-0.6x))
int g_rowlane = -1;
int g_rowlane_rpl = 0;   // rows per lane: 0 = read INSMOS_CONV_ROWLANE_RPL (default 1)

}  // namespace

size_t rowlane_tail_floats(int K, int cin, int cout) {
    if ((cin != 8 && cin != 16) || (cout != 8 && cout != 16) || K < 2) return 0;
    return (size_t)K * (size_t)cin * (size_t)cout;
}

static void rowlane_env() {
    if (g_rowlane < 0) {
        const char* e = getenv("INSMOS_CONV_ROWLANE");
        g_rowlane = e ? atoi(e) : 3;
    }
    if (!g_rowlane_rpl) {
        const char* e = getenv("INSMOS_CONV_ROWLANE_RPL");
        g_rowlane_rpl = (e && atoi(e) == 2) ? 2 : 1;
    }
}

bool conv_rowlane_ok(const ConvP& P) {
    rowlane_env();
    if (!g_rowlane || !P.nbr || !P.w_rl || !rowlane_tail_floats(P.K, P.cin, P.cout)) return false;
    const int cc = P.cin * P.cout;
    const int need = cc > 128 ? 8 : P.K < 16 ? 2 : cc > 64 ? 4 : 1;
    return (g_rowlane & need) != 0 && pick_rowlane(P.cin, P.cout, g_rowlane_rpl) != nullptr;
}

bool conv_rowlane_try(const ConvP& P, long n_rows, hipStream_t s, int* rc) {
    if (!conv_rowlane_ok(P)) return false;
    RlKernel kern = pick_rowlane(P.cin, P.cout, g_rowlane_rpl);
    const long tiles = (n_rows + 64 * g_rowlane_rpl - 1) / (64 * g_rowlane_rpl);
    // probe only (INSMOS_ROWLANE_PROBE=1): INSMOS_ROWLANE_LDS bytes of dynamic LDS per one-wave workgroup cap the waves resident per CU
    static const bool probe_env = getenv("INSMOS_ROWLANE_PROBE") != nullptr;
    size_t lds = 0;
    if (probe_env) { const char* e = getenv("INSMOS_ROWLANE_LDS"); lds = e ? (size_t)atoi(e) : 0; }
    INSMOS_LAUNCH(kern, dim3((unsigned)tiles), dim3(64), lds, s, P);
    *rc = hipGetLastError() == hipSuccess ? INSMOS_OK : INSMOS_EHIP;
    return true;
}

}  // namespace insmos

extern "C" int insmos_debug_conv_rowlane(int mode, int rows_per_lane) {
    if (mode < -1 || mode > 1023 || (rows_per_lane != 0 && rows_per_lane != 1 && rows_per_lane != 2)) return INSMOS_EINVAL;
    insmos::g_rowlane_dbg = mode >= 0 ? mode >> 4 : 0;   // (probe builds: results are wrong by construction)
    if (mode >= 0) mode &= 15;
    insmos::g_rowlane = mode;
    insmos::g_rowlane_rpl = rows_per_lane;
    return INSMOS_OK;
}
