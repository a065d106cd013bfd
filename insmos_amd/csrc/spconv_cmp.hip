// insmos_amd/csrc/spconv_cmp.hip -- output-stationary sparse convolution with ROW COMPACTION for the wide layers
// (Cout = 64 / 128, Cin a multiple of 16: UNetV2 levels 3-4 and their decoder blocks, spconv_unet.py:120-207).
//
// Why: on these layers the tile kernel (spconv.hip) is bound by two things the tile shape cannot fix.  (1) Only ~70 % of the
// rows of an ACTIVE (16-row group, tap) slot have the neighbour (profiles/r01_layer_work_s0.csv, group_fill): the MFMAs
// of the absent rows are issued for nothing.  (2) Every 16-row tile re-reads the tap's whole Cin x Cout weight block (64 KiB
// at 128 x 128) from L2: at ~20 active taps per tile that is the L2 bandwidth of the chip.
// Here a workgroup owns RT = 64 (32, 16 for small launches) consecutive output rows.  Per active tap it
//   * ballots the rows that HAVE the neighbour and compacts them (neighbour row, output row) into LDS lists -- one pass
//     over the tile's slice of the table, before any arithmetic;
//   * keeps the tap's weight fragments IN REGISTERS while it walks the compacted list 16 rows at a time: full MFMA columns
//     (only the last chunk of a tap is partial) and one weight fetch per tap and tile instead of one per 16-row group;
//   * adds each chunk's product into per-row accumulators in LDS (the compacted column j belongs to output row list[j]).
// The four waves split the OUTPUT CHANNELS (a wave owns Cout/4 of them for all rows), so no two waves ever touch the same
// accumulator: no atomics, no reduction, and each row's value is a fixed expression of its own neighbourhood --
//     out[o] = epilogue( sum over taps k ascending of ( sum over 16-channel chunks ascending, 4 steps each, of x[nbr[k][o]] W[k] ) )
// -- whatever rows share its tile and whatever RT is: deterministic and layout-independent (a batch and a single window
// give the same bits).  Exact fp32 on v_mfma_f32_16x16x4_f32; fragment roles as in spconv.hip (A = weights, B = rows).
#include <cstdlib>
#include "common.h"

namespace insmos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CMP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct CmpP {
    const float* in;
    const int32_t* nbr;
    const uint32_t* mask16;
    const float* w;
    const float* bias;
    float* out;
    const float* res;
    uint32_t n_out, row0, in_bytes;
    int ld_in, n16, K, ld_out, cout, ld_res, res_mode, relu_pre, relu_post, ntile_co;
};

// NTW: output-channel tiles per wave (Cout = 64 * NTW); RT: output rows per workgroup; CB: 16-channel chunks whose weights a
// wave holds at a time (Cin is walked in blocks of CB chunks; CB * NTW * 4 VGPRs)
template <int NTW, int RT, int CB>
__global__ void __launch_bounds__(256) k_sparse_conv_cmp(CmpP P) {
    constexpr int PITCH = 64 * NTW + 4;               // floats per accumulator row (+16 B: rows r, r+1 land on adjacent bank slots)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* acc = (float*)smem;                                             // [RT][PITCH]
    int32_t* l_idx = (int32_t*)(smem + (size_t)RT * PITCH * 4);            // [K][RT] neighbour rows, compacted
    uint8_t* l_row = (uint8_t*)(l_idx + (size_t)P.K * RT);                 // [K][RT] output rows (inside the tile)
    int32_t* l_cnt = (int32_t*)(l_row + (((size_t)P.K * RT + 15) & ~(size_t)15));  // [K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t n_out = P.n_out;
    const uint32_t trow0 = P.row0 + blockIdx.x * RT;   // first output row of the tile (a multiple of 16)
    const int K = P.K;

    // ---- active taps of the tile = union over its 16-row groups
    uint64_t tlo = 0, thi = 0;
    {
        const uint32_t ngrp = (n_out + 15) >> 4;
#pragma unroll
        for (int q = 0; q < RT / 16; ++q) {
            const uint32_t grp = (trow0 >> 4) + q;
            if (grp < ngrp) {
                const uint32_t* mp = P.mask16 + (size_t)grp * 4;
                tlo |= ((uint64_t)__builtin_amdgcn_readfirstlane(mp[1]) << 32) | __builtin_amdgcn_readfirstlane(mp[0]);
                thi |= ((uint64_t)__builtin_amdgcn_readfirstlane(mp[3]) << 32) | __builtin_amdgcn_readfirstlane(mp[2]);
            }
        }
    }
    const int nt = __builtin_popcountll(tlo) + __builtin_popcountll(thi);

    // ---- phase 1: compacted (neighbour row, output row) lists, one wave per tap (slot t -> wave t % 4); my accumulator
    // columns start at zero
    {
        uint64_t lo = tlo, hi = thi;
        for (int t = 0; t < nt; ++t) {
            const bool use_lo = lo != 0;
            const uint64_t wd = use_lo ? lo : hi;
            const int k = __builtin_ctzll(wd) + (use_lo ? 0 : 64);
            if (use_lo) lo = wd & (wd - 1); else hi = wd & (wd - 1);
            if ((t & 3) != wave) continue;
            const uint32_t o = trow0 + (uint32_t)lane;
            int32_t idx = -1;
            if (lane < RT && o < n_out) idx = P.nbr[(size_t)k * n_out + o];
            const unsigned long long bal = __ballot(idx >= 0);
            if (idx >= 0) {
                const int pos = __popcll(bal & ((1ull << lane) - 1ull));
                l_idx[t * RT + pos] = idx;
                l_row[t * RT + pos] = (uint8_t)lane;
            }
            if (lane == 0) l_cnt[t] = __popcll(bal);
        }
        for (int i = lane; i < RT * 16 * NTW / 4; i += 64) {   // float4 pieces of this wave's columns
            const int r = i / (4 * NTW), c4 = i % (4 * NTW);
            *(f32x4*)(acc + r * PITCH + wave * 16 * NTW + c4 * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    const uint32_t blk_bytes = (uint32_t)P.ntile_co * 1024u;                 // one (tap, chunk) block of A fragments
    const uint32_t tap_bytes = (uint32_t)P.n16 * blk_bytes;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)K * tap_bytes), 0x00020000);
    const uint32_t aoff = (uint32_t)(wave * NTW) * 1024u + (uint32_t)lane * 16u;   // my first fragment inside a block
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;
    const uint32_t goff = (uint32_t)g * 16u;
    const int n16 = P.n16;

    // ---- phase 2: taps ascending; per tap, Cin in blocks of CB chunks (weights in registers), the compacted rows 16 at a time
    {
        uint64_t lo = tlo, hi = thi;
        for (int t = 0; t < nt; ++t) {
            const bool use_lo = lo != 0;
            const uint64_t wd = use_lo ? lo : hi;
            const int k = __builtin_ctzll(wd) + (use_lo ? 0 : 64);
            if (use_lo) lo = wd & (wd - 1); else hi = wd & (wd - 1);
            const int cnt = l_cnt[t];
            const int nq = (cnt + 15) >> 4;
            for (int c0 = 0; c0 < n16; c0 += CB) {
                f32x4 a[CB][NTW];
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int it = 0; it < NTW; ++it)
                        a[c][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_w, aoff + (uint32_t)it * 1024u, (uint32_t)k * tap_bytes + (uint32_t)(c0 + c) * blk_bytes, 0));
                // first chunk's rows
                int p = j;
                int32_t idx = p < cnt ? l_idx[t * RT + p] : -1;
                uint32_t roff = (uint32_t)idx * ld4 + goff;          // (-1 wraps out of range: the loads return 0)
                f32x4 b[CB];
#pragma unroll
                for (int c = 0; c < CB; ++c)
                    b[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, roff, (uint32_t)(c0 + c) * 64u, 0));
                for (int q = 0; q < nq; ++q) {
                    // next chunk's rows are requested before this chunk's MFMAs
                    const int pn = (q + 1) * 16 + j;
                    const int32_t idxn = (q + 1 < nq && pn < cnt) ? l_idx[t * RT + pn] : -1;
                    const uint32_t roffn = (uint32_t)idxn * ld4 + goff;
                    f32x4 bn[CB];
#pragma unroll
                    for (int c = 0; c < CB; ++c)
                        bn[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, roffn, (uint32_t)(c0 + c) * 64u, 0));
                    f32x4 d[NTW];
#pragma unroll
                    for (int it = 0; it < NTW; ++it) d[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < CB; ++c)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int it = 0; it < NTW; ++it) d[it] = CMP_MFMA(a[c][it][s], b[c][s], d[it]);
                    // column j of the product belongs to output row l_row[p]: add into that row's accumulator (my channels)
                    const int pc = q * 16 + j;
                    if (pc < cnt) {
                        float* ap = acc + (int)l_row[t * RT + pc] * PITCH + wave * 16 * NTW + 4 * g;
#pragma unroll
                        for (int it = 0; it < NTW; ++it) {
                            f32x4 v = *(f32x4*)(ap + it * 16);
                            v += d[it];
                            *(f32x4*)(ap + it * 16) = v;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CB; ++c) b[c] = bn[c];
                }
            }
        }
    }
    // (a wave reads back only the columns it wrote itself: no barrier)

    // ---- epilogue: lane (g, j) finishes channels co0..co0+3 of rows j, 16 + j, ...
    const uint32_t cout = (uint32_t)P.cout;
#pragma unroll
    for (int q = 0; q < RT / 16; ++q) {
        const uint32_t o = trow0 + q * 16 + j;
        if (o >= n_out) continue;
#pragma unroll
        for (int it = 0; it < NTW; ++it) {
            const uint32_t co0 = (uint32_t)(wave * NTW + it) * 16u + 4u * g;
            if (co0 >= cout) continue;
            f32x4 v = *(const f32x4*)(acc + (q * 16 + j) * PITCH + wave * 16 * NTW + it * 16 + 4 * g);
            v += *(const f32x4*)(P.bias + co0);
            if (P.relu_pre) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (P.res_mode == 1) {
                v += *(const f32x4*)(P.res + (size_t)o * P.ld_res + co0);
            } else if (P.res_mode == 2) {
                const float* rp = P.res + (size_t)o * P.ld_res + 2 * co0;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rp[2 * r] + rp[2 * r + 1];
            }
            if (P.relu_post) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            *(f32x4*)(P.out + (size_t)o * P.ld_out + co0) = v;
        }
    }
}

typedef void (*CmpKernel)(CmpP);

template <int NTW>
static CmpKernel pick_cmp(int rt, int cb) {
    if (rt == 64) return cb == 8 ? k_sparse_conv_cmp<NTW, 64, 8> : cb == 4 ? k_sparse_conv_cmp<NTW, 64, 4> : nullptr;
    if (rt == 32) return cb == 8 ? k_sparse_conv_cmp<NTW, 32, 8> : cb == 4 ? k_sparse_conv_cmp<NTW, 32, 4> : nullptr;
    if (rt == 16) return cb == 8 ? k_sparse_conv_cmp<NTW, 16, 8> : cb == 4 ? k_sparse_conv_cmp<NTW, 16, 4> : nullptr;
    return nullptr;
}

// 1 = handled here, 0 = not a layer this kernel is built for (the caller falls back to the tile kernel), < 0 = error.
// Eligibility depends on the LAYER only (K, Cin, Cout, epilogue alignment), never on the launch size, so that a layer's
// summation order -- its bits -- is the same for a single window and a batch; the launch size only picks RT.
int sparse_conv_compact(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr, const uint32_t* mask16, int K,
                        int64_t n_out, int64_t row0, const float* wpacked, const float* bias, float* out, int ld_out, int cout,
                        const float* res, int ld_res, int res_mode, int relu_pre, int relu_post, hipStream_t s) {
    static const int enabled = [] { const char* e = getenv("INSMOS_CONV_COMPACT"); return (e && e[0] == '0') ? 0 : 1; }();
    if (!enabled || !nbr || !mask16 || K < 16 || K > 128 || cin % 16 != 0 || (cout != 64 && cout != 128)) return 0;
    const int n16 = cin / 16;
    const int cb = n16 % 8 == 0 ? 8 : n16 % 4 == 0 ? 4 : 0;     // Cin = 64, 128, 192, 256, ... (80 and 144 stay on the tile kernel)
    if (!cb) return 0;
    if ((ld_out & 3) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) ||
        (res_mode == 1 && ((ld_res & 3) || ((uintptr_t)res & 15))) || (res_mode != 0 && !res))
        return 0;
    const int64_t n_rows = n_out - row0;
    const int rt = n_rows >= 32768 ? 64 : n_rows >= 8192 ? 32 : 16;
    CmpKernel kern = cout == 128 ? pick_cmp<2>(rt, cb) : pick_cmp<1>(rt, cb);
    if (!kern) return 0;
    CmpP P;
    P.in = in; P.nbr = nbr; P.mask16 = mask16; P.w = wpacked; P.bias = bias; P.out = out; P.res = res;
    P.n_out = (uint32_t)n_out; P.row0 = (uint32_t)row0;
    P.in_bytes = (uint32_t)((n_in - 1) * (int64_t)ld_in * 4 + (int64_t)cin * 4);
    P.ld_in = ld_in; P.n16 = n16; P.K = K; P.ld_out = ld_out; P.cout = cout; P.ld_res = ld_res; P.res_mode = res_mode;
    P.relu_pre = relu_pre; P.relu_post = relu_post; P.ntile_co = cout / 16;
    const int pitch = cout + 4;
    const size_t lds = (size_t)rt * pitch * 4 + (size_t)K * rt * 4 + (((size_t)K * rt + 15) & ~(size_t)15) + (size_t)K * 4;
    if (lds > (64u << 10)) {
        static bool raised[2][3] = {{false, false, false}, {false, false, false}};  // (idempotent; a race only repeats the call)
        bool& r = raised[cout == 128][rt == 64 ? 0 : rt == 32 ? 1 : 2];
        if (!r) {
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96u << 10)));
            r = true;
        }
        if (lds > (96u << 10)) return 0;
    }
    const unsigned grid = (unsigned)((n_rows + rt - 1) / rt);
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = K; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = n_rows;
    INSMOS_LAUNCH(kern, dim3(grid), dim3(256), lds, s, P);
    HIP_TRY(hipGetLastError());
    return 1;
}

}  // namespace insmos
