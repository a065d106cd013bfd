// insmos_amd/csrc/train.hip -- first pieces of the training step (BASELINE.json configs[4]):
// the gradients of the output-stationary sparse convolution and the MOS loss with its gradient.
//
//   forward   y[o]  = sum_k x[nbr[k][o]] @ W[k] + b                         (spconv.hip)
//   d/dx      dx[i] = sum_k dy[nbrT[k][i]] @ W[k]^T   -- the SAME kernel on the transposed table with transposed taps
//                     (for a submanifold layer nbrT[k] = nbr[K-1-k]; for strided layers the engine already builds the
//                     transposed table: down <-> inverse, dn <-> up), packed on the device by insmos_pack_weights_device
//   d/dW      dW[k] = sum_o x[nbr[k][o]]^T (x) dy[o]  -- k_conv_dw_rows below (default): per (row chunk, tap, channel block)
//                     partial sums on the matrix cores over the rows that have the tap, then a fixed-order reduction over the
//                     chunks (deterministic, no atomics: the reference's libraries scatter-add with atomics); k_conv_dw (LDS
//                     slabs) and k_conv_dw_mfma (first MFMA design) remain selectable as cross-checks
//   d/db      db    = sum_o dy[o]                     -- k_col_sum (two stages, fixed order)
//   loss      MOSLoss.compute_loss (models/loss.py:20-34): ignored classes -> -inf, softmax, log(clamp(., 1e-8)),
//             class-weighted NLL; k_mos_loss writes the per-point terms and d loss / d logits, reduced in fixed order.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace insmos {

// ---- device-side weight packing (the host twin is insmos_pack_weights_host; training repacks every step) ----
__global__ void k_pack_weights(const float* __restrict__ taps, int K, int cin_real, int cout_real, int cin, int cout, int nblk,
                               int ntile, int n16, int has8, int transpose, int mirror, float* __restrict__ packed) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)K * nblk * ntile * 256;
    if (t >= total) return;
    const int s = (int)(t & 3), l = (int)((t >> 2) & 63);
    int64_t r = t >> 8;
    const int tile = (int)(r % ntile);
    r /= ntile;
    const int blk = (int)(r % nblk);
    const int k = (int)(r / nblk);
    const int g = l >> 4, i = l & 15;
    int c0, width;
    if (blk < n16) { c0 = blk * 16; width = 4; }
    else if (has8 && blk == n16) { c0 = n16 * 16; width = 2; }
    else { c0 = n16 * 16 + (has8 ? 8 : 0); width = 1; }
    // single-chunk layers (Cin = 8 / 4) store `width` floats per lane, everything else 4 (insmos_pack_weights_host)
    const int lf = (nblk == 1 && n16 == 0) ? width : 4;
    if (s >= lf) return;
    float v = 0.f;
    if (s < width) {
        const int ci = c0 + width * g + s, co = tile * 16 + i;  // channel in / out of the PACKED layer
        const int ks = mirror ? K - 1 - k : k;
        if (!transpose) {
            if (ci < cin_real && co < cout_real) v = taps[((int64_t)ks * cin_real + ci) * cout_real + co];
        } else if (ci < cout_real && co < cin_real) {
            // the packed layer maps cout_real-wide rows back to cin_real-wide ones: its W'[ci][co] = taps[ks][co][ci]
            v = taps[((int64_t)ks * cin_real + co) * cout_real + ci];
        }
    }
    packed[(t >> 2) * lf + s] = v;
}

// the row-lane tail of the small-channel layers ([tap][p][co], spconv_rowlane.hip), same transpose / mirror rules as above
__global__ void k_pack_rowlane_tail(const float* __restrict__ taps, int K, int cin_real, int cout_real, int cin, int cout,
                                    int transpose, int mirror, float* __restrict__ tail) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * cin * cout) return;
    const int co = t % cout, p = (t / cout) % cin, k = t / (cout * cin);
    const int ci = rowlane_ci(cin, p);
    const int ks = mirror ? K - 1 - k : k;
    float v = 0.f;
    if (!transpose) {
        if (ci < cin_real && co < cout_real) v = taps[((int64_t)ks * cin_real + ci) * cout_real + co];
    } else if (ci < cout_real && co < cin_real) {
        v = taps[((int64_t)ks * cin_real + co) * cout_real + ci];
    }
    tail[t] = v;
}

// ---- dW: block = 256 threads = 16 x 16 (ci, co) pairs of a 16 x 16 channel tile, one tap, one chunk of rows ----
constexpr int DW_SLAB = 64;
__global__ void __launch_bounds__(256) k_conv_dw(const float* __restrict__ x, int ld_x, const float* __restrict__ dy, int ld_dy,
                                                  const int32_t* __restrict__ nbr, int64_t n_out, int cin, int cout,
                                                  int rows_per_chunk, int n_co_tiles, float* __restrict__ partial, int K) {
    __shared__ float xs[DW_SLAB][17], ds[DW_SLAB][17];
    const int chunk = blockIdx.x, k = blockIdx.y;
    const int ci_t = blockIdx.z / n_co_tiles, co_t = blockIdx.z % n_co_tiles;
    const int tid = threadIdx.x, ci = tid >> 4, co = tid & 15;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = min(r_begin + rows_per_chunk, n_out);
    float acc = 0.f;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += DW_SLAB) {
        // stage: thread (row = tid >> 2, quarter = tid & 3) loads 4 channels of x (gathered) and of dy
        const int rr = tid >> 2, q = tid & 3;
        const int64_t o = r0 + rr;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < r_end) {
            const int idx = nbr ? nbr[(int64_t)k * n_out + o] : (int)o;
            const int c = ci_t * 16 + q * 4, d = co_t * 16 + q * 4;
            if (idx >= 0) {
                const float* xp = x + (int64_t)idx * ld_x + c;
                xv.x = c + 0 < cin ? xp[0] : 0.f; xv.y = c + 1 < cin ? xp[1] : 0.f;
                xv.z = c + 2 < cin ? xp[2] : 0.f; xv.w = c + 3 < cin ? xp[3] : 0.f;
                const float* dp = dy + o * ld_dy + d;
                dv.x = d + 0 < cout ? dp[0] : 0.f; dv.y = d + 1 < cout ? dp[1] : 0.f;
                dv.z = d + 2 < cout ? dp[2] : 0.f; dv.w = d + 3 < cout ? dp[3] : 0.f;
            }
        }
        xs[rr][q * 4 + 0] = xv.x; xs[rr][q * 4 + 1] = xv.y; xs[rr][q * 4 + 2] = xv.z; xs[rr][q * 4 + 3] = xv.w;
        ds[rr][q * 4 + 0] = dv.x; ds[rr][q * 4 + 1] = dv.y; ds[rr][q * 4 + 2] = dv.z; ds[rr][q * 4 + 3] = dv.w;
        __syncthreads();
#pragma unroll 16
        for (int j = 0; j < DW_SLAB; ++j) acc = fmaf(xs[j][ci], ds[j][co], acc);  // rows in ascending order: deterministic
        __syncthreads();
    }
    const int gci = ci_t * 16 + ci, gco = co_t * 16 + co;
    if (gci < cin && gco < cout) partial[(((int64_t)chunk * K + k) * cin + gci) * cout + gco] = acc;
}

// ---- dW on the matrix cores, first design (INSMOS_DW_KERNEL=1 / INSMOS_DW_MFMA=1; tools/dw_bench.py: slower than the LDS
// kernel on the wide layers, 5-15x slower than k_conv_dw_rows) ----
// D[i = ci][j = co] += sum_r A[i][r] * B[r][j] with the contraction over ROWS, 4 rows per v_mfma_f32_16x16x4_f32:
//   A[i = lane & 15][r = lane >> 4] = x[nbr[k][o_r]][ci0 + i]   (0 where the row has no neighbour under tap k)
//   B[r = lane >> 4][j = lane & 15] = dy[o_r][co0 + j]
// One wave = one (row chunk, tap, 16-channel ci tile) and NT co tiles (NT * 4 accumulator registers); 64 neighbour
// indices are fetched per 64 rows and broadcast per 4-row step; a step whose four rows all lack the tap is skipped
// (wave-uniform).  Rows are visited in ascending order and the per-chunk partial sums are reduced in fixed order by
// k_conv_dw_reduce, so the result is deterministic (it differs from k_conv_dw's only in summation order).
#define DW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));

template <int NT>
__global__ void __launch_bounds__(64) k_conv_dw_mfma(const float* __restrict__ x, int ld_x, const float* __restrict__ dy,
                                                     int ld_dy, const int32_t* __restrict__ nbr, int64_t n_out, int cin,
                                                     int cout, int rows_per_chunk, int n_co_groups,
                                                     float* __restrict__ partial, int K) {
    const int chunk = blockIdx.x, k = blockIdx.y;
    const int ci_t = blockIdx.z / n_co_groups, co_g = blockIdx.z % n_co_groups;
    const int lane = threadIdx.x, li = lane & 15, lr = lane >> 4;
    const int ci = ci_t * 16 + li;
    const bool ci_ok = ci < cin;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = min(r_begin + rows_per_chunk, n_out);
    dw_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    int co[NT];
    bool co_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        co[t] = (co_g * NT + t) * 16 + li;
        co_ok[t] = co[t] < cout;
    }
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 64) {
        const int64_t o_mine = r0 + lane;
        const int idx64 = (o_mine < r_end) ? (nbr ? nbr[(int64_t)k * n_out + o_mine] : (int)o_mine) : -1;
        if (__ballot(idx64 >= 0) == 0ull) continue;
#pragma unroll 4
        for (int sidx = 0; sidx < 16; ++sidx) {
            const int idx = __shfl(idx64, 4 * sidx + lr, 64);  // neighbour of row r0 + 4*sidx + lr
            if (__ballot(idx >= 0) == 0ull) continue;            // wave-uniform: none of the four rows has tap k
            const int64_t o = r0 + 4 * sidx + lr;
            const float a = (idx >= 0 && ci_ok) ? x[(int64_t)idx * ld_x + ci] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = (idx >= 0 && co_ok[t]) ? dy[o * ld_dy + co[t]] : 0.f;
                acc[t] = DW_MFMA(a, b, acc[t]);
            }
        }
    }
    // D[i = 4 * (lane >> 4) + reg][j = lane & 15]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!co_ok[t]) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gci = ci_t * 16 + 4 * lr + reg;
            if (gci < cin) partial[(((int64_t)chunk * K + k) * cin + gci) * cout + co[t]] = acc[t][reg];
        }
    }
}

// ---- dW on the matrix cores, second design (the default; insmos_debug_dw_kernel selects the older two) -------------------
// One wave = one (row chunk, tap, ci block, co block) with up to 4 x 4 accumulator tiles.  What it does differently:
//  * PRESENT rows only: per 64 output rows the rows that have a neighbour under tap k are compacted (ballot + prefix popcount
//    -> a 64-entry list in LDS) and walked four at a time, so the matrix cores see no absent rows (a 4D layer has a neighbour in
//    ~35 % of its (row, tap) slots) and the inner loop has no skip branches;
//  * WIDE loads: with VA (VB) = 1, 2 or 4 channel tiles per block, lane (r, i) loads the VA consecutive channels
//    VA*i .. VA*i + VA-1 of its row in ONE dword / b64 / b128 load and element t feeds tile t -- tile t of a block is the
//    channel set {VA*i + t}, a permutation that only matters when the 16 x 16 results are written out.  A 64 x 64 block costs two
//    b128 loads per 4-row step for 16 MFMAs (the first design: 1 + NT dword loads for NT MFMAs);
//  * buffer loads with the row offset in the voffset: the padding rows of the last step of a list read past the end of the
//    buffer and contribute zeros.
// Rows are visited in ascending order and the chunk partials are reduced in fixed order: deterministic (summation order differs
// from the other two kernels').
template <int VA, int VB>
__global__ void __launch_bounds__(64) k_conv_dw_rows(const float* __restrict__ x, uint32_t x_bytes, int ld_x,
                                                     const float* __restrict__ dy, uint32_t dy_bytes, int ld_dy,
                                                     const int32_t* __restrict__ nbr, int64_t n_out, int cin, int cout,
                                                     int rows_per_chunk, int n_co_blk, float* __restrict__ partial, int K) {
    __shared__ int2 list[2][64];
    const int chunk = blockIdx.x, k = blockIdx.y;
    const int ci_b = blockIdx.z / n_co_blk, co_b = blockIdx.z % n_co_blk;
    const int lane = threadIdx.x, li = lane & 15, lr = lane >> 4;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = min(r_begin + rows_per_chunk, n_out);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)dy_bytes, 0x00020000);
    // this lane's first channel inside the block, as a byte offset; lanes whose channels lie beyond the layer read zeros
    const int ca = ci_b * 16 * VA + VA * li, cb = co_b * 16 * VB + VB * li;
    const bool a_ok = ca + VA - 1 < cin, b_ok = cb + VB - 1 < cout;
    const uint32_t offa = (uint32_t)ca * 4u, offb = (uint32_t)cb * 4u;
    const uint32_t ldx4 = (uint32_t)ld_x * 4u, ldd4 = (uint32_t)ld_dy * 4u;

    dw_f32x4 acc[VA][VB];
#pragma unroll
    for (int ta = 0; ta < VA; ++ta)
#pragma unroll
        for (int tb = 0; tb < VB; ++tb) acc[ta][tb] = dw_f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_a = [&](uint32_t off, float (&a)[VA]) {
        if constexpr (VA == 4) {
            dw_f32x4 v = __builtin_bit_cast(dw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
            a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
        } else if constexpr (VA == 2) {
            dw_f32x2 v = __builtin_bit_cast(dw_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, off, 0, 0));
            a[0] = v[0]; a[1] = v[1];
        } else {
            a[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
        }
    };
    auto load_b = [&](uint32_t off, float (&b)[VB]) {
        if constexpr (VB == 4) {
            dw_f32x4 v = __builtin_bit_cast(dw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 0, 0));
            b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
        } else if constexpr (VB == 2) {
            dw_f32x2 v = __builtin_bit_cast(dw_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_d, off, 0, 0));
            b[0] = v[0]; b[1] = v[1];
        } else {
            b[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, off, 0, 0));
        }
    };
    auto mma = [&](const float (&a)[VA], const float (&b)[VB]) {
#pragma unroll
        for (int ta = 0; ta < VA; ++ta)
#pragma unroll
            for (int tb = 0; tb < VB; ++tb) acc[ta][tb] = DW_MFMA(a[ta], b[tb], acc[ta][tb]);
    };

    int buf = 0;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 64, buf ^= 1) {
        const int64_t o_mine = r0 + lane;
        const int idx64 = (o_mine < r_end) ? (nbr ? nbr[(int64_t)k * n_out + o_mine] : (int)o_mine) : -1;
        const uint64_t m = __ballot(idx64 >= 0);
        if (m == 0ull) continue;
        const int cnt = __builtin_popcountll(m);
        if (idx64 >= 0) list[buf][__builtin_popcountll(m & ((1ull << lane) - 1ull))] = make_int2((int)(o_mine - r_begin), idx64);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): one wave, in-order LDS -- the list is complete
        const int ns = (cnt + 3) >> 2;
        // step s: lane group r takes list entry 4 s + r (past the end: a row that reads zeros)
        auto offsets = [&](int s, uint32_t& oa, uint32_t& ob) {
            const int e = 4 * s + lr;
            const int2 v = list[buf][e < cnt ? e : 0];
            const bool ok = e < cnt;
            oa = (ok && a_ok) ? (uint32_t)v.y * ldx4 + offa : 0xFFFFFFF0u;
            ob = (ok && b_ok) ? (uint32_t)(r_begin + v.x) * ldd4 + offb : 0xFFFFFFF0u;
        };
        float a0[VA], b0[VB], a1[VA], b1[VB];
        uint32_t oa, ob;
        offsets(0, oa, ob);
        load_a(oa, a0);
        load_b(ob, b0);
        int s = 0;
        for (; s + 2 <= ns; s += 2) {
            offsets(s + 1, oa, ob);
            load_a(oa, a1);
            load_b(ob, b1);
            mma(a0, b0);
            offsets(s + 2 < ns ? s + 2 : s + 1, oa, ob);  // (clamped re-request on the last pair)
            load_a(oa, a0);
            load_b(ob, b0);
            mma(a1, b1);
        }
        if (s < ns) mma(a0, b0);
    }
    // D[i = 4 * (lane >> 4) + reg][j = lane & 15] of tile (ta, tb) = dW[ci = block + VA * i + ta][co = block + VB * j + tb]
#pragma unroll
    for (int ta = 0; ta < VA; ++ta)
#pragma unroll
        for (int tb = 0; tb < VB; ++tb) {
            const int gco = co_b * 16 * VB + VB * li + tb;
            if (gco >= cout) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gci = ci_b * 16 * VA + VA * (4 * lr + reg) + ta;
                if (gci < cin) partial[(((int64_t)chunk * K + k) * cin + gci) * cout + gco] = acc[ta][tb][reg];
            }
        }
}

__global__ void k_conv_dw_reduce(const float* __restrict__ partial, int n_chunks, int64_t per_chunk, float* __restrict__ dw,
                                 int accumulate) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per_chunk) return;
    float acc = accumulate ? dw[t] : 0.f;
    for (int c = 0; c < n_chunks; ++c) acc += partial[(int64_t)c * per_chunk + t];  // fixed order
    dw[t] = acc;
}

// Two-level form for long chunk lists (round 4: a 1.9 M-row layer leaves ~3 700 partials per element, and the single loop above
// walked them serially in a few blocks -- k_conv_dw_reduce was 6 of the 62 ms of a training step).  Stage 1: group g of GROUPS
// sums its run of `len` consecutive chunks in order, in place into the run's first slot (a thread only ever touches its own
// element column); stage 2: the loop above over the GROUPS slots (stride len).  Fixed order: deterministic.
__global__ void k_reduce_chunk_groups(float* __restrict__ partial, int n_chunks, int64_t per_chunk, int len) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per_chunk) return;
    const int c0 = (int)blockIdx.y * len, c1 = min(c0 + len, n_chunks);
    if (c0 >= c1) return;
    float acc = partial[(int64_t)c0 * per_chunk + t];
    for (int c = c0 + 1; c < c1; ++c) acc += partial[(int64_t)c * per_chunk + t];
    partial[(int64_t)c0 * per_chunk + t] = acc;
}
__global__ void k_conv_dw_reduce_strided(const float* __restrict__ partial, int n_slots, int stride, int64_t per_chunk,
                                         float* __restrict__ dw, int accumulate) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per_chunk) return;
    float acc = accumulate ? dw[t] : 0.f;
    for (int c = 0; c < n_slots; ++c) acc += partial[(int64_t)c * stride * per_chunk + t];  // fixed order
    dw[t] = acc;
}
// dw[t] (+)= sum over the chunks' partials; `partial` is scratch (stage 1 overwrites slots of it)
static void launch_chunk_reduce(float* partial, int n_chunks, int64_t per_chunk, float* dw, int accumulate, hipStream_t s) {
    constexpr int GROUPS = 16;
    if (n_chunks <= 2 * GROUPS) {
        INSMOS_LAUNCH(k_conv_dw_reduce, dim3(cdiv(per_chunk, 256)), dim3(256), 0, s, partial, n_chunks, per_chunk, dw, accumulate);
        return;
    }
    const int len = (n_chunks + GROUPS - 1) / GROUPS;
    const int n_slots = (n_chunks + len - 1) / len;
    INSMOS_LAUNCH(k_reduce_chunk_groups, dim3(cdiv(per_chunk, 256), n_slots), dim3(256), 0, s, partial, n_chunks, per_chunk, len);
    INSMOS_LAUNCH(k_conv_dw_reduce_strided, dim3(cdiv(per_chunk, 256)), dim3(256), 0, s, partial, n_slots, len, per_chunk, dw, accumulate);
}

// ---- column sums (bias gradient): stage 1 per 1024-row block, stage 2 over the blocks ----
__global__ void __launch_bounds__(256) k_col_sum(const float* __restrict__ a, int ld, int c, int64_t n, int rows_per_block,
                                                  float* __restrict__ partial) {
    __shared__ float sm[256];
    const int col = blockIdx.y, tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, n);
    float acc = 0.f;
    for (int64_t r = r0 + tid; r < r1; r += 256) acc += a[r * ld + col];
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) partial[(int64_t)blockIdx.x * c + col] = sm[0];
}

// ---- MOSLoss (models/loss.py:20-34) ----
// term[i] = -w[g] * log(max(softmax_g, 1e-8)), wsum[i] = w[g]; grad[i][c] = d(sum term)/d logit[i][c] (unnormalised)
__global__ void k_mos_loss(const float* __restrict__ logits, int ld, const int64_t* __restrict__ gt, int64_t n, int ncls,
                           unsigned ignore_mask, const float* __restrict__ w, float* __restrict__ term,
                           float* __restrict__ wsum, float* __restrict__ grad, int ld_grad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* z = logits + i * ld;
    float m = -INFINITY;
    for (int c = 0; c < ncls; ++c)
        if (!((ignore_mask >> c) & 1u)) m = fmaxf(m, z[c]);
    float den = 0.f;
    for (int c = 0; c < ncls; ++c)
        if (!((ignore_mask >> c) & 1u)) den += expf(z[c] - m);
    const int g = (int)gt[i];
    const bool g_live = g >= 0 && g < ncls && !((ignore_mask >> g) & 1u);
    const float pg = g_live ? expf(z[g] - m) / den : 0.f;   // softmax of an ignored class is exactly 0
    const float wg = (g >= 0 && g < ncls) ? w[g] : 0.f;
    term[i] = -wg * logf(fmaxf(pg, 1e-8f));
    wsum[i] = wg;
    if (grad) {
        // d term / d z_c = -wg * [pg >= 1e-8] * (1/pg) * pg * (delta_gc - p_c) = -wg * [pg >= 1e-8] * (delta_gc - p_c)
        const float live = (g_live && pg >= 1e-8f) ? wg : 0.f;
        for (int c = 0; c < ncls; ++c) {
            float v = 0.f;
            if (!((ignore_mask >> c) & 1u)) {
                const float pc = expf(z[c] - m) / den;
                v = -live * ((c == g ? 1.f : 0.f) - pc);
            }
            grad[i * ld_grad + c] = v;
        }
    }
}

__global__ void k_scale_rows(float* __restrict__ a, int ld, int c, int64_t n, const float* __restrict__ sums) {
    // grad /= sum of weights (sums[1]); loss = sums[0] / sums[1] is finished on the host side of the call
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    const float inv = 1.0f / sums[1];
    a[(t / c) * ld + (t % c)] *= inv;
}


// ---- BatchNorm1d over sparse rows, training mode (spconv_unet.py:36-60 / minkunet.py blocks use nn.BatchNorm1d /
// MinkowskiBatchNorm in train()): per-channel batch statistics, normalise + affine (+ ReLU), and the backward ----
// stage 1: per 1024-row block and channel: sum(x), sum((x - shift)^2 ...) is avoided: two passes (mean first) keep fp32 exact enough
__global__ void __launch_bounds__(256) k_bn_partial(const float* __restrict__ a, int ld, const float* __restrict__ b, int ld_b,
                                                     const float* __restrict__ sub, int c, int64_t n, int mode,
                                                     float* __restrict__ partial) {
    // mode 0: sum a            mode 1: sum (a - sub[col])^2          mode 2: sum a*b (b = normalised x)
    __shared__ float sm[256];
    const int col = blockIdx.y, tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 1024, r1 = min(r0 + 1024, n);
    const float sh = (mode == 1) ? sub[col] : 0.f;
    float acc = 0.f;
    for (int64_t r = r0 + tid; r < r1; r += 256) {
        const float v = a[r * ld + col];
        acc += mode == 0 ? v : mode == 1 ? (v - sh) * (v - sh) : v * b[r * ld_b + col];
    }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) partial[(int64_t)blockIdx.x * c + col] = sm[0];
}

__global__ void k_bn_finish_stats(const float* __restrict__ part, int nb, int c, int64_t n, float* __restrict__ out, int mode,
                                  float eps) {
    // mode 0: out[col] = mean = sum / n      mode 1: out[col] = invstd = 1/sqrt(sum/n + eps), out[c + col] = biased var
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) acc += part[(int64_t)b * c + col];
    if (mode == 0) out[col] = acc / (float)n;
    else { const float var = acc / (float)n; out[col] = 1.0f / sqrtf(var + eps); out[c + col] = var; }
}

__global__ void k_bn_apply(const float* __restrict__ x, int ld_x, const float* __restrict__ mean, const float* __restrict__ invstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta, int c, int64_t n, int relu,
                           float* __restrict__ xhat, float* __restrict__ y, int ld_y) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    const int64_t r = t / c;
    const int col = (int)(t % c);
    const float h = (x[r * ld_x + col] - mean[col]) * invstd[col];
    if (xhat) xhat[r * c + col] = h;
    const float v = h * gamma[col] + beta[col];
    y[r * ld_y + col] = relu ? fmaxf(v, 0.f) : v;
}

// dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)); dy is first masked by the ReLU (y > 0) when relu != 0
__global__ void k_bn_mask_relu(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y, int c, int64_t n,
                               float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    const int64_t r = t / c;
    const int col = (int)(t % c);
    out[t] = y[r * ld_y + col] > 0.f ? dy[r * ld_dy + col] : 0.f;
}

__global__ void k_bn_backward_dx(const float* __restrict__ g, const float* __restrict__ xhat, const float* __restrict__ gamma,
                                 const float* __restrict__ invstd, const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                 int c, int64_t n, float* __restrict__ dx, int ld_dx) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * c) return;
    const int64_t r = t / c;
    const int col = (int)(t % c);
    const float inv_n = 1.0f / (float)n;
    dx[r * ld_dx + col] = gamma[col] * invstd[col] * (g[t] - dbeta[col] * inv_n - xhat[t] * (dgamma[col] * inv_n));
}


// ---- BatchNorm1d over sparse rows, SEGMENTED and fused (round 3) -----------------------------------------------------------
// A training step over B windows in one set of launches must keep the reference's statistics: models/models.py:313 walks the
// batch list item by item, so every BatchNorm sees ONE window's rows (batch statistics per item, running statistics updated
// item after item).  Rows of a window are one contiguous range in the 3D branch and B x 10 interleaved runs in the 4D branch
// (rows are (t * B + b)-major), so the kernels take a CHUNK TABLE: (row_start, row_end, segment) with <= 1024 rows of ONE segment
// per chunk, sorted by segment (seg_first[s] .. seg_first[s + 1]).  S = 1 with plain 1024-row chunks is the unsegmented layer.
// Fewer passes than the first BatchNorm kernels: forward = one statistics read (per-chunk mean and centred sum of squares, merged
// per segment with Chan's formula in fixed order, in double) + apply; backward = one read of (dy, y, x^) for both sums with the
// ReLU mask applied on the fly + dx.  Deterministic.
struct BnChunk { int r0, r1, seg, pad; };

// "Last block done": every block of a partial-sum kernel takes a ticket after publishing its partials; the block that draws the
// last one folds them (fixed table order: the result does not depend on which block that is) -- the merge costs no launch.
__device__ __forceinline__ bool last_block_done(int* __restrict__ counter, int total_blocks) {
    __shared__ int s_last;
    __threadfence();   // this block's partials are visible device-wide before its ticket is
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(counter, 1);
        s_last = (t == total_blocks - 1) ? 1 : 0;
        if (s_last) *counter = 0;   // (the plan's counter is ready for the next layer: launches of one stream are ordered)
    }
    __syncthreads();
    if (s_last) __threadfence();
    return s_last != 0;
}

// block = 16 row lanes x 16 channels; grid (chunks, ceil(c / 16)).  part[chunk][0][col] = chunk mean, [1][col] = centred M2.
// The last block merges every (segment, channel) with Chan's formula in table order (in double; 256 / (S c) threads share a pair,
// each over a contiguous part of the segment's chunk list, then one of them folds their aggregates in order) and moves the running
// statistics segment after segment.  stats[s] = [mean (c) | invstd (c) | biased var (c)].
__global__ void __launch_bounds__(256) k_bnseg_stats(const float* __restrict__ x, int ld, int c, const BnChunk* __restrict__ chunks,
                                                     float* __restrict__ part, int* __restrict__ counter,
                                                     const int* __restrict__ seg_first, const int* __restrict__ seg_rows, int S, float eps,
                                                     float* __restrict__ stats, float momentum, float* __restrict__ rmean,
                                                     float* __restrict__ rvar) {
    __shared__ float sm[16][17];
    __shared__ double sagg[256][3];
    const BnChunk ch = chunks[blockIdx.x];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.y * 16 + cl;
    const bool ok = col < c;
    float acc = 0.f;
    if (ok)
        for (int r = ch.r0 + rl; r < ch.r1; r += 16) acc += x[(int64_t)r * ld + col];
    sm[rl][cl] = acc;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += sm[q][cl];
    const float mean = tot / (float)(ch.r1 - ch.r0);
    __syncthreads();
    acc = 0.f;
    if (ok)
        for (int r = ch.r0 + rl; r < ch.r1; r += 16) {
            const float d = x[(int64_t)r * ld + col] - mean;
            acc += d * d;
        }
    sm[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && ok) {
        float m2 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) m2 += sm[q][cl];
        part[((int64_t)blockIdx.x * 2 + 0) * c + col] = mean;
        part[((int64_t)blockIdx.x * 2 + 1) * c + col] = m2;
    }
    if (!last_block_done(counter, (int)(gridDim.x * gridDim.y))) return;
    const int pairs = S * c;
    int tpp = 1;
    while (tpp * 2 * pairs <= 256) tpp *= 2;   // threads per (segment, channel) pair
    for (int p0 = 0; p0 < pairs; p0 += 256 / tpp) {
        const int pr = p0 + (int)threadIdx.x / tpp, sub = (int)threadIdx.x % tpp;
        double n = 0.0, mu = 0.0, m2 = 0.0;
        if (pr < pairs) {
            const int sgi = pr / c, cc = pr % c;
            const int q0 = seg_first[sgi], q1 = seg_first[sgi + 1];
            const int per = (q1 - q0 + tpp - 1) / tpp;
            const int a0 = q0 + sub * per, a1 = min(a0 + per, q1);
            for (int q = a0; q < a1; ++q) {
                const double nb = (double)(chunks[q].r1 - chunks[q].r0);
                const double mb = part[((int64_t)q * 2 + 0) * c + cc], m2b = part[((int64_t)q * 2 + 1) * c + cc];
                const double d = mb - mu, nt = n + nb;
                mu += d * nb / nt;
                m2 += m2b + d * d * n * nb / nt;
                n = nt;
            }
        }
        sagg[threadIdx.x][0] = n; sagg[threadIdx.x][1] = mu; sagg[threadIdx.x][2] = m2;
        __syncthreads();
        if (pr < pairs && sub == 0) {
            double N = 0.0, MU = 0.0, M2 = 0.0;
            for (int t = 0; t < tpp; ++t) {
                const double nb = sagg[threadIdx.x + t][0], mb = sagg[threadIdx.x + t][1], m2b = sagg[threadIdx.x + t][2];
                if (nb <= 0.0) continue;
                const double d = mb - MU, nt = N + nb;
                MU += d * nb / nt;
                M2 += m2b + d * d * N * nb / nt;
                N = nt;
            }
            const int sgi = pr / c, cc = pr % c;
            const float var = N > 0.0 ? (float)(M2 / N) : 0.f;
            float* st = stats + (int64_t)sgi * 3 * c;
            st[cc] = (float)MU;
            st[c + cc] = 1.0f / sqrtf(var + eps);
            st[2 * c + cc] = var;
        }
        __syncthreads();
    }
    if (rmean && rvar) {   // running statistics, item after item like the reference's sequential forwards (unbiased variance)
        __threadfence_block();
        for (int cc = threadIdx.x; cc < c; cc += 256) {
            float m = rmean[cc], v = rvar[cc];
            for (int sgi = 0; sgi < S; ++sgi) {
                const int n = seg_rows[sgi];
                if (n <= 0) continue;
                const float* st = stats + (int64_t)sgi * 3 * c;
                m = m * (1.0f - momentum) + momentum * st[cc];
                v = v * (1.0f - momentum) + momentum * (st[2 * c + cc] * ((float)n / (float)max(n - 1, 1)));
            }
            rmean[cc] = m;
            rvar[cc] = v;
        }
    }
}
// grid (chunks, 4): a quarter of the chunk's rows per block
__global__ void __launch_bounds__(256) k_bnseg_apply(const float* __restrict__ x, int ld_x, int c, const BnChunk* __restrict__ chunks,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int relu, float* __restrict__ xhat,
                                                     float* __restrict__ y, int ld_y) {
    const BnChunk ch = chunks[blockIdx.x];
    const float* st = stats + (int64_t)ch.seg * 3 * c;
    const int rows = ch.r1 - ch.r0, q = (rows + 3) / 4;
    const int ra = ch.r0 + (int)blockIdx.y * q, rb = min(ra + q, ch.r1);
    const int64_t total = (int64_t)max(rb - ra, 0) * c;
    for (int64_t t = threadIdx.x; t < total; t += blockDim.x) {
        const int64_t r = ra + t / c;
        const int col = (int)(t % c);
        const float h = (x[r * ld_x + col] - st[col]) * st[c + col];
        xhat[r * c + col] = h;
        const float v = h * gamma[col] + beta[col];
        y[r * ld_y + col] = relu ? fmaxf(v, 0.f) : v;
    }
}
// backward sums of one chunk: [0][col] = sum g, [1][col] = sum g * x^, g = dy masked by the ReLU (y > 0).  The last block folds
// them per segment (segsum[s] = [sum g (c) | sum g x^ (c)], table order) and over the segments (dbeta, dgamma).
__global__ void __launch_bounds__(256) k_bnseg_bwd_sums(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                        const float* __restrict__ xhat, int c, int relu,
                                                        const BnChunk* __restrict__ chunks, float* __restrict__ part,
                                                        int* __restrict__ counter, const int* __restrict__ seg_first, int S,
                                                        float* __restrict__ segsum, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta) {
    __shared__ float sa[16][17], sb[16][17];
    const BnChunk ch = chunks[blockIdx.x];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.y * 16 + cl;
    const bool ok = col < c;
    float a = 0.f, b = 0.f;
    if (ok)
        for (int r = ch.r0 + rl; r < ch.r1; r += 16) {
            float g = dy[(int64_t)r * ld_dy + col];
            if (relu && !(y[(int64_t)r * ld_y + col] > 0.f)) g = 0.f;
            a += g;
            b += g * xhat[(int64_t)r * c + col];
        }
    sa[rl][cl] = a;
    sb[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && ok) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { ta += sa[q][cl]; tb += sb[q][cl]; }
        part[((int64_t)blockIdx.x * 2 + 0) * c + col] = ta;
        part[((int64_t)blockIdx.x * 2 + 1) * c + col] = tb;
    }
    if (!last_block_done(counter, (int)(gridDim.x * gridDim.y))) return;
    for (int pr = threadIdx.x; pr < S * c; pr += 256) {   // per (segment, channel), chunks in table order
        const int sgi = pr / c, cc = pr % c;
        float ta = 0.f, tb = 0.f;
        for (int q = seg_first[sgi]; q < seg_first[sgi + 1]; ++q) {
            ta += part[((int64_t)q * 2 + 0) * c + cc];
            tb += part[((int64_t)q * 2 + 1) * c + cc];
        }
        segsum[((int64_t)sgi * 2 + 0) * c + cc] = ta;
        segsum[((int64_t)sgi * 2 + 1) * c + cc] = tb;
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < c; cc += 256) {   // dgamma / dbeta = sums over the segments (fixed order)
        float ta = 0.f, tb = 0.f;
        for (int sgi = 0; sgi < S; ++sgi) {
            ta += segsum[((int64_t)sgi * 2 + 0) * c + cc];
            tb += segsum[((int64_t)sgi * 2 + 1) * c + cc];
        }
        dbeta[cc] = ta;
        dgamma[cc] = tb;
    }
}
// dx = gamma * invstd_s * (g - mean_s(g) - x^ * mean_s(g x^)); grid (chunks, 4)
__global__ void __launch_bounds__(256) k_bnseg_bwd_dx(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                      const float* __restrict__ xhat, int c, int relu,
                                                      const BnChunk* __restrict__ chunks, const int* __restrict__ seg_rows,
                                                      const float* __restrict__ gamma, const float* __restrict__ stats,
                                                      const float* __restrict__ segsum, float* __restrict__ dx, int ld_dx) {
    const BnChunk ch = chunks[blockIdx.x];
    const float* st = stats + (int64_t)ch.seg * 3 * c;
    const float* sg = segsum + (int64_t)ch.seg * 2 * c;
    const float inv_n = 1.0f / (float)seg_rows[ch.seg];
    const int rows = ch.r1 - ch.r0, q = (rows + 3) / 4;
    const int ra = ch.r0 + (int)blockIdx.y * q, rb = min(ra + q, ch.r1);
    const int64_t total = (int64_t)max(rb - ra, 0) * c;
    for (int64_t t = threadIdx.x; t < total; t += blockDim.x) {
        const int64_t r = ra + t / c;
        const int col = (int)(t % c);
        float g = dy[r * ld_dy + col];
        if (relu && !(y[r * ld_y + col] > 0.f)) g = 0.f;
        dx[r * ld_dx + col] = gamma[col] * st[c + col] * (g - sg[col] * inv_n - xhat[r * c + col] * (sg[c + col] * inv_n));
    }
}


// ---- the same four kernels with 16-byte accesses (round 4).  The kernels above walk a chunk with 16 row lanes x 16 channels of
// scalar loads (half the lanes idle at c = 8, two integer divisions per element in the apply / dx loops) and ran at ~0.3 TB/s;
// BatchNorm was 18 of the 67 ms of a training step (profiles/r04_bench_cfg5.json).  Here a thread owns ONE group of four
// consecutive channels (c / 4 = 2^L4 groups per row, a power of two) and walks the chunk's rows with float4 loads: consecutive
// threads read consecutive 16-byte pieces.  Same outputs as the scalar kernels up to the summation order inside a chunk
// (fixed, deterministic); the cross-chunk merges are the same code.
typedef float bn_f4 __attribute__((ext_vector_type(4)));
constexpr int BN_U = 8;   // rows a thread of the statistics kernel keeps in flight

template <int NV>   // tree reduction of NV float4 per thread over the row lanes (threads t, t + G, t + 2G, ... share channel group t % G)
__device__ __forceinline__ void bn_reduce_rows(bn_f4 (&v)[NV], bn_f4* __restrict__ sm, int G) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[i * 256 + t] = v[i];
    __syncthreads();
    for (int sdist = 128; sdist >= G; sdist >>= 1) {
        if (t < sdist) {
#pragma unroll
            for (int i = 0; i < NV; ++i) sm[i * 256 + t] += sm[i * 256 + t + sdist];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sm[i * 256 + (t & (G - 1))];
    __syncthreads();
}

// rows [ra, rb) of quarter y of a chunk (the split of k_bnseg_apply_v4 / k_bnseg_bwd_dx_v4; a short chunk leaves quarters empty)
__device__ __forceinline__ void bn_quarter(const BnChunk& ch, int y, int& ra, int& rb) {
    const int rows = ch.r1 - ch.r0, q = (rows + 3) / 4;
    ra = ch.r0 + y * q;
    rb = min(ra + q, ch.r1);
    if (rb < ra) rb = ra;
}
__device__ __forceinline__ int bn_quarter_rows(const BnChunk& ch, int y) {
    int ra, rb;
    bn_quarter(ch, y, ra, rb);
    return rb - ra;
}

// Round 6: the two REDUCTION kernels (statistics, backward sums) no longer fold their partials in "the block that finishes last".
// Measured (tools/bn_kernel_probe.sh, tools/bn_camp_probe.sh): the ticket -- a device-scope fence + one atomic on ONE address per
// block -- costs ~0.1 us PER BLOCK, serialised (1.57 M x 8: 58 us on 384 blocks, 91-104 us on 768; the element-wise kernels of the
// same layer move twice the bytes in 16-24 us), and the merge itself ran in one block after all others (350 of the 430 us of the
// 256-channel deblock layer).  Now: grid (chunks, 4) -- one QUARTER of a chunk per block, partial slot = 4 chunk + quarter, no fence,
// no ticket -- and a second small launch that merges the slots in parallel (a block per few channel groups, 16-byte loads, eight
// slots in flight per thread, float64 sums without a division):
//   N = sum n_b,  A = sum n_b mean_b,  Q = sum (M2_b + n_b mean_b^2);   mean = A / N,  var = (Q - A mean) / N
// (float64 over fp32 partials: the cancellation in Q - A mean costs ~(mean / std)^2 x 1e-16 relative).
// grid (chunks, 4); block 256; G = c / 4 = 1 << L4 channel groups (G <= 64)
__global__ void __launch_bounds__(256) k_bnseg_stats_v4(const float* __restrict__ x, int ld, int c, int L4, const BnChunk* __restrict__ chunks,
                                                        float* __restrict__ part) {
    __shared__ bn_f4 smv[256];
    const BnChunk ch = chunks[blockIdx.x];
    int qa, qb;
    bn_quarter(ch, (int)blockIdx.y, qa, qb);
    const int pslot = (int)blockIdx.x * 4 + (int)blockIdx.y;
    const int G = 1 << L4, t = threadIdx.x;
    const int cg = t & (G - 1), rl = t >> L4, RL = 256 >> L4;
    // (a thread's rows are requested BN_U at a time and added in row order: the same sum chain as a one-row loop, with BN_U loads in
    //  flight per thread instead of one)
    bn_f4 acc[1] = {(bn_f4){0.f, 0.f, 0.f, 0.f}};
    {
        int r = qa + rl;
        for (; r + (BN_U - 1) * RL < qb; r += BN_U * RL) {
            bn_f4 v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) v[u] = *(const bn_f4*)(x + (int64_t)(r + u * RL) * ld + 4 * cg);
#pragma unroll
            for (int u = 0; u < BN_U; ++u) acc[0] += v[u];
        }
        for (; r < qb; r += RL) acc[0] += *(const bn_f4*)(x + (int64_t)r * ld + 4 * cg);
    }
    bn_reduce_rows<1>(acc, smv, G);
    const float inv = qb > qa ? 1.0f / (float)(qb - qa) : 0.f;
    const bn_f4 mean = acc[0] * inv;
    bn_f4 m2[1] = {(bn_f4){0.f, 0.f, 0.f, 0.f}};
    {
        int r = qa + rl;
        for (; r + (BN_U - 1) * RL < qb; r += BN_U * RL) {
            bn_f4 v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) v[u] = *(const bn_f4*)(x + (int64_t)(r + u * RL) * ld + 4 * cg);
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const bn_f4 d = v[u] - mean;
                m2[0] += d * d;
            }
        }
        for (; r < qb; r += RL) {
            const bn_f4 d = *(const bn_f4*)(x + (int64_t)r * ld + 4 * cg) - mean;
            m2[0] += d * d;
        }
    }
    bn_reduce_rows<1>(m2, smv, G);
    if (rl == 0) {
        *(bn_f4*)(part + ((int64_t)pslot * 2 + 0) * c + 4 * cg) = mean;
        *(bn_f4*)(part + ((int64_t)pslot * 2 + 1) * c + 4 * cg) = m2[0];
    }
}

// channel groups a merge block owns (all S segments of each: the running statistics / dgamma, dbeta of a channel need every segment)
__host__ __device__ inline int bn_merge_groups_per_block(int S) { return S >= 8 ? 1 : 8 / S; }   // (few pairs per block: 32+ threads walk one pair's slots)

// grid (ceil(G / GB)); block 256: `tpp` threads share a (channel group, segment) pair, each over a contiguous run of the segment's
// partial slots (table order), then one of them adds their sums in thread order: the result depends on (S, c, chunk table) only
__global__ void __launch_bounds__(256) k_bnseg_stats_merge_v4(const float* __restrict__ part, int c, int L4, const BnChunk* __restrict__ chunks,
                                                              const int* __restrict__ seg_first, const int* __restrict__ seg_rows, int S,
                                                              float eps, float* __restrict__ stats, float momentum,
                                                              float* __restrict__ rmean, float* __restrict__ rvar) {
    __shared__ double sagg[256][9];   // (n, sum n_b mean_b [4], sum M2_b + n_b mean_b^2 [4]) per thread
    const int G = 1 << L4, GB = bn_merge_groups_per_block(S);
    const int g_base = (int)blockIdx.x * GB, ng = min(GB, G - g_base);
    const int pairs = ng * S;   // pair p = (group g_base + p / S, segment p % S)
    int tpp = 1;
    while (tpp * 2 * pairs <= 256) tpp *= 2;
    for (int p0 = 0; p0 < pairs; p0 += 256 / tpp) {
        const int pr = p0 + (int)threadIdx.x / tpp, sub = (int)threadIdx.x % tpp;
        double n = 0.0, sa[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
        if (pr < pairs) {
            const int sgi = pr % S, g4 = g_base + pr / S;
            const int s0 = 4 * seg_first[sgi], s1 = 4 * seg_first[sgi + 1];
            const int per = (s1 - s0 + tpp - 1) / tpp;
            const int a0 = s0 + sub * per, a1 = min(a0 + per, s1);
            for (int q = a0; q < a1; q += 8) {
                bn_f4 pm[8], p2[8];
                int pn[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int qq = min(q + u, a1 - 1);
                    pn[u] = q + u < a1 ? bn_quarter_rows(chunks[qq >> 2], qq & 3) : 0;
                    pm[u] = *(const bn_f4*)(part + ((int64_t)qq * 2 + 0) * c + 4 * g4);
                    p2[u] = *(const bn_f4*)(part + ((int64_t)qq * 2 + 1) * c + 4 * g4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double nb = (double)pn[u];
                    n += nb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double mb = (double)pm[u][e];
                        sa[e] += nb * mb;
                        sq[e] += nb > 0.0 ? (double)p2[u][e] + nb * mb * mb : 0.0;
                    }
                }
            }
        }
        sagg[threadIdx.x][0] = n;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sagg[threadIdx.x][1 + e] = sa[e]; sagg[threadIdx.x][5 + e] = sq[e]; }
        __syncthreads();
        if (pr < pairs && sub == 0) {
            double N = 0.0, A[4] = {0.0, 0.0, 0.0, 0.0}, Q[4] = {0.0, 0.0, 0.0, 0.0};
            for (int q = 0; q < tpp; ++q) {
                N += sagg[threadIdx.x + q][0];
#pragma unroll
                for (int e = 0; e < 4; ++e) { A[e] += sagg[threadIdx.x + q][1 + e]; Q[e] += sagg[threadIdx.x + q][5 + e]; }
            }
            const int sgi = pr % S, g4 = g_base + pr / S;
            float* st = stats + (int64_t)sgi * 3 * c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cc = 4 * g4 + e;
                const double MU = N > 0.0 ? A[e] / N : 0.0;
                double V = N > 0.0 ? (Q[e] - A[e] * MU) / N : 0.0;
                if (V < 0.0) V = 0.0;
                const float var = (float)V;
                st[cc] = (float)MU;
                st[c + cc] = 1.0f / sqrtf(var + eps);
                st[2 * c + cc] = var;
            }
        }
        __syncthreads();
    }
    if (rmean && rvar) {   // (this block wrote every segment's statistics of its channels)
        __threadfence_block();
        __syncthreads();
        for (int cl = threadIdx.x; cl < 4 * ng; cl += 256) {
            const int cc = 4 * g_base + cl;
            float m = rmean[cc], v = rvar[cc];
            for (int sgi = 0; sgi < S; ++sgi) {
                const int n = seg_rows[sgi];
                if (n <= 0) continue;
                const float* st = stats + (int64_t)sgi * 3 * c;
                m = m * (1.0f - momentum) + momentum * st[cc];
                v = v * (1.0f - momentum) + momentum * (st[2 * c + cc] * ((float)n / (float)max(n - 1, 1)));
            }
            rmean[cc] = m;
            rvar[cc] = v;
        }
    }
}

// grid (chunks, 4): a quarter of the chunk's rows per block
__global__ void __launch_bounds__(256) k_bnseg_apply_v4(const float* __restrict__ x, int ld_x, int c, int L4, const BnChunk* __restrict__ chunks,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu, float* __restrict__ xhat,
                                                        float* __restrict__ y, int ld_y) {
    const BnChunk ch = chunks[blockIdx.x];
    const float* st = stats + (int64_t)ch.seg * 3 * c;
    const int rows = ch.r1 - ch.r0, q = (rows + 3) / 4;
    const int ra = ch.r0 + (int)blockIdx.y * q, rb = min(ra + q, ch.r1);
    const int G = 1 << L4, cg = threadIdx.x & (G - 1), rl = threadIdx.x >> L4, RL = 256 >> L4;
    const bn_f4 mu = *(const bn_f4*)(st + 4 * cg), is = *(const bn_f4*)(st + c + 4 * cg);
    const bn_f4 ga = *(const bn_f4*)(gamma + 4 * cg), be = *(const bn_f4*)(beta + 4 * cg);
    for (int r = ra + rl; r < rb; r += RL) {
        const bn_f4 h = (*(const bn_f4*)(x + (int64_t)r * ld_x + 4 * cg) - mu) * is;
        if (xhat) *(bn_f4*)(xhat + (int64_t)r * c + 4 * cg) = h;   // (null: the backward recomputes it from x -- RECOMP below)
        bn_f4 v = h * ga + be;
        if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        *(bn_f4*)(y + (int64_t)r * ld_y + 4 * cg) = v;
    }
}

// RECOMP (round 6): the forward kept the layer's INPUT x instead of writing x^ (one store pass less), so `xhat` = x with pitch ld_y here
// and x^ = (x - mean) * invstd is recomputed -- the apply kernel's expression, the same bits --; the ReLU mask comes from
// gamma * x^ + beta > 0 (what y > 0 says, y = max(gamma * x^ + beta, 0)) instead of a read of y: 9 instead of 12 passes over the
// layer's elements per forward + backward.  `y` then carries nothing: RX = {stats, gamma, beta} ride along.
struct BnRecomp { const float* stats; const float* gamma; const float* beta; };
template <bool RECOMP>
__global__ void __launch_bounds__(256) k_bnseg_bwd_sums_v4(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                           const float* __restrict__ xhat, int c, int L4, int relu,
                                                           const BnChunk* __restrict__ chunks, float* __restrict__ part, BnRecomp RX) {
    // grid (chunks, 4): one quarter of a chunk per block, partial slot = 4 chunk + quarter; k_bnseg_sums_merge_v4 folds the slots
    __shared__ bn_f4 smv[512];
    const BnChunk ch = chunks[blockIdx.x];
    int qa, qb;
    bn_quarter(ch, (int)blockIdx.y, qa, qb);
    const int pslot = (int)blockIdx.x * 4 + (int)blockIdx.y;
    const int G = 1 << L4, t = threadIdx.x;
    const int cg = t & (G - 1), rl = t >> L4, RL = 256 >> L4;
    bn_f4 ab[2] = {(bn_f4){0.f, 0.f, 0.f, 0.f}, (bn_f4){0.f, 0.f, 0.f, 0.f}};
    if (RECOMP) {
        const float* st = RX.stats + (int64_t)ch.seg * 3 * c;
        const bn_f4 mu = *(const bn_f4*)(st + 4 * cg), is = *(const bn_f4*)(st + c + 4 * cg);
        const bn_f4 ga = *(const bn_f4*)(RX.gamma + 4 * cg), be = *(const bn_f4*)(RX.beta + 4 * cg);
        constexpr int BN_UB = 8;
        auto fold = [&](bn_f4 g, bn_f4 xr) {
            const bn_f4 h = (xr - mu) * is;
            if (relu) {
                const bn_f4 v = h * ga + be;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(v[i] > 0.f)) g[i] = 0.f;
            }
            ab[0] += g;
            ab[1] += g * h;
        };
        int r = qa + rl;
        for (; r + (BN_UB - 1) * RL < qb; r += BN_UB * RL) {
            bn_f4 g[BN_UB], xr[BN_UB];
#pragma unroll
            for (int u = 0; u < BN_UB; ++u) {
                g[u] = *(const bn_f4*)(dy + (int64_t)(r + u * RL) * ld_dy + 4 * cg);
                xr[u] = *(const bn_f4*)(xhat + (int64_t)(r + u * RL) * ld_y + 4 * cg);
            }
#pragma unroll
            for (int u = 0; u < BN_UB; ++u) fold(g[u], xr[u]);
        }
        for (; r < qb; r += RL)
            fold(*(const bn_f4*)(dy + (int64_t)r * ld_dy + 4 * cg), *(const bn_f4*)(xhat + (int64_t)r * ld_y + 4 * cg));
    } else {
        // (BN_UB rows' (dy, y, x^) requested together, folded in row order: the sums' chains are those of a one-row loop)
        constexpr int BN_UB = 4;
        int r = qa + rl;
        for (; r + (BN_UB - 1) * RL < qb; r += BN_UB * RL) {
            bn_f4 g[BN_UB], yy[BN_UB], xh[BN_UB];
#pragma unroll
            for (int u = 0; u < BN_UB; ++u) {
                g[u] = *(const bn_f4*)(dy + (int64_t)(r + u * RL) * ld_dy + 4 * cg);
                xh[u] = *(const bn_f4*)(xhat + (int64_t)(r + u * RL) * c + 4 * cg);
            }
            if (relu) {
#pragma unroll
                for (int u = 0; u < BN_UB; ++u) yy[u] = *(const bn_f4*)(y + (int64_t)(r + u * RL) * ld_y + 4 * cg);
            }
#pragma unroll
            for (int u = 0; u < BN_UB; ++u) {
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (!(yy[u][i] > 0.f)) g[u][i] = 0.f;
                }
                ab[0] += g[u];
                ab[1] += g[u] * xh[u];
            }
        }
        for (; r < qb; r += RL) {
            bn_f4 g = *(const bn_f4*)(dy + (int64_t)r * ld_dy + 4 * cg);
            if (relu) {
                const bn_f4 yy = *(const bn_f4*)(y + (int64_t)r * ld_y + 4 * cg);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(yy[i] > 0.f)) g[i] = 0.f;
            }
            ab[0] += g;
            ab[1] += g * *(const bn_f4*)(xhat + (int64_t)r * c + 4 * cg);
        }
    }
    bn_reduce_rows<2>(ab, smv, G);
    if (rl == 0) {
        *(bn_f4*)(part + ((int64_t)pslot * 2 + 0) * c + 4 * cg) = ab[0];
        *(bn_f4*)(part + ((int64_t)pslot * 2 + 1) * c + 4 * cg) = ab[1];
    }
}

// grid (ceil(G / GB)); block 256 -- see k_bnseg_stats_merge_v4; per channel the slots' sums are added in table order
__global__ void __launch_bounds__(256) k_bnseg_sums_merge_v4(const float* __restrict__ part, int c, int L4, const int* __restrict__ seg_first,
                                                             int S, float* __restrict__ segsum, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    __shared__ bn_f4 sa4[256], sb4[256];
    const int G = 1 << L4, GB = bn_merge_groups_per_block(S);
    const int g_base = (int)blockIdx.x * GB, ng = min(GB, G - g_base);
    const int pairs = ng * S;
    int tpp = 1;
    while (tpp * 2 * pairs <= 256) tpp *= 2;
    for (int p0 = 0; p0 < pairs; p0 += 256 / tpp) {
        const int pr = p0 + (int)threadIdx.x / tpp, sub = (int)threadIdx.x % tpp;
        bn_f4 ta = (bn_f4){0.f, 0.f, 0.f, 0.f}, tb = ta;
        if (pr < pairs) {
            const int sgi = pr % S, g4 = g_base + pr / S;
            const int s0 = 4 * seg_first[sgi], s1 = 4 * seg_first[sgi + 1];
            const int per = (s1 - s0 + tpp - 1) / tpp;
            const int a0 = s0 + sub * per, a1 = min(a0 + per, s1);
            int q = a0;
            for (; q + 8 <= a1; q += 8) {
                bn_f4 pa[8], pb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    pa[u] = *(const bn_f4*)(part + ((int64_t)(q + u) * 2 + 0) * c + 4 * g4);
                    pb[u] = *(const bn_f4*)(part + ((int64_t)(q + u) * 2 + 1) * c + 4 * g4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { ta += pa[u]; tb += pb[u]; }
            }
            for (; q < a1; ++q) {
                ta += *(const bn_f4*)(part + ((int64_t)q * 2 + 0) * c + 4 * g4);
                tb += *(const bn_f4*)(part + ((int64_t)q * 2 + 1) * c + 4 * g4);
            }
        }
        sa4[threadIdx.x] = ta;
        sb4[threadIdx.x] = tb;
        __syncthreads();
        if (pr < pairs && sub == 0) {
            bn_f4 xa = (bn_f4){0.f, 0.f, 0.f, 0.f}, xb = xa;
            for (int q = 0; q < tpp; ++q) { xa += sa4[threadIdx.x + q]; xb += sb4[threadIdx.x + q]; }
            const int sgi = pr % S, g4 = g_base + pr / S;
            *(bn_f4*)(segsum + ((int64_t)sgi * 2 + 0) * c + 4 * g4) = xa;
            *(bn_f4*)(segsum + ((int64_t)sgi * 2 + 1) * c + 4 * g4) = xb;
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    for (int cl = threadIdx.x; cl < 4 * ng; cl += 256) {   // (this block wrote every segment's sums of its channels)
        const int cc = 4 * g_base + cl;
        float ta = 0.f, tb = 0.f;
        for (int sgi = 0; sgi < S; ++sgi) {
            ta += segsum[((int64_t)sgi * 2 + 0) * c + cc];
            tb += segsum[((int64_t)sgi * 2 + 1) * c + cc];
        }
        dbeta[cc] = ta;
        dgamma[cc] = tb;
    }
}

template <bool RECOMP>
__global__ void __launch_bounds__(256) k_bnseg_bwd_dx_v4(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                         const float* __restrict__ xhat, int c, int L4, int relu,
                                                         const BnChunk* __restrict__ chunks, const int* __restrict__ seg_rows,
                                                         const float* __restrict__ gamma, const float* __restrict__ stats,
                                                         const float* __restrict__ segsum, float* __restrict__ dx, int ld_dx,
                                                         const float* __restrict__ beta) {
    const BnChunk ch = chunks[blockIdx.x];
    const float* st = stats + (int64_t)ch.seg * 3 * c;
    const float* sg = segsum + (int64_t)ch.seg * 2 * c;
    const float inv_n = 1.0f / (float)seg_rows[ch.seg];
    const int rows = ch.r1 - ch.r0, q = (rows + 3) / 4;
    const int ra = ch.r0 + (int)blockIdx.y * q, rb = min(ra + q, ch.r1);
    const int G = 1 << L4, cg = threadIdx.x & (G - 1), rl = threadIdx.x >> L4, RL = 256 >> L4;
    const bn_f4 gi = *(const bn_f4*)(gamma + 4 * cg) * *(const bn_f4*)(st + c + 4 * cg);
    const bn_f4 s0 = *(const bn_f4*)(sg + 4 * cg) * inv_n, s1 = *(const bn_f4*)(sg + c + 4 * cg) * inv_n;
    const bn_f4 ga = *(const bn_f4*)(gamma + 4 * cg), is = *(const bn_f4*)(st + c + 4 * cg);
    (void)gi;
    if (RECOMP) {   // (xhat = the layer's input x, pitch ld_y; see k_bnseg_bwd_sums_v4)
        const bn_f4 mu = *(const bn_f4*)(st + 4 * cg), be = *(const bn_f4*)(beta + 4 * cg);
        for (int r = ra + rl; r < rb; r += RL) {
            bn_f4 g = *(const bn_f4*)(dy + (int64_t)r * ld_dy + 4 * cg);
            const bn_f4 xh = (*(const bn_f4*)(xhat + (int64_t)r * ld_y + 4 * cg) - mu) * is;
            if (relu) {
                const bn_f4 v = xh * ga + be;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(v[i] > 0.f)) g[i] = 0.f;
            }
            *(bn_f4*)(dx + (int64_t)r * ld_dx + 4 * cg) = ga * is * (g - s0 - xh * s1);
        }
        return;
    }
    for (int r = ra + rl; r < rb; r += RL) {
        bn_f4 g = *(const bn_f4*)(dy + (int64_t)r * ld_dy + 4 * cg);
        if (relu) {
            const bn_f4 yy = *(const bn_f4*)(y + (int64_t)r * ld_y + 4 * cg);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!(yy[i] > 0.f)) g[i] = 0.f;
        }
        const bn_f4 xh = *(const bn_f4*)(xhat + (int64_t)r * c + 4 * cg);
        // the scalar kernel's expression, element for element: gamma * invstd * (g - sum_g / n - x^ * (sum_gx / n))
        *(bn_f4*)(dx + (int64_t)r * ld_dx + 4 * cg) = ga * is * (g - s0 - xh * s1);
    }
}

}  // namespace insmos

using namespace insmos;

static void chunking_dev(int cin, int& n16, int& has8, int& has4) {
    n16 = cin / 16;
    int rem = cin % 16;
    has8 = (rem & 8) ? 1 : 0;
    has4 = (rem & 4) ? 1 : 0;
}

extern "C" int insmos_pack_weights_device(const float* taps, int K, int cin_real, int cout_real, int cin, int cout,
                                          int transpose, int mirror_taps, float* packed, void* stream) {
    if (!taps || !packed || K <= 0 || cin % 4 != 0 || cin < (transpose ? cout_real : cin_real) ||
        cout < (transpose ? cin_real : cout_real))
        return INSMOS_EINVAL;
    int n16, h8, h4;
    chunking_dev(cin, n16, h8, h4);
    const int nblk = n16 + h8 + h4, ntile = (cout + 15) / 16;
    const int64_t total = (int64_t)K * nblk * ntile * 256;
    hipStream_t s = (hipStream_t)stream;
    INSMOS_LAUNCH(k_pack_weights, dim3(cdiv(total, 256)), dim3(256), 0, s, taps, K, cin_real, cout_real, cin, cout, nblk, ntile,
                  n16, h8, transpose, mirror_taps, packed);
    if (const size_t tail = rowlane_tail_floats(K, cin, cout)) {
        float* tp = packed + (insmos_packed_weight_floats(K, cin, cout) - tail);
        INSMOS_LAUNCH(k_pack_rowlane_tail, dim3(cdiv((int64_t)tail, 256)), dim3(256), 0, s, taps, K, cin_real, cout_real, cin, cout,
                      transpose, mirror_taps, tp);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// dW kernel: 2 = k_conv_dw_rows (default), 1 = k_conv_dw_mfma (first MFMA design; INSMOS_DW_MFMA=1), 0 = k_conv_dw (LDS slabs)
static int g_dw_mode = -1;
static int dw_mode() {
    if (g_dw_mode < 0) {
        const char* e = getenv("INSMOS_DW_KERNEL");
        const char* m = getenv("INSMOS_DW_MFMA");
        g_dw_mode = e ? atoi(e) : (m && m[0] == '1') ? 1 : 2;
        if (g_dw_mode < 0 || g_dw_mode > 2) g_dw_mode = 2;
    }
    return g_dw_mode;
}
extern "C" int insmos_debug_dw_kernel(int mode) {
    if (mode < 0 || mode > 2) return INSMOS_EINVAL;
    g_dw_mode = mode;
    return INSMOS_OK;
}

static int dw_vec(int c) { return c <= 16 ? 1 : c <= 32 ? 2 : 4; }

// rows per chunk (one partial dW per chunk): 4096 for the first two kernels; the row-compacting kernel takes smaller chunks on
// small layers so that a launch has a few thousand waves
static int dw_chunks(int64_t n_out, int K, int cin, int cout, int* rows_per_chunk) {
    int rpc = 4096;
    if (dw_mode() == 2) {
        const int64_t nblk = (int64_t)cdiv(cin, 16 * dw_vec(cin)) * cdiv(cout, 16 * dw_vec(cout)) * K;
        while (rpc > 512 && ((n_out + rpc - 1) / rpc) * nblk < 4096) rpc >>= 1;
    }
    *rows_per_chunk = rpc;
    return (int)((n_out + rpc - 1) / rpc);
}

extern "C" size_t insmos_sparse_conv_backward_weight_ws_floats(int64_t n_out, int K, int cin, int cout) {
    int rpc;
    return (size_t)dw_chunks(n_out, K, cin, cout, &rpc) * (size_t)K * (size_t)cin * (size_t)cout;
}

extern "C" int insmos_sparse_conv_backward_weight(const float* x, int64_t n_in, int ld_x, int cin, const float* dy, int ld_dy,
                                                  int cout, const int32_t* nbr, int K, int64_t n_out, float* dw,
                                                  int accumulate, float* ws, void* stream) {
    if (n_out <= 0) return INSMOS_OK;
    if (!x || !dy || !dw || !ws || cin <= 0 || cout <= 0 || K <= 0 || ld_x < cin || ld_dy < cout || n_in <= 0 ||
        (!nbr && (K != 1 || n_in < n_out)))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int rpc;
    const int nch = dw_chunks(n_out, K, cin, cout, &rpc);
    const int n_ci = (cin + 15) / 16, n_co = (cout + 15) / 16;
    ProfScope ps(KK_SPARSE_CONV, s);
    const int mode = dw_mode();
    // the row-compacting kernel addresses both operands through 32-bit buffer offsets and loads VA / VB channels per lane
    const int64_t xb = ((int64_t)n_in - 1) * ld_x * 4 + (int64_t)cin * 4, db = ((int64_t)n_out - 1) * ld_dy * 4 + (int64_t)cout * 4;
    int va = dw_vec(cin), vb = dw_vec(cout);
    // a lane loads va (vb) consecutive channels and is all-or-nothing (a_ok / b_ok in the kernel): the width must divide the
    // channel count, or the trailing channels of e.g. cin = 17 / 34 would silently get dW = 0
    while (va > 1 && (cin % va != 0 || ld_x % va != 0 || ((uintptr_t)x & (va * 4 - 1)))) va >>= 1;
    while (vb > 1 && (cout % vb != 0 || ld_dy % vb != 0 || ((uintptr_t)dy & (vb * 4 - 1)))) vb >>= 1;
    // (the chunk plan above assumed dw_vec(): a narrower vector only means more blocks per chunk)
    if (mode == 2 && xb < (1ll << 31) && db < (1ll << 31)) {
        const int n_cib = cdiv(cin, 16 * va), n_cob = cdiv(cout, 16 * vb);
        const dim3 grid(nch, K, n_cib * n_cob);
#define DW_GO(A, B)                                                                                                              \
    INSMOS_LAUNCH((k_conv_dw_rows<A, B>), grid, dim3(64), 0, s, x, (uint32_t)xb, ld_x, dy, (uint32_t)db, ld_dy, nbr, n_out, cin, cout, \
                  rpc, n_cob, ws, K)
#define DW_B(A) do { if (vb == 4) DW_GO(A, 4); else if (vb == 2) DW_GO(A, 2); else DW_GO(A, 1); } while (0)
        if (va == 4) DW_B(4); else if (va == 2) DW_B(2); else DW_B(1);
#undef DW_B
#undef DW_GO
    } else if (mode >= 1) {
        // co tiles per wave: as many as the layer has, up to 8 (32 accumulator registers)
        const int nt = n_co >= 8 ? 8 : n_co >= 4 ? 4 : n_co >= 2 ? 2 : 1;
        const int n_cg = (n_co + nt - 1) / nt;
        const dim3 grid(nch, K, n_ci * n_cg);
        if (nt == 8) INSMOS_LAUNCH(k_conv_dw_mfma<8>, grid, dim3(64), 0, s, x, ld_x, dy, ld_dy, nbr, n_out, cin, cout, rpc, n_cg, ws, K);
        else if (nt == 4) INSMOS_LAUNCH(k_conv_dw_mfma<4>, grid, dim3(64), 0, s, x, ld_x, dy, ld_dy, nbr, n_out, cin, cout, rpc, n_cg, ws, K);
        else if (nt == 2) INSMOS_LAUNCH(k_conv_dw_mfma<2>, grid, dim3(64), 0, s, x, ld_x, dy, ld_dy, nbr, n_out, cin, cout, rpc, n_cg, ws, K);
        else INSMOS_LAUNCH(k_conv_dw_mfma<1>, grid, dim3(64), 0, s, x, ld_x, dy, ld_dy, nbr, n_out, cin, cout, rpc, n_cg, ws, K);
    } else
    INSMOS_LAUNCH(k_conv_dw, dim3(nch, K, n_ci * n_co), dim3(256), 0, s, x, ld_x, dy, ld_dy, nbr, n_out, cin, cout, rpc, n_co, ws,
                  K);
    const int64_t per = (int64_t)K * cin * cout;
    launch_chunk_reduce(ws, nch, per, dw, accumulate, s);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" size_t insmos_col_sum_ws_floats(int64_t n, int c) { return (size_t)((n + 1023) / 1024) * (size_t)c; }

extern "C" int insmos_col_sum(const float* a, int ld, int c, int64_t n, float* out, int accumulate, float* ws, void* stream) {
    if (c <= 0) return INSMOS_OK;
    if (!a || !out || !ws || ld < c || n < 0) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)((n + 1023) / 1024);
    if (nb > 0) INSMOS_LAUNCH(k_col_sum, dim3(nb, c), dim3(256), 0, s, a, ld, c, n, 1024, ws);
    launch_chunk_reduce(ws, nb, (int64_t)c, out, accumulate, s);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" size_t insmos_mos_loss_ws_floats(int64_t n) { return (size_t)2 * (size_t)n + 2 * (size_t)((n + 1023) / 1024) + 16; }

extern "C" int insmos_mos_loss(const float* logits, int ld, const int64_t* gt, int64_t n, int ncls, unsigned ignore_mask,
                               const float* class_weights, float* loss_sums, float* grad, int ld_grad, float* ws,
                               void* stream) {
    if (n <= 0 || !logits || !gt || !class_weights || !loss_sums || !ws || ncls <= 0 || ncls > 32 || ld < ncls ||
        (grad && ld_grad < ncls))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* term = ws;            // (n) then wsum (n): column-summed as an (n, 2)-strided pair via two calls
    float* wsum = ws + n;
    float* cs = ws + 2 * n;
    ProfScope ps(KK_CONFUSION, s);
    INSMOS_LAUNCH(k_mos_loss, dim3(cdiv(n, 256)), dim3(256), 0, s, logits, ld, gt, n, ncls, ignore_mask, class_weights, term,
                  wsum, grad, ld_grad);
    int rc = insmos_col_sum(term, 1, 1, n, loss_sums, 0, cs, stream);       // loss_sums[0] = sum of weighted terms
    if (rc) return rc;
    rc = insmos_col_sum(wsum, 1, 1, n, loss_sums + 1, 0, cs, stream);       // loss_sums[1] = sum of weights
    if (rc) return rc;
    if (grad) INSMOS_LAUNCH(k_scale_rows, dim3(cdiv(n * ncls, 256)), dim3(256), 0, s, grad, ld_grad, ncls, n, loss_sums);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// stats: [mean (c) | invstd (c) | biased var (c)]; xhat (n, c) dense; ws: (n/1024 + 1) * c floats
extern "C" size_t insmos_batchnorm_ws_floats(int64_t n, int c) { return (size_t)((n + 1023) / 1024 + 1) * (size_t)c; }

extern "C" int insmos_batchnorm_train_forward(const float* x, int ld_x, int c, int64_t n, const float* gamma, const float* beta,
                                              float eps, int relu, float* y, int ld_y, float* xhat, float* stats, float* ws,
                                              void* stream) {
    if (n <= 0 || c <= 0) return INSMOS_OK;
    if (!x || !gamma || !beta || !y || !stats || !ws || ld_x < c || ld_y < c) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)((n + 1023) / 1024);
    float *mean = stats, *invstd = stats + c;
    ProfScope ps(KK_BATCHNORM, s);
    INSMOS_LAUNCH(k_bn_partial, dim3(nb, c), dim3(256), 0, s, x, ld_x, (const float*)nullptr, 0, (const float*)nullptr, c, n, 0, ws);
    INSMOS_LAUNCH(k_bn_finish_stats, dim3(cdiv(c, 64)), dim3(64), 0, s, ws, nb, c, n, mean, 0, eps);
    INSMOS_LAUNCH(k_bn_partial, dim3(nb, c), dim3(256), 0, s, x, ld_x, (const float*)nullptr, 0, mean, c, n, 1, ws);
    INSMOS_LAUNCH(k_bn_finish_stats, dim3(cdiv(c, 64)), dim3(64), 0, s, ws, nb, c, n, invstd, 1, eps);
    INSMOS_LAUNCH(k_bn_apply, dim3(cdiv(n * c, 256)), dim3(256), 0, s, x, ld_x, mean, invstd, gamma, beta, c, n, relu, xhat, y, ld_y);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// ---- segmented BatchNorm (see k_bnseg_stats): chunks (n_chunks x 4 int32: row_start, row_end, segment, 0) sorted by segment,
// seg_first (S + 1 int32, chunk index ranges), seg_rows (S int32, rows per segment), ticket (one int32, zero: the last-block counter,
// left zero) are DEVICE arrays; stats: S x 3c floats
// ([mean | invstd | biased var] per segment); ws: insmos_batchnorm_seg_ws_floats.  running_mean / running_var (optional) are updated
// segment after segment (momentum, unbiased variance), as the reference's item-by-item forwards do.
// log2(c / 4) when the 16-byte kernels apply (c / 4 a power of two <= 64, every pitch a multiple of 4 floats, every pointer
// 16-byte aligned; INSMOS_BN_VEC=0 keeps the scalar kernels), else -1
static int bn_vec_log2(int c, int ld_a, int ld_b, int ld_c, const void* p0, const void* p1, const void* p2, const void* p3, const void* p4) {
    static const bool on = [] { const char* e = getenv("INSMOS_BN_VEC"); return !(e && e[0] == '0'); }();
    if (!on || c < 4 || (c & 3) || ((c / 4) & (c / 4 - 1)) || c / 4 > 64 || (ld_a & 3) || (ld_b & 3) || (ld_c & 3)) return -1;
    if ((((uintptr_t)p0) | ((uintptr_t)p1) | ((uintptr_t)p2) | ((uintptr_t)p3) | ((uintptr_t)p4)) & 15) return -1;
    int l = 0;
    while ((4 << l) < c) ++l;
    return l;
}

extern "C" size_t insmos_batchnorm_seg_ws_floats(int n_chunks, int c, int S) {
    // (partials: 2 rows of c floats per chunk QUARTER -- the 16-byte reduction kernels run a block per quarter)
    return (size_t)8 * (size_t)(n_chunks > 0 ? n_chunks : 1) * (size_t)c + (size_t)2 * (size_t)(S > 0 ? S : 1) * (size_t)c + 64;
}

extern "C" int insmos_batchnorm_seg_forward(const float* x, int ld_x, int c, int64_t n, const int32_t* chunks, int n_chunks,
                                            const int32_t* seg_first, const int32_t* seg_rows, int S, const float* gamma,
                                            const float* beta, float eps, int relu, float* y, int ld_y, float* xhat, float* stats,
                                            float* running_mean, float* running_var, float momentum, int32_t* ticket, float* ws,
                                            void* stream) {
    if (n <= 0 || c <= 0 || n_chunks <= 0) return INSMOS_OK;
    if (!x || !chunks || !seg_first || !seg_rows || S < 1 || !gamma || !beta || !y || !stats || !ws || !ticket || ld_x < c ||
        ld_y < c || n >= (1ll << 31))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const BnChunk* ch = (const BnChunk*)chunks;
    const int l4 = bn_vec_log2(c, ld_x, ld_y, c, x, y, xhat ? xhat : y, gamma, beta);
    // xhat NULL: x^ is not stored -- the caller keeps x and takes insmos_batchnorm_seg_backward_x (16-byte kernels only:
    // insmos_batchnorm_seg_recompute_ok says whether a shape qualifies)
    if (!xhat && l4 < 0) return INSMOS_EINVAL;
    ProfScope ps(KK_BATCHNORM, s);
    if (l4 >= 0) {
        const int mg = (int)cdiv((int64_t)(1 << l4), bn_merge_groups_per_block(S));
        INSMOS_LAUNCH(k_bnseg_stats_v4, dim3(n_chunks, 4), dim3(256), 0, s, x, ld_x, c, l4, ch, ws);
        INSMOS_LAUNCH(k_bnseg_stats_merge_v4, dim3(mg), dim3(256), 0, s, ws, c, l4, ch, seg_first, seg_rows, S, eps, stats, momentum,
                      running_mean, running_var);
        INSMOS_LAUNCH(k_bnseg_apply_v4, dim3(n_chunks, 4), dim3(256), 0, s, x, ld_x, c, l4, ch, stats, gamma, beta, relu, xhat, y, ld_y);
        HIP_TRY(hipGetLastError());
        return INSMOS_OK;
    }
    INSMOS_LAUNCH(k_bnseg_stats, dim3(n_chunks, cdiv(c, 16)), dim3(256), 0, s, x, ld_x, c, ch, ws, ticket, seg_first, seg_rows, S, eps, stats,
                  momentum, running_mean, running_var);
    INSMOS_LAUNCH(k_bnseg_apply, dim3(n_chunks, 4), dim3(256), 0, s, x, ld_x, c, ch, stats, gamma, beta, relu, xhat, y, ld_y);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_batchnorm_seg_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* xhat, int c, int64_t n,
                                             const int32_t* chunks, int n_chunks, const int32_t* seg_first, const int32_t* seg_rows,
                                             int S, const float* gamma, const float* stats, int relu, float* dx, int ld_dx,
                                             float* dgamma, float* dbeta, int32_t* ticket, float* ws, void* stream) {
    if (n <= 0 || c <= 0 || n_chunks <= 0) return INSMOS_OK;
    if (!dy || !xhat || !chunks || !seg_first || !seg_rows || S < 1 || !gamma || !stats || !dx || !dgamma || !dbeta || !ws || !ticket ||
        (relu && !y) || ld_dy < c || ld_dx < c)
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const BnChunk* ch = (const BnChunk*)chunks;
    float* part = ws;
    float* segsum = ws + (size_t)8 * n_chunks * c;
    ProfScope ps(KK_BATCHNORM, s);
    const int l4 = bn_vec_log2(c, ld_dy, relu ? ld_y : c, ld_dx, dy, relu ? y : dy, xhat, gamma, dx);
    if (l4 >= 0) {
        INSMOS_LAUNCH(k_bnseg_bwd_sums_v4<false>, dim3(n_chunks, 4), dim3(256), 0, s, dy, ld_dy, y, ld_y, xhat, c, l4, relu, ch, part,
                      BnRecomp{nullptr, nullptr, nullptr});
        INSMOS_LAUNCH(k_bnseg_sums_merge_v4, dim3((int)cdiv((int64_t)(1 << l4), bn_merge_groups_per_block(S))), dim3(256), 0, s, part, c, l4,
                      seg_first, S, segsum, dgamma, dbeta);
        INSMOS_LAUNCH(k_bnseg_bwd_dx_v4<false>, dim3(n_chunks, 4), dim3(256), 0, s, dy, ld_dy, y, ld_y, xhat, c, l4, relu, ch, seg_rows, gamma,
                      stats, segsum, dx, ld_dx, (const float*)nullptr);
        HIP_TRY(hipGetLastError());
        return INSMOS_OK;
    }
    INSMOS_LAUNCH(k_bnseg_bwd_sums, dim3(n_chunks, cdiv(c, 16)), dim3(256), 0, s, dy, ld_dy, y, ld_y, xhat, c, relu, ch, part, ticket, seg_first,
                  S, segsum, dgamma, dbeta);
    INSMOS_LAUNCH(k_bnseg_bwd_dx, dim3(n_chunks, 4), dim3(256), 0, s, dy, ld_dy, y, ld_y, xhat, c, relu, ch, seg_rows, gamma, stats, segsum,
                  dx, ld_dx);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// The backward of a forward that did NOT store x^ (insmos_batchnorm_seg_forward with xhat = NULL): x = the layer's input (pitch ld_x),
// x^ and the ReLU mask are recomputed from it -- the same expressions, the same gradients bit for bit, three passes over the layer's
// elements fewer per step.  16-byte kernels only: EINVAL for shapes insmos_batchnorm_seg_recompute_ok refuses.
extern "C" int insmos_batchnorm_seg_recompute_ok(int c, int ld_x, int ld_dy, int ld_dx) {
    return (c >= 4 && !(c & 3) && !((c / 4) & (c / 4 - 1)) && c / 4 <= 64 && !(ld_x & 3) && !(ld_dy & 3) && !(ld_dx & 3)) ? 1 : 0;
}
extern "C" int insmos_batchnorm_seg_backward_x(const float* dy, int ld_dy, const float* x, int ld_x, int c, int64_t n, const int32_t* chunks,
                                               int n_chunks, const int32_t* seg_first, const int32_t* seg_rows, int S, const float* gamma,
                                               const float* beta, const float* stats, int relu, float* dx, int ld_dx, float* dgamma,
                                               float* dbeta, int32_t* ticket, float* ws, void* stream) {
    if (n <= 0 || c <= 0 || n_chunks <= 0) return INSMOS_OK;
    if (!dy || !x || !chunks || !seg_first || !seg_rows || S < 1 || !gamma || !beta || !stats || !dx || !dgamma || !dbeta || !ws || !ticket ||
        ld_dy < c || ld_dx < c || ld_x < c)
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const BnChunk* ch = (const BnChunk*)chunks;
    float* part = ws;
    float* segsum = ws + (size_t)8 * n_chunks * c;
    const int l4 = bn_vec_log2(c, ld_dy, ld_x, ld_dx, dy, x, beta, gamma, dx);
    if (l4 < 0) return INSMOS_EINVAL;
    ProfScope ps(KK_BATCHNORM, s);
    INSMOS_LAUNCH(k_bnseg_bwd_sums_v4<true>, dim3(n_chunks, 4), dim3(256), 0, s, dy, ld_dy, (const float*)nullptr, ld_x, x, c, l4, relu, ch, part,
                  BnRecomp{stats, gamma, beta});
    INSMOS_LAUNCH(k_bnseg_sums_merge_v4, dim3((int)cdiv((int64_t)(1 << l4), bn_merge_groups_per_block(S))), dim3(256), 0, s, part, c, l4,
                  seg_first, S, segsum, dgamma, dbeta);
    INSMOS_LAUNCH(k_bnseg_bwd_dx_v4<true>, dim3(n_chunks, 4), dim3(256), 0, s, dy, ld_dy, (const float*)nullptr, ld_x, x, c, l4, relu, ch, seg_rows,
                  gamma, stats, segsum, dx, ld_dx, beta);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// g_ws: (n, c) floats (the ReLU-masked upstream gradient); dgamma, dbeta (c) out; dx (n, ld_dx) out
extern "C" int insmos_batchnorm_train_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* xhat, int c,
                                               int64_t n, const float* gamma, const float* stats, int relu, float* dx, int ld_dx,
                                               float* dgamma, float* dbeta, float* g_ws, float* ws, void* stream) {
    if (n <= 0 || c <= 0) return INSMOS_OK;
    if (!dy || !xhat || !gamma || !stats || !dx || !dgamma || !dbeta || !g_ws || !ws || (relu && !y)) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)((n + 1023) / 1024);
    ProfScope ps(KK_BATCHNORM, s);
    const float* g = dy;
    int ld_g = ld_dy;
    if (relu || ld_dy != c) {
        if (relu) INSMOS_LAUNCH(k_bn_mask_relu, dim3(cdiv(n * c, 256)), dim3(256), 0, s, dy, ld_dy, y, ld_y, c, n, g_ws);
        else HIP_TRY(hipMemcpy2DAsync(g_ws, (size_t)c * 4, dy, (size_t)ld_dy * 4, (size_t)c * 4, (size_t)n, hipMemcpyDeviceToDevice, s));
        g = g_ws;
        ld_g = c;
    }
    INSMOS_LAUNCH(k_bn_partial, dim3(nb, c), dim3(256), 0, s, g, ld_g, (const float*)nullptr, 0, (const float*)nullptr, c, n, 0, ws);
    INSMOS_LAUNCH(k_conv_dw_reduce, dim3(cdiv(c, 256)), dim3(256), 0, s, ws, nb, (int64_t)c, dbeta, 0);
    INSMOS_LAUNCH(k_bn_partial, dim3(nb, c), dim3(256), 0, s, g, ld_g, xhat, c, (const float*)nullptr, c, n, 2, ws);
    INSMOS_LAUNCH(k_conv_dw_reduce, dim3(cdiv(c, 256)), dim3(256), 0, s, ws, nb, (int64_t)c, dgamma, 0);
    if (ld_g != c) return INSMOS_EINVAL;
    INSMOS_LAUNCH(k_bn_backward_dx, dim3(cdiv(n * c, 256)), dim3(256), 0, s, g, xhat, gamma, stats + c, dbeta, dgamma, c, n, dx, ld_dx);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
