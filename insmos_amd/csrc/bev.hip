// insmos_amd/csrc/bev.hip -- the dense BEV backbone convolution (base_bev_backbone.py:33-61: ZeroPad2d(1) + Conv2d(3x3) /
// Conv2d(3x3, padding 1), each followed by BatchNorm2d + ReLU) as an LDS-tiled implicit GEMM on the matrix cores.
//
// 40 % of the window's flops are these six dense layers.  Run through the sparse gather kernel (a 9-tap table over all
// sites) every tap re-gathers its 16 rows from global memory and every 16-row tile re-reads the whole 3x3xCinxCout weight
// set from L2 -- the layer is L2-bandwidth bound at ~55 % of the MFMA peak.  Here a workgroup owns a TH x 16 patch of one
// image and ALL output channels:
//   * the patch's input halo ((TH+2) x 18 sites) is staged in LDS one 16-channel chunk at a time (double buffered: the next
//     chunk's global loads are in flight during the current chunk's MFMAs), so every input element is read from global
//     memory once per workgroup instead of nine times per output-channel group;
//   * the waves split the patch 2 (row halves) x NCG (output-channel groups): a wave keeps TH/2 row groups x COT channel
//     tiles of accumulators, so each 1 KiB weight fragment it loads feeds TH/2 row groups (4-5x fewer weight bytes per
//     flop than a 16-row tile) and each B fragment it reads from LDS feeds COT channel tiles;
//   * B fragments are `ds_read_b128` at a 96-byte site pitch: with lane (g, j) reading the 16 bytes of channel group g
//     of site j, a 6-slot pitch puts the 16 lanes of every hardware lane group of ds_read_b128 on 16 distinct 16-byte bank
//     slots (brute-forced over the gfx950 lane groups, MI355X_MICROARCH.md section LDS): conflict-free.
// Arithmetic: v_mfma_f32_16x16x4_f32, exact fp32, fixed summation order (chunk, tap, 4 channel steps): deterministic,
// every output site a function of its own 3x3 neighbourhood only.  Weights are the same pre-packed A fragments
// k_sparse_conv uses ([tap][chunk][channel tile][lane][4]); fragment roles as there: i = output channel, j = site.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "prec.h"

namespace insmos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define BEV_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int BEV_TW = 16;           // sites per row group (x extent of a patch)
constexpr int BEV_HW = BEV_TW + 2;   // halo width
constexpr int BEV_PITCH = 24;        // floats per halo site in LDS: 16 channels + 8 pad = 96 bytes (conflict-free, see above)

// YM: the 16-site row groups run along y (and the TH extent along x) instead of along x -- picked per launch so that the
// patches pad the image as little as possible (150 x 125: 16-wide strips along x waste 6 % of the last strip, along y 2 %)
// NCG: output-channel groups (waves = 2 row halves x NCG; a wave owns COT = Cout / 16 / NCG channel tiles)
// PREC = 3: the split-bf16 x 3 experiment (prec.h; never the default): the halo is split into (hi4 | lo4) bf16 when it is
// staged into LDS -- once per element, same 16 bytes per lane and the same bank pattern --, the weights arrive pre-split
// SKIP (round 4): constant-region skipping.  A BEV map is mostly EMPTY (13.6 % of the sites of the S0 window hold a voxel), and an
// empty region stays a CONSTANT through the stack: layer 0 maps an all-zero 3x3 neighbourhood to relu(bias) =: c_0, and layer
// l >= 1 maps a neighbourhood that is c_{l-1} everywhere (and inside the image: zero padding breaks the constant) to one fixed
// vector c_l -- the same expression on the same operands, hence the same bits, at every such site.  c_l depends on the weights
// only (insmos_bev_constant computes it with THIS kernel on a constant image).  `dist` = Chebyshev distance of every site to the
// nearest occupied site (bytes, capped); a site is non-constant after layer l iff dist <= reach (= l + 1) or it lies within
// `breach` (= l - 1; -1 for layer 0, whose constant is the padding value itself) of the image border.  A 16-site row group
// without such a site skips its MFMAs and LDS reads and stores c_l; a workgroup without one skips its halo loads as well.
// Output bits are identical with and without SKIP (tests/test_gpu_conv.py, and the native runner against the step path).
template <int TH, int COT, int NCG, bool YM, int PREC = 0, bool SKIP = false>
__global__ void __launch_bounds__(128 * NCG, NCG == 4 ? 4 : 2) k_bev_conv3x3(const float* __restrict__ x, int H, int W, int n_img, int ld_x, int n16,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, int ld_out, int relu, int n_tx, int n_ty,
                                                     const uint8_t* __restrict__ dist, int reach, int breach,
                                                     const float* __restrict__ cvec, int ty_fast, int ntile) {
    constexpr int JT = TH / 2;                          // row groups per wave
    constexpr int NSITE = (TH + 2) * BEV_HW;            // halo sites
    constexpr int NTHR = 128 * NCG;
    constexpr int NQ = (NSITE * 4 + NTHR - 1) / NTHR;   // float4 pieces per thread per chunk
    __shared__ __attribute__((aligned(16))) float halo[2][NSITE * BEV_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wave & 1, ch = wave >> 1;            // row half, output-channel half
    const int g = lane >> 4, j = lane & 15;
    const int bid = blockIdx.x;
    // workgroup b runs on XCD b % 8.  With constant-region skipping the work of a patch depends on WHERE it lies (voxels cluster
    // around the sensor), so the patches of one XCD must be a spatial mix: the axis whose patch count shares the fewest factors of
    // two with the XCD count runs fastest (cfg-2: 8 x 15 patches -- with the 8-patch axis fastest every XCD owned one strip of the
    // map and the launch lasted as long as the busiest strip: measured, tools/bev_skip_probe.py)
    const int img = bid / (n_tx * n_ty);
    const int pq = bid % (n_tx * n_ty);
    const int tx = ty_fast ? pq / n_ty : pq % n_tx, ty = ty_fast ? pq % n_ty : pq / n_tx;
    // patch origin: (u, v) = (fast axis, slow axis) of the patch; u is x (v is y) unless YM
    const int u0 = tx * BEV_TW, v0 = ty * TH;
    const int U = YM ? H : W, V = YM ? W : H;
    // ntile = channel tiles of the LAYER (the weight blocks' pitch); this workgroup owns NCG * COT of them from tile0 on
    // (blockIdx.y: a 128-channel layer run as two 64-channel workgroups per patch when the patches alone cannot fill the chip)
    const int tile0 = (int)blockIdx.y * NCG * COT;

    // ---- SKIP: which of the patch's TH row groups hold a non-constant output site (bit r of `pact`; wave-uniform, and the same in
    // every wave of the workgroup: each wave looks at the whole patch)
    uint32_t pact = ~0u;
    if constexpr (SKIP) {
        pact = 0;
        const int gu_l = u0 + (lane & 15);
#pragma unroll
        for (int r4 = 0; r4 < TH; r4 += 4) {          // lane group g tests row group r4 + g
            const int r = r4 + (lane >> 4);
            const int gv_l = v0 + r;
            bool on = false;
            if (r < TH && gu_l < U && gv_l < V) {
                const int gy_l = YM ? gu_l : gv_l, gx_l = YM ? gv_l : gu_l;
                const int bd = min(min(gy_l, H - 1 - gy_l), min(gx_l, W - 1 - gx_l));
                on = (int)dist[((size_t)img * H + gy_l) * W + gx_l] <= reach || bd <= breach;
            }
            const unsigned long long bal = __ballot(on);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r4 + q < TH && ((bal >> (16 * q)) & 0xFFFFull)) pact |= 1u << (r4 + q);
        }
        pact = __builtin_amdgcn_readfirstlane(pact);
    }

    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)n_img * H * W * ld_x * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((size_t)9 * n16 * ntile * 1024), 0x00020000);

    // ---- the thread's pieces of a halo chunk: byte offset in x (out of range -> the buffer load returns 0: zero padding
    // at the image border and beyond the patch list) and float offset in the LDS image
    uint32_t src[NQ], dst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + NTHR * q;
        const int s = i >> 2, quarter = i & 3;
        const int hv = s / BEV_HW, hu = s % BEV_HW;
        const int gu = u0 - 1 + hu, gv = v0 - 1 + hv;
        const bool ok = i < NSITE * 4 && gu >= 0 && gu < U && gv >= 0 && gv < V;
        const int gy = YM ? gu : gv, gx = YM ? gv : gu;
        src[q] = ok ? (uint32_t)((((size_t)img * H + gy) * W + gx) * (size_t)ld_x * 4u + quarter * 16u) : 0xFFFFFFF0u;
        dst[q] = i < NSITE * 4 ? (uint32_t)(s * BEV_PITCH + quarter * 4) : 0xFFFFFFFFu;
    }
    f32x4 pf[NQ];
    auto fetch = [&](int c) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            pf[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, src[q], (uint32_t)c * 64u, 0));
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (dst[q] != 0xFFFFFFFFu) {
                if constexpr (PREC == 3) {
                    s16x4 hi, lo;
                    split_bf16(pf[q], hi, lo);
                    *(f32x4*)(&halo[buf][dst[q]]) = pack_split(hi, lo);
                } else {
                    *(f32x4*)(&halo[buf][dst[q]]) = pf[q];
                }
            }
    };

    f32x4 acc[COT][JT];
#pragma unroll
    for (int it = 0; it < COT; ++it)
#pragma unroll
        for (int r = 0; r < JT; ++r) acc[it][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS float offset of this lane's B fragment for tap (0, 0) of row group r: site (rh*JT + r, j), channels 4g..4g+3
    uint32_t boff[JT];
#pragma unroll
    for (int r = 0; r < JT; ++r) boff[r] = (uint32_t)(((rh * JT + r) * BEV_HW + j) * BEV_PITCH + 4 * g);
    // byte offset of this lane's A fragment inside a (tap, chunk) block of `ntile` fragments
    const uint32_t aoff = (uint32_t)((tile0 + ch * COT) * 1024 + lane * 16);
    const uint32_t blk_bytes = (uint32_t)ntile * 1024u;          // one (tap, chunk) block
    const uint32_t tap_bytes = (uint32_t)n16 * blk_bytes;

    const uint32_t ract = SKIP ? ((pact >> (rh * JT)) & ((1u << JT) - 1u)) : ~0u;   // this wave's row groups
    if (!SKIP || pact != 0) {
        fetch(0);
        stage(0);
        __syncthreads();
        f32x4 a[3][COT];  // weight ring: item (c, k) lives in slot k % 3 (9 taps = 3 turns: the slot of a tap is chunk-independent)
        auto load_a = [&](int slot, int k, int c) {
            const uint32_t so = (uint32_t)k * tap_bytes + (uint32_t)c * blk_bytes;
    #pragma unroll
            for (int it = 0; it < COT; ++it)
                a[slot][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, aoff + (uint32_t)it * 1024u, so, 0));
        };
        load_a(0, 0, 0);
        for (int c = 0; c < n16; ++c) {
            const int buf = c & 1;
            const float* hb = halo[buf];
    #pragma unroll
            for (int k = 0; k < 9; ++k) {
                // next item's weights: tap k+1 of this chunk, or tap 0 of the next chunk (clamped on the very last item)
                if (k < 8) load_a((k + 1) % 3, k + 1, c);
                else load_a(0, 0, c + 1 < n16 ? c + 1 : c);
                // the next chunk's halo: requested behind the second tap's weights (so that the first tap's MFMAs wait for
                // their own operands only), in flight during the rest of this chunk's MFMAs
                if (k == 1 && c + 1 < n16) fetch(c + 1);
                const int ky = k / 3, kx = k % 3;
                const int ku = YM ? ky : kx, kv = YM ? kx : ky;   // tap offset along the patch's fast / slow axis
                if constexpr (SKIP) {
                    static_assert(!SKIP || PREC == 0, "constant-region skipping is built for the exact fp32 path");
                    // row-group major: a skipped group costs one scalar branch; inside a group the COT accumulators alternate, so
                    // two MFMAs on the same accumulator are 64 cycles apart (dependent latency 40).  Same chain per accumulator.
    #pragma unroll
                    for (int r = 0; r < JT; ++r) {
                        if (!((ract >> r) & 1u)) continue;
                        const f32x4 br = *(const f32x4*)(hb + boff[r] + (kv * BEV_HW + ku) * BEV_PITCH);
    #pragma unroll
                        for (int s = 0; s < 4; ++s)
    #pragma unroll
                            for (int it = 0; it < COT; ++it) acc[it][r] = BEV_MFMA(a[k % 3][it][s], br[s], acc[it][r]);
                    }
                    continue;
                }
                f32x4 b[JT];
    #pragma unroll
                for (int r = 0; r < JT; ++r) b[r] = *(const f32x4*)(hb + boff[r] + (kv * BEV_HW + ku) * BEV_PITCH);
                if constexpr (PREC == 3) {
                    s16x4 ah[COT], al[COT];
    #pragma unroll
                    for (int it = 0; it < COT; ++it) unpack_split(a[k % 3][it], ah[it], al[it]);
    #pragma unroll
                    for (int r = 0; r < JT; ++r) {
                        s16x4 bh, bl;
                        unpack_split(b[r], bh, bl);
    #pragma unroll
                        for (int it = 0; it < COT; ++it) {
                            acc[it][r] = MFMA_BF16(al[it], bh, acc[it][r]);  // small terms first
                            acc[it][r] = MFMA_BF16(ah[it], bl, acc[it][r]);
                            acc[it][r] = MFMA_BF16(ah[it], bh, acc[it][r]);
                        }
                    }
                } else {
    #pragma unroll
                    for (int s = 0; s < 4; ++s)
    #pragma unroll
                        for (int r = 0; r < JT; ++r)
    #pragma unroll
                            for (int it = 0; it < COT; ++it) acc[it][r] = BEV_MFMA(a[k % 3][it][s], b[r][s], acc[it][r]);
                }
            }
            if (c + 1 < n16) stage(buf ^ 1);  // (that buffer was last read in chunk c-1; every wave is past that barrier)
            __syncthreads();
        }
    }

    // ---- epilogue: lane (g, j) holds channels co0..co0+3 of site (u0 + j, v0 + rh*JT + r)
    const int gu = u0 + j;
#pragma unroll
    for (int r = 0; r < JT; ++r) {
        const int gv = v0 + rh * JT + r;
        if (gv >= V || gu >= U) continue;
        const int gy = YM ? gu : gv, gx = YM ? gv : gu;
        float* op = out + (((size_t)img * H + gy) * W + gx) * (size_t)ld_out;
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            const int co0 = (tile0 + ch * COT + it) * 16 + 4 * g;
            if (SKIP && !((ract >> r) & 1u)) {   // a constant row group: the layer's constant vector (bias and ReLU included)
                *(f32x4*)(op + co0) = *(const f32x4*)(cvec + co0);
                continue;
            }
            f32x4 v = acc[it][r] + *(const f32x4*)(bias + co0);
            if (relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            *(f32x4*)(op + co0) = v;
        }
    }
}

// ---- constant-region skipping over a COMPACTED list of row groups (round 4, second half).  The patch kernel above skips inside a
// fixed 16 x 4 patch: a patch with one active row group of four still stages its whole halo, walks every chunk's barriers and keeps
// six of its eight waves idle -- the stack executed 58-72 % of the dense (site, tap) pairs in 83-87 % of the dense time
// (profiles/r04_bev_skip_layers.txt).  Here the ACTIVE 16-site row groups of an image are listed first (k_bev_group_lists:
// ascending, deterministic) and a workgroup takes TH consecutive list entries, wherever they lie: every wave has work, every
// workgroup the same amount.  A group brings its own 3 x 18-site halo (groups of one workgroup need not be neighbours: 54 halo
// sites per group against 21.6 in a dense 16 x 10 patch -- LDS and L2 reads, not HBM: the input was just written by the previous
// layer).  The inactive groups are the list's tail, written back to front; the workgroups behind the computing ones store the
// layer's constant into them.  Per output element the MFMA chain is that of k_bev_conv3x3: same bits (tests/test_gpu_conv.py).
template <int TH, int COT, int NCG, bool YM>
__global__ void __launch_bounds__(128 * NCG, NCG == 4 ? (TH == 4 ? 6 : 4) : (TH == 4 ? 3 : 2)) k_bev_conv3x3_list(
    const float* __restrict__ x, int H, int W, int n_img, int ld_x, int n16, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int ld_out, int relu, const int32_t* __restrict__ lists, const int32_t* __restrict__ n_act, int G,
    int n_tu, const float* __restrict__ cvec, int ntile) {
    constexpr int JT = TH / 2;                          // row groups per wave
    constexpr int GS = 3 * BEV_HW;                      // halo sites of one group
    constexpr int NSITE = TH * GS;
    constexpr int NTHR = 128 * NCG;
    constexpr int NQ = (NSITE * 4 + NTHR - 1) / NTHR;   // float4 pieces per thread per chunk
    __shared__ __attribute__((aligned(16))) float halo[2][NSITE * BEV_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wave & 1, ch = wave >> 1;
    const int g = lane >> 4, j = lane & 15;
    // chunk-major over the images: the computing workgroups (low chunk numbers) of ALL images are dispatched before the constant
    // fills (the lists' tails), so the launch ends on light workgroups; the image a workgroup takes rotates with the chunk number,
    // so that an XCD (workgroup b -> XCD b % 8) sees every image of a set of 8 instead of one
    const int c = (int)blockIdx.x / n_img;
    const int img = ((int)blockIdx.x + c) % n_img;
    const int U = YM ? H : W, V = YM ? W : H;
    const int tile0 = (int)blockIdx.y * NCG * COT;
    const int32_t* __restrict__ L = lists + (size_t)img * G;
    const int na = n_act[img];
    const int nca = (na + TH - 1) / TH;
    if (c >= nca) {
        // ---- constant fill: TH inactive groups (list tail, back to front) per workgroup, this workgroup's channel tiles
        constexpr int NV = NCG * COT * 4;               // float4 per site
        const int first = (c - nca) * TH, n_in = G - na;
        for (int r = 0; r < TH; ++r) {
            const int k = first + r;
            if (k >= n_in) break;
            const int gid = L[G - 1 - k];
            const int gv = gid / n_tu, u0 = (gid % n_tu) * BEV_TW;
            for (int i = tid; i < BEV_TW * NV; i += NTHR) {
                const int site = i / NV, q = i % NV;
                const int gu = u0 + site;
                if (gu >= U) continue;
                const int gy = YM ? gu : gv, gx = YM ? gv : gu;
                const int co0 = tile0 * 16 + q * 4;
                *(f32x4*)(out + (((size_t)img * H + gy) * W + gx) * (size_t)ld_out + co0) = *(const f32x4*)(cvec + co0);
            }
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)n_img * H * W * ld_x * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((size_t)9 * n16 * ntile * 1024), 0x00020000);
    const int k0 = c * TH;                               // first list entry of this workgroup
    uint32_t src[NQ], dst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + NTHR * q;
        const int sidx = i >> 2, quarter = i & 3;
        const int r = sidx / GS, rem = sidx % GS;
        const int hv = rem / BEV_HW, hu = rem % BEV_HW;
        bool ok = i < NSITE * 4 && k0 + r < na;
        int gu = 0, gv = 0;
        if (ok) {
            const int gid = L[k0 + r];
            gv = gid / n_tu - 1 + hv;
            gu = (gid % n_tu) * BEV_TW - 1 + hu;
            ok = gu >= 0 && gu < U && gv >= 0 && gv < V;
        }
        const int gy = YM ? gu : gv, gx = YM ? gv : gu;
        src[q] = ok ? (uint32_t)((((size_t)img * H + gy) * W + gx) * (size_t)ld_x * 4u + quarter * 16u) : 0xFFFFFFF0u;
        dst[q] = i < NSITE * 4 ? (uint32_t)(sidx * BEV_PITCH + quarter * 4) : 0xFFFFFFFFu;
    }
    f32x4 pf[NQ];
    auto fetch = [&](int cc) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            pf[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, src[q], (uint32_t)cc * 64u, 0));
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (dst[q] != 0xFFFFFFFFu) *(f32x4*)(&halo[buf][dst[q]]) = pf[q];
    };
    f32x4 acc[COT][JT];
#pragma unroll
    for (int it = 0; it < COT; ++it)
#pragma unroll
        for (int r = 0; r < JT; ++r) acc[it][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // LDS float offset of this lane's B fragment for tap (0, 0) of the wave's group r: site (row 0, j) of group rh*JT + r
    uint32_t boff[JT];
#pragma unroll
    for (int r = 0; r < JT; ++r) boff[r] = (uint32_t)((((rh * JT + r) * 3) * BEV_HW + j) * BEV_PITCH + 4 * g);
    const uint32_t aoff = (uint32_t)((tile0 + ch * COT) * 1024 + lane * 16);
    const uint32_t blk_bytes = (uint32_t)ntile * 1024u;
    const uint32_t tap_bytes = (uint32_t)n16 * blk_bytes;

    fetch(0);
    stage(0);
    __syncthreads();
    f32x4 a[3][COT];
    auto load_a = [&](int slot, int k, int cc) {
        const uint32_t so = (uint32_t)k * tap_bytes + (uint32_t)cc * blk_bytes;
#pragma unroll
        for (int it = 0; it < COT; ++it)
            a[slot][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, aoff + (uint32_t)it * 1024u, so, 0));
    };
    load_a(0, 0, 0);
    for (int cc = 0; cc < n16; ++cc) {
        const int buf = cc & 1;
        const float* hb = halo[buf];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k < 8) load_a((k + 1) % 3, k + 1, cc);
            else load_a(0, 0, cc + 1 < n16 ? cc + 1 : cc);
            if (k == 1 && cc + 1 < n16) fetch(cc + 1);
            const int ky = k / 3, kx = k % 3;
            const int ku = YM ? ky : kx, kv = YM ? kx : ky;
            f32x4 b[JT];
#pragma unroll
            for (int r = 0; r < JT; ++r) b[r] = *(const f32x4*)(hb + boff[r] + (kv * BEV_HW + ku) * BEV_PITCH);
            // (the order of k_bev_conv3x3's dense loop; per accumulator the chain (chunk, tap, s) is the same in every variant)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int r = 0; r < JT; ++r)
#pragma unroll
                    for (int it = 0; it < COT; ++it) acc[it][r] = BEV_MFMA(a[k % 3][it][s], b[r][s], acc[it][r]);
        }
        if (cc + 1 < n16) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < JT; ++r) {
        const int kk = k0 + rh * JT + r;
        if (kk >= na) continue;                          // (wave-uniform: the list's last workgroup may be short)
        const int gid = L[kk];
        const int gv = gid / n_tu, gu = (gid % n_tu) * BEV_TW + j;
        if (gu >= U) continue;
        const int gy = YM ? gu : gv, gx = YM ? gv : gu;
        float* op = out + (((size_t)img * H + gy) * W + gx) * (size_t)ld_out;
#pragma unroll
        for (int it = 0; it < COT; ++it) {
            const int co0 = (tile0 + ch * COT + it) * 16 + 4 * g;
            f32x4 v = acc[it][r] + *(const f32x4*)(bias + co0);
            if (relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            *(f32x4*)(op + co0) = v;
        }
    }
}

// The row-group lists of one layer: block = one image; lists[img][0 .. n_act) = ids (v * n_tu + u_tile) of the groups that hold a
// non-constant site (the predicate of k_bev_conv3x3<SKIP>), ascending; lists[img][G - 1 - k] = the k-th inactive group.
__global__ void __launch_bounds__(1024) k_bev_group_lists(const uint8_t* __restrict__ dist, int H, int W, int ym, int reach, int breach,
                                                          int n_tu, int G, int32_t* __restrict__ lists, int32_t* __restrict__ n_act) {
    __shared__ int wsum[16];
    __shared__ int s_tot;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int U = ym ? H : W;
    int32_t* __restrict__ L = lists + (size_t)img * G;
    int off_act = 0, off_in = 0;
    for (int base = 0; base < G; base += 1024) {
        const int gid = base + tid;
        bool on = false;
        if (gid < G) {
            const int v = gid / n_tu, u0 = (gid % n_tu) * BEV_TW;
            for (int jj = 0; jj < BEV_TW; ++jj) {
                const int u = u0 + jj;
                if (u >= U) break;
                const int gy = ym ? u : v, gx = ym ? v : u;
                const int bd = min(min(gy, H - 1 - gy), min(gx, W - 1 - gx));
                on = on || (int)dist[((size_t)img * H + gy) * W + gx] <= reach || bd <= breach;
            }
        }
        const unsigned long long bal = __ballot(on);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int q = 0; q < 16; ++q) { const int t = wsum[q]; wsum[q] = run; run += t; }
            s_tot = run;
        }
        __syncthreads();
        if (gid < G) {
            const int pos = wsum[wv] + before;            // active groups of this pass before this one
            if (on) L[off_act + pos] = gid;
            else L[G - 1 - (off_in + (tid - pos))] = gid;
        }
        const int tot = s_tot, cnt = min(1024, G - base);
        off_act += tot;
        off_in += cnt - tot;
        __syncthreads();
    }
    if (tid == 0) n_act[img] = off_act;
}

// ---- Chebyshev distance of every BEV site to the nearest occupied site (bytes, capped at cap + 1), two separable passes
__global__ void k_bev_occupancy(const int32_t* __restrict__ coords, int64_t n, int H, int W, uint8_t* __restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 q = *(const int4*)(coords + i * 4);   // [b, d, y, x]
    occ[((int64_t)q.x * H + q.z) * W + q.w] = 1;
}
// rowd[site] = min |dx| over occupied sites of the same row within `cap` (cap + 1 if none)
__global__ void k_bev_dist_rows(const uint8_t* __restrict__ occ, int64_t n_site, int W, int cap, uint8_t* __restrict__ rowd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_site) return;
    const int x = (int)(i % W);
    int best = cap + 1;
    for (int d = 0; d <= cap && d < best; ++d) {
        if ((x - d >= 0 && occ[i - d]) || (x + d < W && occ[i + d])) best = d;
    }
    rowd[i] = (uint8_t)best;
}
// dist[site] = min over dy of max(|dy|, rowd[y + dy][x])
__global__ void k_bev_dist_cols(const uint8_t* __restrict__ rowd, int64_t n_site, int H, int W, int cap, uint8_t* __restrict__ dist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_site) return;
    const int y = (int)((i / W) % H);
    int best = cap + 1;
    for (int d = 0; d <= cap && d < best; ++d) {
        if (y - d >= 0) best = min(best, max(d, (int)rowd[i - (int64_t)d * W]));
        if (y + d < H) best = min(best, max(d, (int)rowd[i + (int64_t)d * W]));
    }
    dist[i] = (uint8_t)best;
}

// valid (in-image) taps of the in-image sites of the row groups the SKIP kernel computes for `layer` (accounting only)
__global__ void k_bev_skip_pairs(const uint8_t* __restrict__ dist, int n_img, int H, int W, int ym, int reach, int breach,
                                 unsigned long long* __restrict__ out) {
    const int U = ym ? H : W, V = ym ? W : H;
    const int n_tu = (U + 15) / 16;
    const int64_t grp = (int64_t)blockIdx.x * (blockDim.x / 16) + threadIdx.x / 16;   // 16 lanes = one row group
    const int j = threadIdx.x & 15;
    const int64_t n_grp = (int64_t)n_img * V * n_tu;
    bool on = false;
    int taps = 0;
    if (grp < n_grp) {
        const int img = (int)(grp / ((int64_t)V * n_tu));
        const int v = (int)((grp / n_tu) % V), u = (int)(grp % n_tu) * 16 + j;
        if (u < U) {
            const int gy = ym ? u : v, gx = ym ? v : u;
            const int bd = min(min(gy, H - 1 - gy), min(gx, W - 1 - gx));
            on = (int)dist[((size_t)img * H + gy) * W + gx] <= reach || bd <= breach;
            taps = ((gy > 0) + 1 + (gy < H - 1)) * ((gx > 0) + 1 + (gx < W - 1));
        }
    }
    // any lane of the 16-lane group on -> the whole group is computed
    const unsigned long long bal = __ballot(on);
    const int sh = (threadIdx.x & 63) & ~15;
    const bool grp_on = ((bal >> sh) & 0xFFFFull) != 0;
    if (grp_on && taps) atomicAdd(out, (unsigned long long)taps);
}

}  // namespace insmos

using namespace insmos;

// x (B, H, W, cin) NHWC with row pitch ld_x floats -> out (B, H, W, cout) with row pitch ld_out: 3x3, stride 1, zero
// padding 1, + bias (folded BatchNorm) + optional ReLU.  wpacked / bias as insmos_pack_weights_host(taps (9, cin, cout))
// with tap = ky * 3 + kx.  Supported: cin a multiple of 16, cout = 128 or 64.
// patch shape: 16 sites along one axis x TH in {10, 8, 4} along the other.  Every site's value is the same expression
// whatever patch it falls into, so the choice is free: least padded work, at least one workgroup per CU, tallest wins ties
// (a taller patch re-uses each weight fragment for more row groups).
// only_th > 0: that patch height only (constant-region skipping takes the flattest patches: more, smaller workgroups balance the
// uneven work per patch better and more of them are skipped whole -- measured 16 x 4 against 16 x 8 / 16 x 10, tools/bev_skip_probe.py)
static void bev_choose_patch(int B, int H, int W, int* th_out, int* ym_out, int only_th = 0) {
    constexpr int kCUs = 256;  // MI355X
    int best_th = 0, best_ym = 0;
    double best_cost = 1e30;
    for (int ym = 0; ym < 2; ++ym)
        for (int th : {10, 8, 4}) {
            if (only_th && th != only_th) continue;
            const int U = ym ? H : W, V = ym ? W : H;
            const int64_t nblk = (int64_t)B * ((U + 15) / 16) * ((V + th - 1) / th);
            double cost = (double)nblk * 16.0 * th;                        // padded sites
            if (nblk < kCUs) cost *= (double)kCUs / (double)nblk;          // idle CUs
            cost *= 1.0 + 0.04 * (10 - th) / 6.0;                           // weight traffic of the flatter patches
            const int64_t slots = (int64_t)kCUs * (th == 4 ? 4 : 2);       // resident workgroups per round
            const double last_round = (double)nblk / (double)(((nblk + slots - 1) / slots) * slots);
            cost /= 0.5 + 0.5 * last_round;                                 // a part-empty last round of workgroups
            if (cost < best_cost) { best_cost = cost; best_th = th; best_ym = ym; }
        }
    // probe only (tools/bev_skip_probe.py): INSMOS_BEV_TH / INSMOS_BEV_YM force the patch height / orientation
    if (const char* e = getenv("INSMOS_BEV_TH")) { const int v = atoi(e); if (v == 10 || v == 8 || v == 4) best_th = v; }
    if (const char* e = getenv("INSMOS_BEV_YM")) { const int v = atoi(e); if (v == 0 || v == 1) best_ym = v; }
    *th_out = best_th;
    *ym_out = best_ym;
}

// workgroup count below which a 128-channel layer is run as two 64-channel workgroups per patch (insmos_bev_cosplit; 0 = never)
static int g_bev_cosplit = -1;
static int bev_cosplit_max_wgs() {
    if (g_bev_cosplit >= 0) return g_bev_cosplit;
    static const int env = [] { const char* e = getenv("INSMOS_BEV_COSPLIT"); return e ? atoi(e) : 1024; }();
    return env < 0 ? 0 : env;
}
extern "C" int insmos_bev_cosplit(int max_wgs) {
    if (max_wgs < -1) return INSMOS_EINVAL;
    g_bev_cosplit = max_wgs;
    return INSMOS_OK;
}

static int bev_conv3x3_impl(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                            float* out, int ld_out, int cout, int relu, const uint8_t* dist, int reach, int breach,
                            const float* cvec, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return INSMOS_OK;
    if (!x || !wpacked || !bias || !out || cin <= 0 || cin % 16 != 0 || ld_x < cin || (ld_x & 3) || (cout != 128 && cout != 64) ||
        ld_out < cout || (ld_out & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) ||
        (int64_t)B * H * W * ld_x * 4 >= (1ll << 31))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int n16 = cin / 16;
    int best_th = 0, best_ym = 0;
    bev_choose_patch(B, H, W, &best_th, &best_ym, (dist && cvec) ? 4 : 0);
    const int U = best_ym ? H : W, V = best_ym ? W : H;
    const int n_tx = (U + BEV_TW - 1) / BEV_TW, n_ty = (V + best_th - 1) / best_th;
    const unsigned grid = (unsigned)((int64_t)B * n_ty * n_tx);
    auto pow2 = [](int v) { int p = 0; while (v > 0 && (v & 1) == 0 && p < 3) { v >>= 1; ++p; } return p; };
    const int ty_fast = pow2(n_ty) < pow2(n_tx) ? 1 : 0;   // (see the kernel: spatial mix of patches per XCD)
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = 9; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = (int64_t)B * H * W;
    // Cout = 128: 2 x 4 waves of 2 channel tiles (10-row patches: the accumulators of 4 tiles x 5 row groups would leave one
    // wave per SIMD); Cout = 64: 2 x 2 waves of 2 tiles
    // (experiment, never the default: split-bf16 x 3 when the mode is set and this layer's split weights are registered)
    const float* wsplit = conv_precision() == 3 ? (const float*)split_weights_of(wpacked) : nullptr;
    const bool skip = dist != nullptr && cvec != nullptr && !wsplit;
    // Channel split (round 4): a 128-channel layer whose patches alone cannot fill the chip -- ONE window: 304 workgroups of 8 waves
    // for 256 CUs, 48 CUs with two of them and the rest with one -- runs as two 64-channel workgroups of 4 waves per patch
    // (gridDim.y = 2, the Cout = 64 kernel on channel tiles tile0 ..): twice the workgroups at half the matrix work each, the halo
    // read twice (55 of 350 KB per workgroup).  Every output channel is the same chain of MFMAs either way: same bits
    // (tests/test_gpu_conv.py).  Launch sets of four windows or more have workgroups enough and keep the 8-wave kernel.
    const int ntile_all = cout / 16;
    const int cosplit = (cout == 128 && (int64_t)grid < (int64_t)bev_cosplit_max_wgs()) ? 2 : 1;
#define BEV_GO(TH_, NCG_, YM_)                                                                                                  \
    do {                                                                                                                        \
        if (wsplit)                                                                                                             \
            INSMOS_LAUNCH((k_bev_conv3x3<TH_, 2, NCG_, YM_, 3>), dim3(grid, cosplit), dim3(128 * NCG_), 0, s, x, H, W, B, ld_x, n16, wsplit, \
                          bias, out, ld_out, relu, n_tx, n_ty, nullptr, 0, 0, nullptr, ty_fast, ntile_all);                               \
        else if (skip)                                                                                                          \
            INSMOS_LAUNCH((k_bev_conv3x3<TH_, 2, NCG_, YM_, 0, true>), dim3(grid, cosplit), dim3(128 * NCG_), 0, s, x, H, W, B, ld_x, n16, \
                          wpacked, bias, out, ld_out, relu, n_tx, n_ty, dist, reach, breach, cvec, ty_fast, ntile_all);                  \
        else                                                                                                                    \
            INSMOS_LAUNCH((k_bev_conv3x3<TH_, 2, NCG_, YM_, 0>), dim3(grid, cosplit), dim3(128 * NCG_), 0, s, x, H, W, B, ld_x, n16, wpacked, \
                          bias, out, ld_out, relu, n_tx, n_ty, nullptr, 0, 0, nullptr, ty_fast, ntile_all);                               \
    } while (0)
#define BEV_TH(NCG_, YM_) \
    do { if (best_th == 10) BEV_GO(10, NCG_, YM_); else if (best_th == 8) BEV_GO(8, NCG_, YM_); else BEV_GO(4, NCG_, YM_); } while (0)
    if (cout == 128 && cosplit == 1) { if (best_ym) BEV_TH(4, true); else BEV_TH(4, false); }
    else                             { if (best_ym) BEV_TH(2, true); else BEV_TH(2, false); }
#undef BEV_TH
#undef BEV_GO
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_bev_conv3x3(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                                  float* out, int ld_out, int cout, int relu, void* stream) {
    return bev_conv3x3_impl(x, B, H, W, ld_x, cin, wpacked, bias, out, ld_out, cout, relu, nullptr, 0, 0, nullptr, stream);
}

// The same layer with constant-region skipping (see k_bev_conv3x3, SKIP): dist = insmos_bev_distance_map of the map's occupied
// sites, `layer` = position of this layer in the 3x3 stack (0 = the layer that reads the scattered BEV map), cvec = the layer's
// constant vector (insmos_bev_constant; relu must be what it was computed with).  Output bits == insmos_bev_conv3x3's.
extern "C" int insmos_bev_conv3x3_skip(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                                       float* out, int ld_out, int cout, int relu, const uint8_t* dist, int layer, const float* cvec,
                                       void* stream) {
    if (!dist || !cvec || ((uintptr_t)cvec & 15) || layer < 0 || layer > 200) return INSMOS_EINVAL;   // cvec is loaded as float4
    return bev_conv3x3_impl(x, B, H, W, ld_x, cin, wpacked, bias, out, ld_out, cout, relu, dist, layer + 1, layer - 1, cvec, stream);
}

// scratch of insmos_bev_conv3x3_skip_ws: one layer's row-group lists and per-image counts
extern "C" size_t insmos_bev_skip_ws_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t gmax = (int64_t)std::max(H, W) * ((std::max(H, W) + 15) / 16);
    return pad256((size_t)B * (size_t)gmax * 4) + pad256((size_t)B * 4) + 256;
}

// insmos_bev_conv3x3_skip over COMPACTED row groups (k_bev_conv3x3_list): the layer's active 16-site row groups are listed per image
// (ws) and the workgroups walk the list, TH = 4 groups each; the inactive groups get the constant.  Output bits == insmos_bev_conv3x3's.
extern "C" int insmos_bev_conv3x3_skip_ws(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                                          float* out, int ld_out, int cout, int relu, const uint8_t* dist, int layer, const float* cvec,
                                          void* ws, size_t ws_bytes, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return INSMOS_OK;
    if (!x || !wpacked || !bias || !out || !dist || !cvec || ((uintptr_t)cvec & 15) || !ws || layer < 0 || layer > 200 || cin <= 0 || cin % 16 != 0 || ld_x < cin ||
        (ld_x & 3) || (cout != 128 && cout != 64) || ld_out < cout || (ld_out & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) ||
        (int64_t)B * H * W * ld_x * 4 >= (1ll << 31))
        return INSMOS_EINVAL;
    if (ws_bytes < insmos_bev_skip_ws_bytes(B, H, W)) return INSMOS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int th = 0, ym = 0;
    bev_choose_patch(B, H, W, &th, &ym, 4);   // (the orientation of the patch kernel's skipping launches: the accounting is shared)
    const int U = ym ? H : W, V = ym ? W : H;
    const int n_tu = (U + BEV_TW - 1) / BEV_TW, G = V * n_tu;
    // groups per workgroup: 4 (41 KB of halo, three workgroups per CU) or 6 (62 KB, two per CU, each weight fragment feeds three
    // row groups per wave instead of two); INSMOS_BEV_LIST_TH, measured default below
    static const int TH = [] { const char* e = getenv("INSMOS_BEV_LIST_TH"); const int v = e ? atoi(e) : 4; return v == 6 ? 6 : 4; }();
    const int n_chunk = (G + TH - 1) / TH + 1;   // ceil(active / TH) + ceil(inactive / TH) <= ceil(G / TH) + 1
    int32_t* lists = (int32_t*)ws;
    int32_t* n_act = (int32_t*)((char*)ws + pad256((size_t)B * (size_t)G * 4));
    const int n16 = cin / 16, ntile_all = cout / 16;
    const unsigned grid = (unsigned)((int64_t)B * n_chunk);
    const int cosplit = (cout == 128 && (int64_t)grid < (int64_t)bev_cosplit_max_wgs()) ? 2 : 1;
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = 9; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = (int64_t)B * H * W;
    INSMOS_LAUNCH(k_bev_group_lists, dim3(B), dim3(1024), 0, s, dist, H, W, ym, layer + 1, layer - 1, n_tu, G, lists, n_act);
#define BEV_LIST_GO_(TH_, NCG_, YM_)                                                                                                   \
    INSMOS_LAUNCH((k_bev_conv3x3_list<TH_, 2, NCG_, YM_>), dim3(grid, cosplit), dim3(128 * NCG_), 0, s, x, H, W, B, ld_x, n16, wpacked, bias, \
                  out, ld_out, relu, lists, n_act, G, n_tu, cvec, ntile_all)
#define BEV_LIST_GO(NCG_, YM_) do { if (TH == 6) BEV_LIST_GO_(6, NCG_, YM_); else BEV_LIST_GO_(4, NCG_, YM_); } while (0)
    if (cout == 128 && cosplit == 1) { if (ym) BEV_LIST_GO(4, true); else BEV_LIST_GO(4, false); }
    else                             { if (ym) BEV_LIST_GO(2, true); else BEV_LIST_GO(2, false); }
#undef BEV_LIST_GO
#undef BEV_LIST_GO_
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" size_t insmos_bev_distance_map_ws_bytes(int B, int H, int W) { return 2 * pad256((size_t)B * H * W); }

// coords (n, 4) int32 [b, z, y, x] of the voxels scattered into the BEV map (spconv indices of the last encoder level) ->
// dist (B * H * W bytes): Chebyshev distance to the nearest occupied site, capped at cap + 1 (cap <= 254)
extern "C" int insmos_bev_distance_map(const int32_t* coords, int64_t n, int B, int H, int W, int cap, uint8_t* dist, void* ws,
                                       size_t ws_bytes, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || cap < 0 || cap > 254 || !dist || !ws || (n > 0 && !coords) ||
        ws_bytes < insmos_bev_distance_map_ws_bytes(B, H, W))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_site = (int64_t)B * H * W;
    uint8_t* occ = (uint8_t*)ws;
    uint8_t* rowd = occ + pad256((size_t)n_site);
    ProfScope ps(KK_TO_BEV, s);
    HIP_TRY(hipMemsetAsync(occ, 0, (size_t)n_site, s));
    if (n > 0) INSMOS_LAUNCH(k_bev_occupancy, dim3(cdiv(n, 256)), dim3(256), 0, s, coords, n, H, W, occ);
    INSMOS_LAUNCH(k_bev_dist_rows, dim3(cdiv(n_site, 256)), dim3(256), 0, s, occ, n_site, W, cap, rowd);
    INSMOS_LAUNCH(k_bev_dist_cols, dim3(cdiv(n_site, 256)), dim3(256), 0, s, rowd, n_site, H, W, cap, dist);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" size_t insmos_bev_constant_ws_floats(int cin, int cout) { return (size_t)25 * (size_t)(cin + cout) + 64; }

// The constant a 3x3 layer maps a constant neighbourhood to: c_out = epilogue(conv(x == c_in everywhere)) evaluated by the product
// kernel itself at the centre of a 5 x 5 image filled with c_in (cin floats; null = zeros), so that it carries exactly the bits
// the kernel produces at such a site.  Depends on the weights only: computed once per checkpoint and layer.
extern "C" int insmos_bev_constant(const float* wpacked, const float* bias, int cin, int cout, int relu, const float* c_in,
                                   float* c_out, float* ws, void* stream) {
    if (!wpacked || !bias || !c_out || !ws || cin <= 0 || cin % 16 != 0 || (cout != 64 && cout != 128)) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* img = ws;                       // (25, cin)
    float* res = ws + (((size_t)25 * cin + 63) & ~(size_t)63);   // (25, cout), 256-byte aligned
    if (c_in) {
        for (int i = 0; i < 25; ++i)
            HIP_TRY(hipMemcpyAsync(img + (size_t)i * cin, c_in, (size_t)cin * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        HIP_TRY(hipMemsetAsync(img, 0, (size_t)25 * cin * sizeof(float), s));
    }
    int rc = bev_conv3x3_impl(img, 1, 5, 5, cin, cin, wpacked, bias, res, cout, cout, relu, nullptr, 0, 0, nullptr, stream);
    if (rc != INSMOS_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c_out, res + (size_t)12 * cout, (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, s));
    return INSMOS_OK;
}

// Accounting (bench.py): the (site, tap) pairs insmos_bev_conv3x3_skip executes for `layer` on a (B, H, W) map with this distance
// map -- in-image taps of the in-image sites of the 16-site row groups (in the orientation the launcher picks) that hold a
// non-constant site.  *pairs_dev (device, 8 bytes) is overwritten.
extern "C" int insmos_bev_skip_executed_pairs(const uint8_t* dist, int B, int H, int W, int layer, unsigned long long* pairs_dev,
                                              void* stream) {
    if (!dist || !pairs_dev || B <= 0 || H <= 0 || W <= 0 || layer < 0) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int th = 0, ym = 0;
    bev_choose_patch(B, H, W, &th, &ym, 4);
    const int U = ym ? H : W, V = ym ? W : H;
    const int64_t n_grp = (int64_t)B * V * ((U + 15) / 16);
    HIP_TRY(hipMemsetAsync(pairs_dev, 0, sizeof(unsigned long long), s));
    INSMOS_LAUNCH(k_bev_skip_pairs, dim3(cdiv(n_grp, 16)), dim3(256), 0, s, dist, B, H, W, ym, layer + 1, layer - 1, pairs_dev);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
