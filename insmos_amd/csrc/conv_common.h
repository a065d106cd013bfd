// insmos_amd/csrc/conv_common.h -- what the convolution kernels of libinsmos_hip.so share: the launch parameter block and the
// fragment types (spconv.hip: generic / split / quad-index kernels; spconv_lds.hip: the LDS-staged 81-tap kernel).
#pragma once
#include "common.h"

namespace insmos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvP {
    const float* in;
    const int32_t* nbr;
    const uint32_t* mask16;  // [ceil(n_out/16)][4] active-tap bits per 16-row group, or null (all active)
    const float* w;
    const float* w_rl;  // the layer's row-lane weight tail ([tap][p][co], spconv_rowlane.hip) behind the packed fragments, or null
    const float* bias;
    float* out;
    const float* res;
    uint32_t n_out;
    uint32_t row0;  // first output row computed (multiple of 16): rows [row0, n_out) -- insmos_sparse_conv_rows
    uint32_t in_bytes;  // extent of the `in` view: (n_in - 1) * ld_in * 4 + cin * 4
    int ld_in, cin, K, ld_out, cout, ld_res, res_mode, relu_pre, relu_post;
    int n16, has8, has4, nblk, ntile_co, n_otiles, vec_store;
    int tap_mod;  // tap-split tiles: 1 = wave ws owns the taps k with k % SPLIT == ws (a row's sum does not depend on which
                  // rows share its tile), 0 = every SPLIT-th ACTIVE tap of the tile (evenest load, tile-dependent order)
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// spconv_lds.hip: the LDS-staged kernel for the 81-tap single-chunk layers; false = not applicable (the caller goes on with the
// generic kernels), true = launched (rc holds the status)
bool conv_lds_ok(const ConvP& P, int ck, int cot);   // applicable (and not switched off)?
bool conv_lds_try(const ConvP& P, int ck, int cot, long n_rows, hipStream_t s, int* rc);

// spconv_rowlane.hip: the row-per-lane VALU kernel of the small-channel layers.  Floats of the row-lane tail a layer's packed
// weights carry behind the MFMA fragments (0 = the layer does not qualify); try = launch if applicable and enabled.
size_t rowlane_tail_floats(int K, int cin, int cout);
bool conv_rowlane_ok(const ConvP& P);
bool conv_rowlane_try(const ConvP& P, long n_rows, hipStream_t s, int* rc);
// spconv_row32.hip: the Cin = 32 layers with whole-row gathers (one 128-byte line per row) staged through LDS -- the kernel for a
// launch shape the dispatcher has already chosen (same tile -> block mapping, same bits), or null
typedef void (*ConvKernelFn)(ConvP);
ConvKernelFn conv_row32_pick(const ConvP& P, int cot, int jt, int split, bool by_chunk);
// spconv_wide.hip: the wide layers (Cin 64 / 128 / 256 -> Cout 64 / 128) the dispatcher would run on CHUNK-SPLIT tiles, on 32-row tiles
// with the gathered rows staged once per tap in LDS (same bits); *blocks = 320-thread workgroups to launch; null = not applicable
ConvKernelFn conv_wide_pick(const ConvP& P, long* blocks);
// [tap][p][co] position -> (ci, co) of the layer: p = 4 s + g walks the input channels in the MFMA kernels' chain order
__host__ __device__ inline int rowlane_ci(int cin, int p) { return (cin / 4) * (p & 3) + (p >> 2); }

}  // namespace insmos
