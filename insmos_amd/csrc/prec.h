// Reduced-precision MFMA variants of the convolution kernels (NEVER the default: the inference path is exact fp32).
//   insmos_conv_precision(1): both operands rounded to bf16, one v_mfma_f32_16x16x16_bf16 per 16-channel chunk, fp32
//       accumulate -- the opt-in for the training convolutions.
//   insmos_conv_precision(3): the "split-bf16 x 3" experiment: x = hi + lo (two bf16 = 16 mantissa bits),
//       x*y ~ hi*hi + hi*lo + lo*hi -- three bf16 MFMAs (48 cycles) instead of four fp32 ones (128 cycles) per chunk.  Weights
//       arrive pre-split (insmos_split_weights_bf16: hi4 | lo4 in the 16 bytes a lane holds), gathered rows are split in
//       registers (sparse kernel: 12 VALU per fragment, shared by the wave's channel tiles) or once per element when the halo
//       is staged into LDS (dense BEV kernel).
// Lane (g, .) holds channels 4g..4g+3 of its row / output channel in both the fp32 16x16x4 and the bf16 16x16x16 layouts
// (K index 4g + s), so the loads and the packed-weight indexing are those of the fp32 kernels.
#pragma once
#include "common.h"

namespace insmos {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // round to nearest even, a in the low half
    pf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ s16x4 round_bf16(pf32x4 x) {
    u32x2 h = {pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3])};
    return __builtin_bit_cast(s16x4, h);
}
__device__ __forceinline__ void split_bf16(pf32x4 x, s16x4& hi, s16x4& lo) {
    const unsigned h01 = pk_bf16(x[0], x[1]), h23 = pk_bf16(x[2], x[3]);
    const float r0 = x[0] - __builtin_bit_cast(float, h01 << 16), r1 = x[1] - __builtin_bit_cast(float, h01 & 0xFFFF0000u);
    const float r2 = x[2] - __builtin_bit_cast(float, h23 << 16), r3 = x[3] - __builtin_bit_cast(float, h23 & 0xFFFF0000u);
    u32x2 H = {h01, h23}, L = {pk_bf16(r0, r1), pk_bf16(r2, r3)};
    hi = __builtin_bit_cast(s16x4, H);
    lo = __builtin_bit_cast(s16x4, L);
}
// a lane's 16 bytes of a pre-split operand: (hi4 | lo4).  (Whole-vector bit casts: __builtin_bit_cast on a single vector
// ELEMENT made hipcc read element 0 for every index.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void unpack_split(pf32x4 raw, s16x4& hi, s16x4& lo) {
    const u32x4 r = __builtin_bit_cast(u32x4, raw);
    const u32x2 h = {r.x, r.y}, l = {r.z, r.w};
    hi = __builtin_bit_cast(s16x4, h);
    lo = __builtin_bit_cast(s16x4, l);
}
__device__ __forceinline__ pf32x4 pack_split(s16x4 hi, s16x4 lo) {
    const u32x2 h = __builtin_bit_cast(u32x2, hi), l = __builtin_bit_cast(u32x2, lo);
    const u32x4 r = {h.x, h.y, l.x, l.y};
    return __builtin_bit_cast(pf32x4, r);
}

// process-wide mode and the packed-fp32 -> split-weights table (spconv.hip)
int conv_precision();
int conv_precision_thread_get();   // this thread's override (-1 = none), for scoped pins
const void* split_weights_of(const float* wpacked);

}  // namespace insmos
