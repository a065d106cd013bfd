// tools/probes/mfma_rate_probe.hip -- what fp32 MFMA rate does an MI355X SUSTAIN, and does the instruction shape matter?  Every kernel of this
// library contracts on v_mfma_f32_16x16x4_f32 (2 operand dwords per lane per 2048 flops); v_mfma_f32_32x32x2_f32 moves the same two
// dwords per 4096 flops.  If register-file traffic is part of what the power limit sees, the larger tile sustains a higher clock.
// Pure issue loops (operands in registers, 8 independent accumulator chains per wave, 4 waves per SIMD), ~0.3 s per shape.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o /tmp/mfma_rate_probe && /tmp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    const int blocks = 256 * 4;   // 4 four-wave blocks per CU: 4 waves per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape = 0; shape < 2; ++shape) {
            const int iters = shape == 0 ? 400000 : 400000;   // per iteration: 8 x 2048 resp. 4 x 4096 flops per wave
            const double flops = (double)blocks * 4 * (double)iters * 8.0 * 2048.0;
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            else hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            printf("rep %d  %s  %.1f ms  %.1f TFLOP/s  (= %.3f of 157.3; implied MFMA clock %.0f MHz)\n", rep,
                   shape == 0 ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_32x32x2_f32", ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3,
                   flops / ms / 1e9 / 157.3 * 2400.0);
        }
    }
    return 0;
}
