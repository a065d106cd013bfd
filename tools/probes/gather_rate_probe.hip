// tools/probes/gather_rate_probe.hip -- the byte rates the convolution kernels' operand loads can reach on an MI355X, by where the data lives.
// One wave-instruction of each kind the kernels issue, in a loop with 8 loads in flight per wave, 8 one-wave workgroups... per CU:
//   gather16x64 : 16 rows x 64 B (lane (g, j) reads 16 B at row[j] * pitch + 16 g)   -- the B fragment of the MFMA tiles
//   gather64x32 : 64 rows x 16 B x 2 (one lane = one row, two b128)                  -- the row-lane kernel's gather (32-B rows)
//   stream1k    : 64 lanes x 16 B contiguous (1 KiB)                                  -- a weight fragment
// over row sets of 16 KiB (L1), 1 MiB (L2), 64 MiB (MALL) and 2 GiB (HBM); rows are picked by a per-wave LCG (no two lanes alike).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_rate_probe.hip -o /tmp/gather_rate_probe && /tmp/gather_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(64) k_probe(const float* __restrict__ base, uint32_t n_rows, uint32_t pitch_f, int iters, float* out) {
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    uint32_t state = (blockIdx.x * 2654435761u) ^ 0x9e3779b9u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            state = state * 1664525u + 1013904223u;             // wave-uniform
            uint32_t r;
            if (KIND == 0) r = (state >> 8) + (uint32_t)j * 2246822519u;          // 16 distinct rows per instruction
            else if (KIND == 1) r = (state >> 8) + (uint32_t)lane * 2246822519u;  // 64 distinct rows
            else if (KIND == 3) r = (state >> 8) + (uint32_t)(lane >> 3) * 2246822519u;   // 8 distinct rows, whole 128-B lines
            else if (KIND == 5) r = (state >> 8) + (uint32_t)j * 2246822519u;             // 16 rows, alternating halves of their lines
            else r = (state >> 8);                                                 // one 1 KiB run
            r %= n_rows;
            const float* p = KIND == 0 ? base + (size_t)r * pitch_f + 4 * g
                           : KIND == 1 ? base + (size_t)r * pitch_f + 4 * (u & 1)
                           : KIND == 3 ? base + (size_t)r * pitch_f + 4 * (lane & 7)
                           : KIND == 5 ? base + (size_t)r * pitch_f + 4 * g + 16 * (j & 1)
                                       : base + (size_t)(r & ~15u) * pitch_f + 4 * lane;   // (pitch 16 floats: 16 rows = 1 KiB)
            v[u] = *(const f32x4*)p;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[blockIdx.x] = acc[0];
}

int main() {
    const size_t max_bytes = (size_t)2 << 30;
    float* buf; float* out;
    hipMalloc(&buf, max_bytes + 4096); hipMalloc(&out, 1 << 20);
    hipMemset(buf, 0, max_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* kinds[6] = {"gather16x64 (MFMA B fragment)", "gather64x16 (row-lane, 32-B rows)", "stream 1 KiB (weight fragment)",
                            "gather8x128 (whole 128-B rows)", "gather16x64 of 128-B rows", "16x64, odd rows the other half"};
    const size_t sets[4] = {16 << 10, 1 << 20, 64 << 20, max_bytes};
    const char* where[4] = {"16 KiB (L1)", "1 MiB (L2)", "64 MiB (MALL)", "2 GiB (HBM)"};
    const int blocks = 256 * 16;   // 16 one-wave workgroups per CU
    for (int kind = 0; kind < 6; ++kind)
        for (int sidx = 0; sidx < 4; ++sidx) {
            const uint32_t pitch_f = kind == 1 ? 8 : kind >= 3 ? 32 : 16;   // floats per row: 32-B rows (row-lane), 128-B rows, else 64 B
            const uint32_t n_rows = (uint32_t)(sets[sidx] / (pitch_f * 4));
            const int iters = sidx == 3 ? 400 : 2000;
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(64), 0, 0, buf, n_rows, pitch_f, iters, out);
                if (kind == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(64), 0, 0, buf, n_rows, pitch_f, iters, out);
                if (kind == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(64), 0, 0, buf, n_rows, pitch_f, iters, out);
                if (kind == 3) hipLaunchKernelGGL(k_probe<3>, dim3(blocks), dim3(64), 0, 0, buf, n_rows, pitch_f, iters, out);
                if (kind == 4) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(64), 0, 0, buf, n_rows, pitch_f, iters, out);
                if (kind == 5) hipLaunchKernelGGL(k_probe<5>, dim3(blocks), dim3(64), 0, 0, buf, n_rows, pitch_f, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)blocks * iters * 8.0 * 1024.0;
            const double instr = (double)blocks * iters * 8.0;
            printf("%-34s %-14s %8.2f ms  %7.2f TB/s requested  %6.1f B/clk/CU at 2.1 GHz  %6.1f ns per wave-instruction per CU\n", kinds[kind],
                   where[sidx], ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / 2.1e9, ms * 1e6 / (instr / 256));
        }
    return 0;
}
