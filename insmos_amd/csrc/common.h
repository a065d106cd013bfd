// insmos_amd/csrc/common.h -- shared host/device helpers of libinsmos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/insmos_hip.h"

namespace insmos {

// ---- error plumbing -------------------------------------------------------------------------------
extern int g_last_hip_error;
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) {                         \
            insmos::g_last_hip_error = (int)_e;         \
            return INSMOS_EHIP;                         \
        }                                               \
    } while (0)
// hipGetLastError() is sticky per host thread: a benign failure inside ANOTHER library's HIP call (e.g. torch probing a
// pointer or counting devices) would otherwise surface at our next launch check.  Clear it right before launching.
#define INSMOS_LAUNCH(...)                \
    do {                                  \
        (void)hipGetLastError();          \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)

// ---- per-kernel profiler (HIP events on the launch stream) ----------------------------------------
enum KernelKind : int {
    KK_QUANT_KEYS = 0, KK_SORT, KK_SCAN, KK_QUANT_SCATTER, KK_LEVEL_DOWN, KK_BUILD_NBR, KK_VOX_KEYS,
    KK_VOX_SEGMENTS, KK_VOX_MEAN, KK_DOWN_CAND, KK_DOWN_UNIQUE, KK_SPARSE_CONV, KK_DENSE_NBR, KK_TO_BEV,
    KK_DECODE, KK_SELECT, KK_NMS_MASK, KK_NMS_REDUCE, KK_IOU, KK_GATHER_PREDS, KK_ONEHOT, KK_GATHER_ROWS,
    KK_CUR_POINTS, KK_FILL, KK_CONFUSION, KK_MEMSET, KK_BATCHNORM, KK_COUNT
};
struct ProfScope {
    int kind;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int64_t meta[4] = {0, 0, 0, 0};  // launch-site figures kept with the span (insmos_prof_read_spans)
    ProfScope(int kind, hipStream_t s);
    ~ProfScope();
};
bool prof_on();

// ---- workspace bump allocator ------------------------------------------------------------------------
struct Bump {
    char* base;
    size_t cap, off = 0;
    bool ok = true;
    Bump(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
    template <class T>
    T* take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) { ok = false; return nullptr; }
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
};
static inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// rocPRIM wrappers (coords.hip): sizes in bytes of the temp storage they need
size_t sort_pairs_u64_u32_temp(size_t n);
size_t sort_keys_u64_temp(size_t n);
size_t scan_i32_temp(size_t n);
int sort_pairs_u64_u32(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                       uint32_t* vout, size_t n, int begin_bit, int end_bit, hipStream_t s);
int sort_keys_u64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, size_t n, int begin_bit,
                  int end_bit, hipStream_t s);
size_t sort_keys_u64_radix_temp(size_t n);   // ... never by merging: radix passes over [begin_bit, end_bit) only, stable
int sort_keys_u64_radix(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, size_t n, int begin_bit, int end_bit,
                        hipStream_t s);
int inclusive_scan_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t s);

// ---- keys -------------------------------------------------------------------------------------------
#define INSMOS_KEY_BIAS 32768
#define INSMOS_INVALID_KEY 0xFFFFFFFFFFFFFFFFull

__host__ __device__ __forceinline__ uint64_t spread3(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__host__ __device__ __forceinline__ uint32_t compact3(uint64_t v) {
    v &= 0x1249249249249249ull;
    v = (v ^ (v >> 2)) & 0x10C30C30C30C30C3ull;
    v = (v ^ (v >> 4)) & 0x100F00F00F00F00Full;
    v = (v ^ (v >> 8)) & 0x1F0000FF0000FFull;
    v = (v ^ (v >> 16)) & 0x1F00000000FFFFull;
    v = (v ^ (v >> 32)) & 0x1FFFFFull;
    return (uint32_t)v;
}
// coords [x,y,z,t] (finest-voxel units) -> key; INVALID if outside the +-32768 window
__host__ __device__ __forceinline__ uint64_t key4_encode(int x, int y, int z, int t) {
    int bx = x + INSMOS_KEY_BIAS, by = y + INSMOS_KEY_BIAS, bz = z + INSMOS_KEY_BIAS, bt = t + INSMOS_KEY_BIAS;
    if (((unsigned)bx | (unsigned)by | (unsigned)bz | (unsigned)bt) & 0xFFFF0000u) return INSMOS_INVALID_KEY;
    return ((uint64_t)bt << 48) | spread3((uint64_t)bx) | (spread3((uint64_t)by) << 1) | (spread3((uint64_t)bz) << 2);
}
__host__ __device__ __forceinline__ void key4_decode(uint64_t k, int& x, int& y, int& z, int& t) {
    t = (int)(k >> 48) - INSMOS_KEY_BIAS;
    uint64_t m = k & 0xFFFFFFFFFFFFull;
    x = (int)compact3(m) - INSMOS_KEY_BIAS;
    y = (int)compact3(m >> 1) - INSMOS_KEY_BIAS;
    z = (int)compact3(m >> 2) - INSMOS_KEY_BIAS;
}
__host__ __device__ __forceinline__ uint64_t key3_encode(int z, int y, int x, int D, int H, int W) {
    if ((unsigned)z >= (unsigned)D || (unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return INSMOS_INVALID_KEY;
    return ((uint64_t)z * (uint64_t)H + (uint64_t)y) * (uint64_t)W + (uint64_t)x;
}

// ---- packed 4D sort keys (coords.hip, k_quant_keys_p): 40-bit key above a 24-bit point index; see the comment there
#define PK_IDX_BITS 24
#define PK_KEY_BITS 40
#define PK_KEY_MASK ((1ull << PK_KEY_BITS) - 1ull)
__host__ __device__ __forceinline__ uint64_t spread2(uint64_t v) {  // 3 bits -> every other bit
    return (v & 1) | ((v & 2) << 1) | ((v & 4) << 2);
}
__host__ __device__ __forceinline__ uint64_t pkey_make(int tprime_biased, int x, int y, int z) {
    const uint64_t ux = (uint64_t)(x + INSMOS_KEY_BIAS), uy = (uint64_t)(y + INSMOS_KEY_BIAS), uz = (uint64_t)(z + INSMOS_KEY_BIAS);
    const uint64_t hi_xy = spread2((ux >> 8) & 7) | (spread2((uy >> 8) & 7) << 1);            // y10 x10 y9 x9 y8 x8
    const uint64_t lo = spread3(ux & 0xFF) | (spread3(uy & 0xFF) << 1) | (spread3(uz & 0xFF) << 2);
    return ((uint64_t)tprime_biased << 33) | ((uint64_t)(z >= 0) << 32) | ((uint64_t)(y >= 0) << 31) | ((uint64_t)(x >= 0) << 30) |
           (hi_xy << 24) | lo;
}
__host__ __device__ __forceinline__ uint64_t pkey_expand(uint64_t p, int B) {  // 40-bit packed key -> canonical key
    const uint64_t bt = (p >> 33) - (uint64_t)(15 * B) + INSMOS_KEY_BIAS;
    const uint64_t lo = p & 0xFFFFFFull, hx = (p >> 24) & 0x3F;
    const uint64_t xh = (hx & 1) | ((hx >> 1) & 2) | ((hx >> 2) & 4), yh = ((hx >> 1) & 1) | ((hx >> 2) & 2) | ((hx >> 3) & 4);
    const uint64_t ux = (((p >> 30) & 1) ? 0x8000u : 0x7800u) | (xh << 8) | compact3(lo);
    const uint64_t uy = (((p >> 31) & 1) ? 0x8000u : 0x7800u) | (yh << 8) | compact3(lo >> 1);
    const uint64_t uz = (((p >> 32) & 1) ? 0x8000u : 0x7F00u) | compact3(lo >> 2);
    return (bt << 48) | spread3(ux) | (spread3(uy) << 1) | (spread3(uz) << 2);
}

// 3D key with the spconv batch column: windows of one batch are stacked along a leading axis, key = b * cells + lin
__host__ __device__ __forceinline__ uint64_t key3b_encode(int b, int z, int y, int x, int D, int H, int W) {
    const uint64_t k = key3_encode(z, y, x, D, H, W);
    return k == INSMOS_INVALID_KEY ? k : k + (uint64_t)b * ((uint64_t)D * (uint64_t)H * (uint64_t)W);
}

// ---- several windows in one launch set ----------------------------------------------
// The point clouds of the B windows of a batch stay where the caller has them: kernels that read points take this table
// by value (kernel-argument memory: every access below is a wave-uniform scalar load) and locate a global point index
// i in [0, start[B]) with a short select chain.
#define INSMOS_MAX_BATCH 16
struct WinPts {
    const float* p[INSMOS_MAX_BATCH];
    int64_t start[INSMOS_MAX_BATCH + 1];
    int B;
};
__device__ __forceinline__ const float* win_point(const WinPts& W, int64_t i, int ld, int& b) {
    const float* base = W.p[0];
    int64_t s0 = 0;
    b = 0;
#pragma unroll
    for (int q = 1; q < INSMOS_MAX_BATCH; ++q)
        if (q < W.B && i >= W.start[q]) { b = q; base = W.p[q]; s0 = W.start[q]; }
    return base + (i - s0) * ld;
}
// window of row i given B+1 ascending row starts in device memory (B <= INSMOS_MAX_BATCH; the loads are wave-uniform)
__device__ __forceinline__ int win_of_row(const int32_t* __restrict__ starts, int B, int64_t i) {
    int b = 0;
    for (int q = 1; q < B; ++q)
        if (i >= starts[q]) b = q;
    return b;
}
int make_win_pts(const float* const* pts_host, const int64_t* n_pts_host, int B, WinPts* out, int64_t* total);

// lower-bound binary search; returns position or -1
__device__ __forceinline__ int64_t find_key(const uint64_t* __restrict__ keys, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && keys[lo] == key) ? lo : -1;
}

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace insmos
