// insmos_amd/csrc/coords.hip -- coordinate-set kernels: 4D quantise/de-dup, strided levels, 3D hard
// voxelisation + mean VFE, strided-conv output sets, and the output-stationary neighbour tables.
//
// Design (MI355X-first): no hash tables.  Every coordinate set is a radix-sorted array
// of 64-bit keys; de-duplication is an adjacent-difference + prefix sum over the sorted array (wide
// coalesced streams), strided 4D levels fall out of the Morton key by a shift, and kernel maps are
// binary searches over the sorted keys whose upper levels stay in L2.  Radix sort and prefix scan are
// rocPRIM device primitives (commodity building blocks); everything domain-specific is hand-written.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "common.h"

namespace insmos {

// ---------------------------------------------------------------------------------------------------
// rocPRIM wrappers
// ---------------------------------------------------------------------------------------------------
size_t sort_pairs_u64_u32_temp(size_t n) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    return pad256(b);
}
size_t sort_keys_u64_temp(size_t n) {
    size_t b = 0;
    (void)rocprim::radix_sort_keys(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64, (hipStream_t)0);
    return pad256(b);
}
size_t scan_i32_temp(size_t n) {
    size_t b = 0;
    (void)rocprim::inclusive_scan(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, n, rocprim::plus<int32_t>(),
                                  (hipStream_t)0);
    return pad256(b);
}
int sort_pairs_u64_u32(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                       uint32_t* vout, size_t n, int b0, int b1, hipStream_t s) {
    ProfScope ps(KK_SORT, s);
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, (unsigned)b0, (unsigned)b1, s));
    return INSMOS_OK;
}
int sort_keys_u64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, size_t n, int b0, int b1,
                  hipStream_t s) {
    ProfScope ps(KK_SORT, s);
    HIP_TRY(rocprim::radix_sort_keys(tmp, tmp_bytes, kin, kout, n, (unsigned)b0, (unsigned)b1, s));
    return INSMOS_OK;
}
// The same sort with rocPRIM's merge-sort path switched off: below 2^20 keys the default configuration sorts by merging -- a launch
// per merge level (~40 for the voxeliser's ~1 M cell keys of a launch set, 260 + 50 us) over ALL 64 key bits, whatever bit range
// was asked for -- where the radix (Onesweep) path makes one pass per 8 bits of [b0, b1) and is stable
using radix_only_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
size_t sort_keys_u64_radix_temp(size_t n) {
    size_t b = 0;
    (void)rocprim::radix_sort_keys<radix_only_config>(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64, (hipStream_t)0);
    return pad256(b);
}
int sort_keys_u64_radix(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, size_t n, int b0, int b1, hipStream_t s) {
    ProfScope ps(KK_SORT, s);
    HIP_TRY(rocprim::radix_sort_keys<radix_only_config>(tmp, tmp_bytes, kin, kout, n, (unsigned)b0, (unsigned)b1, s));
    return INSMOS_OK;
}
int inclusive_scan_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t s) {
    ProfScope ps(KK_SCAN, s);
    HIP_TRY(rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::plus<int32_t>(), s));
    return INSMOS_OK;
}

static inline int bits_for(uint64_t maxval) {
    int b = 1;
    while (b < 64 && (maxval >> b)) ++b;
    return b;
}

// ---------------------------------------------------------------------------------------------------
// 4D quantise (motionnet.py:22-36)
// ---------------------------------------------------------------------------------------------------
// COMPACT SORT KEYS.  A LiDAR window lives within +-2048 voxels and 16 time steps, where the 64-bit canonical key has
// only 40 informative bits: bits 15..11 of a biased coordinate are a function of its sign.  ckey = (t+15) << 36 |
// (z>=0, y>=0, x>=0) << 33 | morton3(low 11 bits) orders exactly like the canonical key, so the radix sort runs 5 byte
// passes instead of 8; the canonical key is rebuilt after the sort.  Points outside that box are counted (counts[3]):
// the caller then repeats the call with the full-width sort.
__host__ __device__ __forceinline__ uint64_t ckey_expand(uint64_t c) {
    const uint64_t bt = (c >> 36) - 15 + INSMOS_KEY_BIAS;
    const uint64_t m = c & 0x1FFFFFFFFull;
    const uint64_t ux = (((c >> 33) & 1) ? 0x8000u : 0x7800u) | compact3(m);
    const uint64_t uy = (((c >> 34) & 1) ? 0x8000u : 0x7800u) | compact3(m >> 1);
    const uint64_t uz = (((c >> 35) & 1) ? 0x8000u : 0x7800u) | compact3(m >> 2);
    return (bt << 48) | spread3(ux) | (spread3(uy) << 1) | (spread3(uz) << 2);
}

template <bool COMPACT>
__global__ void k_quant_keys(const float* __restrict__ pts, int64_t n, int ld, float q0, float q1, float q2, float q3,
                             uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, int32_t* __restrict__ tflag,
                             int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + i * ld;
    // IEEE fp32 division then floor -- exactly torch.div(point_cloud, quantization) + ME's floor
    float fx = p[0] / q0, fy = p[1] / q1, fz = p[2] / q2, ft = p[4] / q3;
    const int x = (int)floorf(fx), y = (int)floorf(fy), z = (int)floorf(fz), t = (int)floorf(ft);
    uint64_t k = key4_encode(x, y, z, t);
    if (k == INSMOS_INVALID_KEY) atomicAdd(&counts[2], 1);
    if (COMPACT && k != INSMOS_INVALID_KEY) {
        if (x < -2048 || x > 2047 || y < -2048 || y > 2047 || z < -2048 || z > 2047 || t < -15 || t > 0) {
            atomicAdd(&counts[3], 1);
            k = INSMOS_INVALID_KEY;
        } else {
            const uint64_t lx = (uint64_t)(x + INSMOS_KEY_BIAS) & 0x7FF, ly = (uint64_t)(y + INSMOS_KEY_BIAS) & 0x7FF,
                           lz = (uint64_t)(z + INSMOS_KEY_BIAS) & 0x7FF;
            k = ((uint64_t)(t + 15) << 36) | ((uint64_t)(z >= 0) << 35) | ((uint64_t)(y >= 0) << 34) |
                ((uint64_t)(x >= 0) << 33) | spread3(lx) | (spread3(ly) << 1) | (spread3(lz) << 2);
        }
    }
    keys[i] = k;
    idx[i] = (uint32_t)i;
    tflag[i] = (ft == 0.0f) ? 1 : 0;
}

// ---- several windows in ONE coordinate set: the window index b is folded into the time
// coordinate, t' = floor(t / dt) * B + b.  Time has no bounds and no striding and every table kernel treats time taps
// symbolically, so with the searched (coarsest) table built on time offsets scaled by B nothing else has to know about
// windows: rows stay time-major (the newest scans of ALL windows are one suffix), windows never share a neighbour.
// Compact sort keys: the time field holds (tq + 15) * B + b in [0, 16 B) instead of tq + 15 in [0, 16).
__host__ __device__ __forceinline__ uint64_t ckey_expand_b(uint64_t c, int B) {
    const uint64_t bt = (c >> 36) - (uint64_t)(15 * B) + INSMOS_KEY_BIAS;
    const uint64_t m = c & 0x1FFFFFFFFull;
    const uint64_t ux = (((c >> 33) & 1) ? 0x8000u : 0x7800u) | compact3(m);
    const uint64_t uy = (((c >> 34) & 1) ? 0x8000u : 0x7800u) | compact3(m >> 1);
    const uint64_t uz = (((c >> 35) & 1) ? 0x8000u : 0x7800u) | compact3(m >> 2);
    return (bt << 48) | spread3(ux) | (spread3(uy) << 1) | (spread3(uz) << 2);
}

// PACKED SORT KEYS (round 3).  The radix sort moves key AND payload through every pass; with the point index carried in the low
// bits of the key itself the sort is keys-only (8 instead of 12 bytes per element and pass).  Room for it comes from z: a LiDAR
// window spans +-256 voxels in z (+-25.6 m at 0.1 m), so z keeps sign + 8 low bits (bits 14..8 of the biased coordinate follow
// the sign): pkey = [t' : 7][z>=0, y>=0, x>=0][y10 x10 y9 x9 y8 x8][morton3 of the low 8 bits : 24] = 40 bits, ordered exactly
// like the canonical key, above a 24-bit point index -- 5 byte passes over 8-byte elements for a set of <= 8 windows and
// < 2^24 points (6 passes over 12-byte pairs before).  Points outside the box are counted (counts[3]) and the caller falls
// back to the pair sort.
__global__ void k_quant_keys_p(WinPts W, int64_t n, int ld, float q0, float q1, float q2, float q3,
                               uint64_t* __restrict__ keys, int32_t* __restrict__ tflag, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b;
    const float* p = win_point(W, i, ld, b);
    const int B = W.B;
    float fx = p[0] / q0, fy = p[1] / q1, fz = p[2] / q2, ft = p[4] / q3;   // the same fp32 ops as k_quant_keys
    const int x = (int)floorf(fx), y = (int)floorf(fy), z = (int)floorf(fz), tq = (int)floorf(ft);
    uint64_t k;
    if (x < -2048 || x > 2047 || y < -2048 || y > 2047 || z < -256 || z > 255 || tq < -15 || tq > 0) {
        atomicAdd(&counts[3], 1);   // (includes everything outside the +-32768 key window: the fallback sorts count those)
        k = ~0ull;
    } else {
        k = ((uint64_t)i << PK_KEY_BITS) | pkey_make((tq + 15) * B + b, x, y, z);
    }
    keys[i] = k;
    tflag[i] = (ft == 0.0f) ? 1 : 0;
}

__global__ void k_head_flags_p(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];   // (all-ones = a point outside the packed box: the caller repeats the call in another mode)
    flag[i] = (k != ~0ull) && ((i == 0) || ((k & PK_KEY_MASK) != (keys[i - 1] & PK_KEY_MASK)));
}

__global__ void k_quant_scatter_p(const uint64_t* __restrict__ keys_s, const int32_t* __restrict__ flag,
                                  const int32_t* __restrict__ scan, int64_t n, int B, uint64_t* __restrict__ vkeys,
                                  int32_t* __restrict__ vcoords, int32_t* __restrict__ inverse, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ks = keys_s[i];
    const int vid = scan[i] - 1;
    if (i == n - 1) counts[0] = scan[i];
    if (ks == ~0ull) return;
    if (flag[i]) {
        const uint64_t k = pkey_expand(ks & PK_KEY_MASK, B);
        vkeys[vid] = k;
        int x, y, z, t;
        key4_decode(k, x, y, z, t);
        *(int4*)(vcoords + (int64_t)vid * 4) = make_int4(x, y, z, t);
    }
    inverse[ks >> PK_KEY_BITS] = vid;
}

template <bool COMPACT>
__global__ void k_quant_keys_b(WinPts W, int64_t n, int ld, float q0, float q1, float q2, float q3,
                               uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, int32_t* __restrict__ tflag,
                               int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b;
    const float* p = win_point(W, i, ld, b);
    const int B = W.B;
    float fx = p[0] / q0, fy = p[1] / q1, fz = p[2] / q2, ft = p[4] / q3;   // the same fp32 ops as k_quant_keys
    const int x = (int)floorf(fx), y = (int)floorf(fy), z = (int)floorf(fz), tq = (int)floorf(ft);
    const int tlim = 32768 / B - 1;                                         // |t'| = |tq * B + b| must fit the 16-bit biased
    const bool t_ok = tq >= -tlim && tq <= tlim;                            // time field: a single window gets all of it
    uint64_t k = t_ok ? key4_encode(x, y, z, tq * B + b) : INSMOS_INVALID_KEY;
    if (k == INSMOS_INVALID_KEY) atomicAdd(&counts[2], 1);
    if (COMPACT && k != INSMOS_INVALID_KEY) {
        if (x < -2048 || x > 2047 || y < -2048 || y > 2047 || z < -2048 || z > 2047 || tq < -15 || tq > 0) {
            atomicAdd(&counts[3], 1);
            k = INSMOS_INVALID_KEY;
        } else {
            const uint64_t lx = (uint64_t)(x + INSMOS_KEY_BIAS) & 0x7FF, ly = (uint64_t)(y + INSMOS_KEY_BIAS) & 0x7FF,
                           lz = (uint64_t)(z + INSMOS_KEY_BIAS) & 0x7FF;
            k = ((uint64_t)((tq + 15) * B + b) << 36) | ((uint64_t)(z >= 0) << 35) | ((uint64_t)(y >= 0) << 34) |
                ((uint64_t)(x >= 0) << 33) | spread3(lx) | (spread3(ly) << 1) | (spread3(lz) << 2);
        }
    }
    keys[i] = k;
    idx[i] = (uint32_t)i;
    tflag[i] = (ft == 0.0f) ? 1 : 0;
}

// cur_start[b] = number of current-scan points (t == 0) in the windows before b: the inclusive scan of the t-flags read at
// the window boundaries.  out[0 .. B] (out[B] = all of them)
__global__ void k_cur_starts(WinPts W, const int32_t* __restrict__ tscan, int32_t* __restrict__ out) {
    const int b = threadIdx.x;
    if (b > W.B) return;
    int64_t s = 0;
#pragma unroll
    for (int q = 1; q <= INSMOS_MAX_BATCH; ++q)
        if (q == b) s = W.start[q];
    out[b] = s > 0 ? tscan[s - 1] : 0;
}

template <bool COMPACT>
__global__ void k_quant_scatter_b(const uint64_t* __restrict__ keys_s, const uint32_t* __restrict__ idx_s,
                                  const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, int64_t n, int B,
                                  uint64_t* __restrict__ vkeys, int32_t* __restrict__ vcoords,
                                  int32_t* __restrict__ inverse, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys_s[i];
    int vid = scan[i] - 1;
    if (k != INSMOS_INVALID_KEY) {
        if (flag[i]) {
            if (COMPACT) k = ckey_expand_b(k, B);
            vkeys[vid] = k;
            int x, y, z, t;
            key4_decode(k, x, y, z, t);
            *(int4*)(vcoords + (int64_t)vid * 4) = make_int4(x, y, z, t);   // t column = t' (window index folded in)
        }
        inverse[idx_s[i]] = vid;
    } else {
        inverse[idx_s[i]] = -1;
    }
    if (i == n - 1) counts[0] = scan[i];
}

__global__ void k_head_flags(const uint64_t* __restrict__ keys, int64_t n, int shift_bits, int32_t* __restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys[i];
    int f = 0;
    if (k != INSMOS_INVALID_KEY) f = (i == 0) || ((k >> shift_bits) != (keys[i - 1] >> shift_bits));
    flag[i] = f;
}

template <bool COMPACT>
__global__ void k_quant_scatter(const uint64_t* __restrict__ keys_s, const uint32_t* __restrict__ idx_s,
                                const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, int64_t n,
                                uint64_t* __restrict__ vkeys, int32_t* __restrict__ vcoords,
                                int32_t* __restrict__ inverse, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys_s[i];
    int vid = scan[i] - 1;
    if (k != INSMOS_INVALID_KEY) {
        if (flag[i]) {
            if (COMPACT) k = ckey_expand(k);
            vkeys[vid] = k;
            int x, y, z, t;
            key4_decode(k, x, y, z, t);
            *(int4*)(vcoords + (int64_t)vid * 4) = make_int4(x, y, z, t);
        }
        inverse[idx_s[i]] = vid;
    } else {
        inverse[idx_s[i]] = -1;
    }
    if (i == n - 1) counts[0] = scan[i];
}

__global__ void k_compact_index(const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, int64_t n,
                                int32_t* __restrict__ out_idx, int32_t* __restrict__ count_slot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) out_idx[scan[i] - 1] = (int32_t)i;
    if (i == n - 1) *count_slot = scan[i];
}

__global__ void k_level_down_scatter(const uint64_t* __restrict__ keys, const int32_t* __restrict__ flag,
                                     const int32_t* __restrict__ scan, int64_t n, int shift_bits,
                                     uint64_t* __restrict__ okeys, int32_t* __restrict__ ocoords,
                                     int32_t* __restrict__ parent, int32_t* __restrict__ child_start,
                                     uint32_t* __restrict__ child_mask, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int vid = scan[i] - 1;
    if (flag[i]) {
        uint64_t k = (keys[i] >> shift_bits) << shift_bits;
        okeys[vid] = k;
        int x, y, z, t;
        key4_decode(k, x, y, z, t);
        *(int4*)(ocoords + (int64_t)vid * 4) = make_int4(x, y, z, t);
        if (child_start) child_start[vid] = (int32_t)i;
        if (child_mask) {
            // the children of a coarse voxel are the <= 8 consecutive fine rows that share its key prefix; octant of a fine
            // voxel inside its parent = the 3 Morton bits just below the parent's stride.  The head row collects them (plain
            // store: no atomics, no zero-fill of the array)
            uint32_t m = 0;
            for (int j = 0; j < 8 && i + j < n; ++j) {
                const uint64_t kj = keys[i + j];
                if ((kj >> shift_bits) != (k >> shift_bits)) break;
                m |= 1u << (unsigned)((kj >> (shift_bits - 3)) & 7ull);
            }
            // bits 8..31: the first child row again, whenever it fits (n < 2^24) -- the table kernels then fetch ONE word per
            // coarse neighbour instead of two (k_resolve_taps); readers of the mask alone take the low byte
            child_mask[vid] = n < (1ll << 24) ? (m | ((uint32_t)i << 8)) : m;
        }
    }
    parent[i] = vid;
    if (i == n - 1) counts[0] = scan[i];
}

// ---------------------------------------------------------------------------------------------------
// search-free neighbour tables from the Morton hierarchy
// ---------------------------------------------------------------------------------------------------
struct TapList { int delta[128][4]; int K; };

// nbr[k][o] for fine voxels (tensor stride 2^L) from the COARSE level's 3x3x3x3 table: the neighbour's
// parent block is one of the 27(x3 in time) blocks around the voxel's own parent; inside that block the
// children are contiguous in Morton order, so its row is child_start + popcount(mask below its octant).
__global__ void __launch_bounds__(256) k_nbr_from_coarse(const int32_t* __restrict__ coords, int64_t n_f,
                                                         const int32_t* __restrict__ parent, int L,
                                                         const int32_t* __restrict__ cnbr, int64_t n_c,
                                                         const int32_t* __restrict__ child_start,
                                                         const uint32_t* __restrict__ child_mask, TapList T,
                                                         int32_t* __restrict__ nbr, uint32_t* __restrict__ mask16) {
    int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (o >= n_f) return;
    int4 c = *(const int4*)(coords + o * 4);
    const int nx = c.x + T.delta[k][0], ny = c.y + T.delta[k][1], nz = c.z + T.delta[k][2], dt = T.delta[k][3];
    const int s1 = L + 1;
    const int bx = (nx >> s1) - (c.x >> s1), by = (ny >> s1) - (c.y >> s1), bz = (nz >> s1) - (c.z >> s1);
    int32_t r = -1;
    if (bx >= -1 && bx <= 1 && by >= -1 && by <= 1 && bz >= -1 && bz <= 1 && dt >= -1 && dt <= 1) {
        const int ctap = (bx + 1) + 3 * (by + 1) + 9 * (bz + 1) + 27 * (dt + 1);
        const int q = cnbr[(int64_t)ctap * n_c + parent[o]];
        if (q >= 0) {
            const unsigned oct = ((nx >> L) & 1) | (((ny >> L) & 1) << 1) | (((nz >> L) & 1) << 2);
            const uint32_t m = child_mask[q] & 0xFFu;   // (bits 8..31 may carry child_start, see k_level_down_scatter)
            if ((m >> oct) & 1u) r = child_start[q] + __popc(m & ((1u << oct) - 1u));
        }
    }
    nbr[(int64_t)k * n_f + o] = r;
    if (mask16) {
        const unsigned long long bal = __ballot(r >= 0);
        const int lane = threadIdx.x & 63;
        if ((lane & 15) == 0 && ((bal >> (lane & 48)) & 0xFFFFull)) atomicOr(&mask16[(o >> 4) * 4 + (k >> 5)], 1u << (k & 31));
    }
}

// strided k2s2 conv (coarse p reads child octant k) and its transpose (fine f reads its parent through k = octant(f)).
// One thread = one row, all 8 taps: the 16-row group's active-tap mask is a ballot per tap, written once by the group's first
// lane (all four words: no atomics, no zero-fill of the mask array).
__device__ __forceinline__ void write_mask8(const int32_t (&r)[8], int64_t row, int64_t n, uint32_t* __restrict__ mask16) {
    uint32_t mk = 0;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long bal = __ballot(r[k] >= 0);
        if ((bal >> (lane & 48)) & 0xFFFFull) mk |= 1u << k;
    }
    if (mask16 && (lane & 15) == 0 && row < n) *(uint4*)(mask16 + (row >> 4) * 4) = make_uint4(mk, 0u, 0u, 0u);
}
__global__ void __launch_bounds__(256) k_nbr_down(int64_t n_c, const int32_t* __restrict__ child_start,
                                                  const uint32_t* __restrict__ child_mask, int32_t* __restrict__ dn,
                                                  uint32_t* __restrict__ mask16) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = p < n_c;
    const uint32_t m = ok ? (child_mask[p] & 0xFFu) : 0u;
    const int32_t cs = ok ? child_start[p] : 0;
    int32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r[k] = ((m >> k) & 1u) ? cs + __popc(m & ((1u << k) - 1u)) : -1;
        if (ok) dn[(int64_t)k * n_c + p] = r[k];
    }
    write_mask8(r, p, n_c, mask16);
}
__global__ void __launch_bounds__(256) k_nbr_up(const int32_t* __restrict__ coords, int64_t n_f,
                                                const int32_t* __restrict__ parent, int L, int32_t* __restrict__ up,
                                                uint32_t* __restrict__ mask16) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = f < n_f;
    int oct = -1, par = -1;
    if (ok) {
        const int4 c = *(const int4*)(coords + f * 4);
        oct = ((c.x >> L) & 1) | (((c.y >> L) & 1) << 1) | (((c.z >> L) & 1) << 2);
        par = parent[f];
    }
    int32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r[k] = (oct == k) ? par : -1;
        if (ok) up[(int64_t)k * n_f + f] = r[k];
    }
    write_mask8(r, f, n_f, mask16);
}

// ---------------------------------------------------------------------------------------------------
// neighbour tables
// ---------------------------------------------------------------------------------------------------
struct NbrParams {
    int delta[128][4];
    int mul[4];
    int dv[4];
    int shape[3];
    int K;
};

template <int KEY_MODE>
__global__ void __launch_bounds__(256) k_build_nbr(const int32_t* __restrict__ out_coords, int64_t n_out,
                                                   const uint64_t* __restrict__ in_keys,
                                                   const int32_t* __restrict__ in_perm, int64_t n_in, NbrParams P,
                                                   int32_t* __restrict__ nbr, uint32_t* __restrict__ mask16) {
    int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (o >= n_out) return;
    int4 c = *(const int4*)(out_coords + o * 4);
    int q[4] = {c.x * P.mul[0] + P.delta[k][0], c.y * P.mul[1] + P.delta[k][1], c.z * P.mul[2] + P.delta[k][2],
                c.w * P.mul[3] + P.delta[k][3]};
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int dv = P.dv[d];
        if (dv > 1) {
            // exact divisibility (floor semantics irrelevant once divisible)
            if (q[d] % dv != 0) ok = false;
            q[d] = q[d] / dv;
        }
    }
    int32_t r = -1;
    if (ok) {
        uint64_t key = (KEY_MODE == 0) ? key4_encode(q[0], q[1], q[2], q[3])
                                       : key3b_encode(q[0], q[1], q[2], q[3], P.shape[0], P.shape[1], P.shape[2]);
        if (key != INSMOS_INVALID_KEY) {
            int64_t pos = find_key(in_keys, n_in, key);
            if (pos >= 0) r = in_perm ? in_perm[pos] : (int32_t)pos;
        }
    }
    nbr[(int64_t)k * n_out + o] = r;
    if (mask16) {
        // active-tap bitmask per 16-row group (rows of a group are 16 consecutive lanes)
        const unsigned long long bal = __ballot(r >= 0);
        const int lane = threadIdx.x & 63;
        if ((lane & 15) == 0 && ((bal >> (lane & 48)) & 0xFFFFull)) atomicOr(&mask16[(o >> 4) * 4 + (k >> 5)], 1u << (k & 31));
    }
}

// The same table for tap lists whose taps come in x-TRIPLES (kx fastest, x offsets d, d+1, d+2, no division on x: the
// submanifold and the strided 3x3x3 maps): the three probes are consecutive integers in key space, so ONE lower-bound search
// finds the first and the other two are the next entries of the sorted array -- a third of the searches.  grid.y = K / 3.
__global__ void __launch_bounds__(256) k_build_nbr_x3(const int32_t* __restrict__ out_coords, int64_t n_out,
                                                      const uint64_t* __restrict__ in_keys,
                                                      const int32_t* __restrict__ in_perm, int64_t n_in, NbrParams P,
                                                      int32_t* __restrict__ nbr, uint32_t* __restrict__ mask16) {
    int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k0 = 3 * blockIdx.y;
    if (o >= n_out) return;
    int4 c = *(const int4*)(out_coords + o * 4);
    int q[4] = {c.x * P.mul[0] + P.delta[k0][0], c.y * P.mul[1] + P.delta[k0][1], c.z * P.mul[2] + P.delta[k0][2],
                c.w * P.mul[3] + P.delta[k0][3]};
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {  // (x is never divided here)
        int dv = P.dv[d];
        if (dv > 1) {
            if (q[d] % dv != 0) ok = false;
            q[d] = q[d] / dv;
        }
    }
    const int D = P.shape[0], H = P.shape[1], W = P.shape[2];
    ok = ok && (unsigned)q[1] < (unsigned)D && (unsigned)q[2] < (unsigned)H && q[3] + 2 >= 0 && q[3] < W;
    int32_t r[3] = {-1, -1, -1};
    if (ok) {
        const uint64_t row_key = (((uint64_t)q[0] * D + (uint64_t)q[1]) * H + (uint64_t)q[2]) * (uint64_t)W;  // key of x = 0
        const int xs = q[3] < 0 ? 0 : q[3];
        int64_t lo = 0, hi = n_in;  // lower_bound(row_key + xs)
        const uint64_t want = row_key + (uint64_t)xs;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (in_keys[mid] < want) lo = mid + 1; else hi = mid;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int x = q[3] + t;
            if (x >= xs && x < W && lo < n_in && in_keys[lo] == row_key + (uint64_t)x) {
                r[t] = in_perm ? in_perm[lo] : (int32_t)lo;
                ++lo;
            }
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = k0 + t;
        nbr[(int64_t)k * n_out + o] = r[t];
        if (mask16) {
            const unsigned long long bal = __ballot(r[t] >= 0);
            if ((lane & 15) == 0 && ((bal >> (lane & 48)) & 0xFFFFull)) atomicOr(&mask16[(o >> 4) * 4 + (k >> 5)], 1u << (k & 31));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 3D hard voxelisation + mean VFE (voxel_generate.py:19-28, mean_vfe.py:47-52)
// ---------------------------------------------------------------------------------------------------
#define VOX_IDX_BITS 23

// Windows of a batch (B >= 1): points are window-major, win_start[0 .. B] (device) are the windows' first points; the
// window index is the leading digit of the cell key (key = b * cells + lin), first-seen order and the voxel cap are PER
// WINDOW, as the reference calls VoxelGenerate once per batch item (models/models.py:326).
__global__ void k_vox_keys(const float* __restrict__ pts, int64_t n, int ld, const int32_t* __restrict__ win_start, int B,
                           uint64_t key_cells, float lx, float ly, float lz, float vx, float vy, float vz, int gx, int gy, int gz,
                           uint64_t* __restrict__ keys, int64_t* __restrict__ pcid, int32_t* __restrict__ mark,
                           int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + i * ld;
    // spconv point-to-voxel: c = floor((p - lo) / vsize) in fp32, keep iff 0 <= c < grid
    float cx = floorf((p[0] - lx) / vx), cy = floorf((p[1] - ly) / vy), cz = floorf((p[2] - lz) / vz);
    bool in = cx >= 0.f && cx < (float)gx && cy >= 0.f && cy < (float)gy && cz >= 0.f && cz < (float)gz;
    uint64_t k = INSMOS_INVALID_KEY;
    if (in) {
        const uint64_t b = win_start ? (uint64_t)win_of_row(win_start, B, i) : 0ull;
        uint64_t lin = ((uint64_t)(int)cz * (uint64_t)gy + (uint64_t)(int)cy) * (uint64_t)gx + (uint64_t)(int)cx;
        lin += b * key_cells;
        k = (lin << VOX_IDX_BITS) | (uint64_t)i;
    }
    // (the number of in-range points is NOT counted here: 15 000 waves adding to one address serialise in the L2, 170 us
    //  for a launch set; k_vox_heads reads it off the sorted keys instead)
    keys[i] = k;
    pcid[i] = -1;
    mark[i] = 0;
}

__global__ void k_vox_heads(const uint64_t* __restrict__ keys_s, const int32_t* __restrict__ flag,
                            const int32_t* __restrict__ scan, int64_t n, int32_t* __restrict__ seg_start,
                            int32_t* __restrict__ seg_first, int32_t* __restrict__ mark, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // in-range points = the sorted keys before the first invalid one (counts[2] was zeroed: stays 0 when there is none)
    if (keys_s[i] != INSMOS_INVALID_KEY && (i == n - 1 || keys_s[i + 1] == INSMOS_INVALID_KEY)) counts[2] = (int32_t)(i + 1);
    if (flag[i]) {
        int sid = scan[i] - 1;
        int p = (int)(keys_s[i] & ((1ull << VOX_IDX_BITS) - 1));
        seg_start[sid] = (int32_t)i;
        seg_first[sid] = p;
        mark[p] = 1;
    }
}

// per-window first-seen base ranks and voxel row starts (one small block):
//   woff[b] = voxels first seen in the windows before b, woff[B+1+b] = first voxel ROW of window b (caps applied)
__global__ void k_vox_offsets(const int32_t* __restrict__ win_start, int B, int64_t n, const int32_t* __restrict__ rank_scan,
                              const int32_t* __restrict__ sid_scan, int max_voxels, int32_t* __restrict__ woff,
                              int32_t* __restrict__ counts, int32_t* __restrict__ kept_state) {
    if (threadIdx.x != 0) return;
    kept_state[0] = sid_scan[n - 1];
    int row = 0;
    for (int b = 0; b <= B; ++b) {
        const int64_t s = b == B ? n : (win_start ? (int64_t)win_start[b] : 0);
        const int base = s > 0 ? rank_scan[s - 1] : 0;
        woff[b] = base;
        if (b > 0) {
            const int seen = base - woff[b - 1];
            row += seen < max_voxels ? seen : max_voxels;
        }
        woff[B + 1 + b] = row;
        if (win_start) counts[4 + b] = row;
    }
    counts[0] = row;               // voxel rows kept
    counts[1] = sid_scan[n - 1];   // occupied cells (search-structure entries, dropped ones included)
}

template <bool FEATS>
__global__ void k_vox_segments(const float* __restrict__ pts, int ld, int n_feat, const uint64_t* __restrict__ keys_s,
                               const int32_t* __restrict__ sid_scan, int64_t n, const int32_t* __restrict__ seg_start,
                               const int32_t* __restrict__ seg_first, const int32_t* __restrict__ rank_scan,
                               const int32_t* __restrict__ woff, int B, uint64_t key_cells, int gx, int gy, int max_voxels,
                               int max_pts,
                               float* __restrict__ feat, int ld_feat, int32_t* __restrict__ coords,
                               int32_t* __restrict__ num_points, uint64_t* __restrict__ ukeys,
                               int32_t* __restrict__ uperm, const int32_t* __restrict__ counts) {
    int64_t sid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int S = sid_scan[n - 1];
    int n_valid = counts[2];
    if (sid >= S) return;
    int start = seg_start[sid];
    int end = (sid + 1 < S) ? seg_start[sid + 1] : n_valid;
    uint64_t lin = keys_s[start] >> VOX_IDX_BITS;   // b * cells + cell
    ukeys[sid] = lin;
    const int b = (int)(lin / key_cells);
    lin -= (uint64_t)b * key_cells;
    const int local = rank_scan[seg_first[sid]] - 1 - woff[b];  // first-come order inside the window
    const bool kept = local < max_voxels;
    const int vid = woff[B + 1 + b] + local;
    uperm[sid] = kept ? vid : -1;
    if (kept) {
        int x = (int)(lin % (uint64_t)gx);
        int y = (int)((lin / (uint64_t)gx) % (uint64_t)gy);
        int z = (int)(lin / ((uint64_t)gx * (uint64_t)gy));
        *(int4*)(coords + (int64_t)vid * 4) = make_int4(b, z, y, x);
        int cnt = end - start;
        int m = cnt < max_pts ? cnt : max_pts;
        num_points[vid] = m;
        if (!FEATS) return;   // coordinates only: k_vox_features averages the point features later (phase 2)
        float acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = 0.f;
        for (int j = 0; j < m; ++j) {  // first max_pts points in point order (keys sort by point index inside a cell)
            int p = (int)(keys_s[start + j] & ((1ull << VOX_IDX_BITS) - 1));
            const float* pp = pts + (int64_t)p * ld;
#pragma unroll
            for (int f = 0; f < 8; ++f)
                if (f < n_feat) acc[f] += pp[f];
        }
        float norm = (float)(m < 1 ? 1 : m);
        float* fo = feat + (int64_t)vid * ld_feat;
#pragma unroll
        for (int f = 0; f < 8; ++f)  // (compile-time indices: acc[] stays in registers)
            if (f < ld_feat) fo[f] = f < n_feat ? acc[f] / norm : 0.f;
        for (int f = 8; f < ld_feat; ++f) fo[f] = 0.f;
    }
}

// Phase 2 of the voxeliser (insmos_voxelize_windows_phased): the mean of the first max_pts points' features per KEPT voxel, from
// the sorted keys / segment starts phase 1 left in the workspace.  Same loop, same order as k_vox_segments<true>: same bits.
__global__ void k_vox_features(const float* __restrict__ pts, int ld, int n_feat, const uint64_t* __restrict__ keys_s,
                               const int32_t* __restrict__ seg_start, const int32_t* __restrict__ kept_state,
                               const int32_t* __restrict__ uperm, const int32_t* __restrict__ num_points,
                               float* __restrict__ feat, int ld_feat) {
    const int64_t sid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= kept_state[0]) return;   // [0] = occupied cells
    const int vid = uperm[sid];
    if (vid < 0) return;
    const int start = seg_start[sid];
    const int m = num_points[vid];
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = 0.f;
    for (int j = 0; j < m; ++j) {
        const int p = (int)(keys_s[start + j] & ((1ull << VOX_IDX_BITS) - 1));
        const float* pp = pts + (int64_t)p * ld;
#pragma unroll
        for (int f = 0; f < 8; ++f)
            if (f < n_feat) acc[f] += pp[f];
    }
    const float norm = (float)(m < 1 ? 1 : m);
    float* fo = feat + (int64_t)vid * ld_feat;
#pragma unroll
    for (int f = 0; f < 8; ++f)
        if (f < ld_feat) fo[f] = f < n_feat ? acc[f] / norm : 0.f;
    for (int f = 8; f < ld_feat; ++f) fo[f] = 0.f;
}

// pc_voxel_id: one thread per sorted point (the voxel of a point = uperm of its segment; no per-voxel loops over 1..50 points)
__global__ void k_vox_pcid(const uint64_t* __restrict__ keys_s, const int32_t* __restrict__ sid_scan, int64_t n,
                           const int32_t* __restrict__ uperm, int64_t* __restrict__ pcid) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t k = keys_s[j];
    if (k == INSMOS_INVALID_KEY) return;  // (pcid was preset to -1)
    pcid[(int64_t)(k & ((1ull << VOX_IDX_BITS) - 1))] = (int64_t)uperm[sid_scan[j] - 1];
}

// ---------------------------------------------------------------------------------------------------
// strided SparseConv3d output coordinate set
// ---------------------------------------------------------------------------------------------------
struct DownParams { int ks[3], st[3], pd[3], oshape[3]; };   // (cells of a batch: window b owns bits [b*cells, (b+1)*cells))

// mark every output cell some tap of some active input reaches, in an occupancy bitmap of the output grid.  One thread per
// INPUT voxel: along each axis the outputs o with 0 <= i + pad - o*stride < ksize form a short interval (1-2 cells for the
// 3/2/1 maps), so the voxel walks its <= 8 reachable cells instead of testing all 27 taps
__global__ void k_down_mark(const int32_t* __restrict__ in_coords, int64_t n_in, DownParams P, uint32_t* __restrict__ bitmap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    int4 c = *(const int4*)(in_coords + i * 4);  // [b,z,y,x]
    const int in[3] = {c.y, c.z, c.w};
    int lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int top = in[d] + P.pd[d];                     // o*st <= top
        const int bot = top - P.ks[d] + 1;                   // o*st >= bot
        hi[d] = top >= 0 ? top / P.st[d] : -1;
        lo[d] = bot > 0 ? (bot + P.st[d] - 1) / P.st[d] : 0;
        if (hi[d] > P.oshape[d] - 1) hi[d] = P.oshape[d] - 1;
    }
    for (int oz = lo[0]; oz <= hi[0]; ++oz)
        for (int oy = lo[1]; oy <= hi[1]; ++oy)
            for (int ox = lo[2]; ox <= hi[2]; ++ox) {
                const uint64_t key = key3b_encode(c.x, oz, oy, ox, P.oshape[0], P.oshape[1], P.oshape[2]);
                atomicOr(&bitmap[key >> 5], 1u << (unsigned)(key & 31));
            }
}
__global__ void k_word_popc(const uint32_t* __restrict__ bitmap, int64_t nwords, int32_t* __restrict__ cnt) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) cnt[w] = __popc(bitmap[w]);
}
// ascending linear order falls out of the bitmap: row of a cell = (#set bits in lower words) + rank inside its word
__global__ void k_down_expand(const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ scan, int64_t nwords,
                              int D, int H, int W, int64_t cap, uint64_t* __restrict__ okeys, int32_t* __restrict__ ocoords,
                              int32_t* __restrict__ counts) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint32_t m = bitmap[w];
    int base = scan[w] - __popc(m);  // scan is inclusive
    while (m) {
        int b = __ffs(m) - 1;
        m &= m - 1;
        if (base < cap) {
            uint64_t k = (uint64_t)w * 32 + b;
            okeys[base] = k;
            const uint64_t zz = k / ((uint64_t)W * H);   // b * D + z
            int x = (int)(k % (uint64_t)W), y = (int)((k / (uint64_t)W) % (uint64_t)H), z = (int)(zz % (uint64_t)D);
            *(int4*)(ocoords + (int64_t)base * 4) = make_int4((int)(zz / (uint64_t)D), z, y, x);
        }
        ++base;
    }
    if (w == nwords - 1) counts[0] = scan[w];
}

// ---------------------------------------------------------------------------------------------------
// RANK MAPS (round 3): search-free 3D kernel maps.  A coordinate set of a (B, D, H, W) grid as an occupancy bitmap in
// 256-bit blocks (4 x u64) + the inclusive count of set bits up to each block: the sorted position of a cell is
//   incl[blk - 1] + popcount(the block's words before it) + popcount(its word below it)
// -- two independent loads per probe (32 bytes of bitmap, one prefix) where the binary search over the sorted keys makes ~18
// dependent ones.  The strided levels' bitmaps already exist (their coordinate sets are generated from them); level 1 gets
// one from the voxeliser's sorted cell keys.  Sorted position -> row: identity for the generated levels, `perm` for level 1
// (first-seen voxel order; -1 = dropped by the voxel cap).
// ---------------------------------------------------------------------------------------------------
__global__ void k_rank_mark_keys(const uint64_t* __restrict__ keys, int64_t n, unsigned long long* __restrict__ bits) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    atomicOr(&bits[k >> 6], 1ull << (k & 63));
}
// Rows of the generated levels come in ascending cell order, so the lanes of a wave aim at the same few bitmap words: a plain
// atomicOr per (voxel, reachable cell) makes ~3.4 same-address atomics per voxel, which serialise in the L2 (~12 ns each: the
// kernel took 50 us per level-1 call).  Per reachable-cell SLOT (2 x 2 x 2: low / high output along each axis) the wave ORs the
// bits of lanes that hit the same word into the LAST lane of each run of equal words (segmented suffix-free scan over lanes,
// 6 shuffle steps) and only that lane issues the atomic.  Unsorted rows (level 1: first-seen order) just form shorter runs.
__global__ void __launch_bounds__(256) k_down_mark64(const int32_t* __restrict__ in_coords, int64_t n_in, DownParams P,
                                                     unsigned long long* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_in;
    const int lane = threadIdx.x & 63;
    int4 c = make_int4(0, 0, 0, 0);
    if (live) c = *(const int4*)(in_coords + i * 4);  // [b,z,y,x]
    const int in[3] = {c.y, c.z, c.w};
    int lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {   // (same reachable-output intervals as k_down_mark)
        const int top = in[d] + P.pd[d];
        const int bot = top - P.ks[d] + 1;
        hi[d] = top >= 0 ? top / P.st[d] : -1;
        lo[d] = bot > 0 ? (bot + P.st[d] - 1) / P.st[d] : 0;
        if (hi[d] > P.oshape[d] - 1) hi[d] = P.oshape[d] - 1;
    }
    // the interval of an axis holds 0, 1 or 2 cells for the kernels in use (3/2/1 and 3x1x1/2x1x1/0); wider ones fall back
    const bool wide = live && (hi[0] - lo[0] > 1 || hi[1] - lo[1] > 1 || hi[2] - lo[2] > 1);
    if (__ballot(wide)) {   // (wave-uniform; never taken by the model's own maps)
        if (live)
            for (int oz = lo[0]; oz <= hi[0]; ++oz)
                for (int oy = lo[1]; oy <= hi[1]; ++oy)
                    for (int ox = lo[2]; ox <= hi[2]; ++ox) {
                        const uint64_t key = key3b_encode(c.x, oz, oy, ox, P.oshape[0], P.oshape[1], P.oshape[2]);
                        atomicOr(&bits[key >> 6], 1ull << (key & 63));
                    }
        return;
    }
#pragma unroll
    for (int slot = 0; slot < 8; ++slot) {
        const int sz = slot >> 2, sy = (slot >> 1) & 1, sx = slot & 1;
        const int oz = lo[0] + sz, oy = lo[1] + sy, ox = lo[2] + sx;
        const bool ok = live && oz <= hi[0] && oy <= hi[1] && ox <= hi[2];
        uint64_t word = ~0ull, m = 0ull;
        if (ok) {
            const uint64_t key = key3b_encode(c.x, oz, oy, ox, P.oshape[0], P.oshape[1], P.oshape[2]);
            word = key >> 6;
            m = 1ull << (key & 63);
        }
        if (__ballot(ok) == 0ull) continue;
        // inclusive OR-scan over runs of equal `word` (a run's lanes are consecutive)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t wu = __shfl_up(word, d, 64);
            const uint64_t mu = __shfl_up(m, d, 64);
            if (lane >= d && wu == word) m |= mu;
        }
        const uint64_t wn = __shfl_down(word, 1, 64);
        if (ok && (lane == 63 || wn != word)) atomicOr(&bits[word], m);
    }
}
__global__ void k_blk_popc(const uint64_t* __restrict__ bits, int64_t nblk, int32_t* __restrict__ cnt) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const ulonglong4 w = *(const ulonglong4*)(bits + 4 * b);
    cnt[b] = __popcll(w.x) + __popcll(w.y) + __popcll(w.z) + __popcll(w.w);
}
// ascending (b, z, y, x) order falls out of the bitmap; one thread per 64-bit word (most words are empty: they leave at once; the
// 64-bit divisions that turn a cell index into coordinates are done once per word, the bits of a word walk on from there)
__global__ void k_down_expand64(const uint64_t* __restrict__ bits, const int32_t* __restrict__ incl, int64_t nwords, int D, int H,
                                int W, int64_t cap, uint64_t* __restrict__ okeys, int32_t* __restrict__ ocoords,
                                int32_t* __restrict__ counts) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const int64_t blk = w >> 2;
    if (w == nwords - 1) counts[0] = incl[blk];
    uint64_t m = bits[w];
    if (!m) return;
    const int j = (int)(w & 3);
    int base = blk ? incl[blk - 1] : 0;
    for (int t = 0; t < j; ++t) base += __popcll(bits[4 * blk + t]);
    const uint64_t k0 = (uint64_t)w * 64;
    const uint64_t zz0 = k0 / ((uint64_t)W * H);   // b * D + z of the word's first cell
    const uint32_t rem = (uint32_t)(k0 - zz0 * ((uint64_t)W * H));
    const int y0 = (int)(rem / (uint32_t)W), x0 = (int)(rem % (uint32_t)W);
    while (m) {
        const int b = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        if (base < cap) {
            int x = x0 + b, y = y0;
            uint32_t zz = (uint32_t)zz0;
            while (x >= W) { x -= W; ++y; }
            while (y >= H) { y -= H; ++zz; }
            okeys[base] = k0 + (uint64_t)b;
            *(int4*)(ocoords + (int64_t)base * 4) = make_int4((int)(zz / (uint32_t)D), (int)(zz % (uint32_t)D), y, x);
        }
        ++base;
    }
}

// One wave = one 16-row group x ALL taps (lane = (tap slot g, row j); K/4 steps): the group's 128-bit active-tap mask is
// accumulated from wave ballots in scalar registers and written once -- no atomics, no zero-fill of the mask array.
__global__ void __launch_bounds__(256) k_build_nbr_rank(const int32_t* __restrict__ out_coords, int64_t n_out,
                                                        const uint64_t* __restrict__ bits, const int32_t* __restrict__ incl,
                                                        const int32_t* __restrict__ perm, NbrParams P, int32_t* __restrict__ nbr,
                                                        uint32_t* __restrict__ mask16, int sparse) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grp * 16 >= n_out) return;   // (wave-uniform)
    const int64_t o = grp * 16 + j;
    const bool row_ok = o < n_out;
    const int4 c = *(const int4*)(out_coords + (row_ok ? o : n_out - 1) * 4);
    const int D = P.shape[0], H = P.shape[1], W = P.shape[2];
    uint32_t mk[4] = {0u, 0u, 0u, 0u};
    for (int kk = 0; kk < P.K; kk += 4) {
        const int k = kk + g;
        const bool k_ok = k < P.K;
        const int kc = k_ok ? k : P.K - 1;
        int q[4] = {c.x * P.mul[0] + P.delta[kc][0], c.y * P.mul[1] + P.delta[kc][1], c.z * P.mul[2] + P.delta[kc][2],
                    c.w * P.mul[3] + P.delta[kc][3]};
        bool ok = k_ok && row_ok;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int dv = P.dv[d];
            if (dv > 1) {
                if (q[d] % dv != 0) ok = false;   // exact divisibility (floor semantics irrelevant once divisible)
                q[d] = q[d] / dv;
            }
        }
        int32_t r = -1;
        if (ok) {
            const uint64_t key = key3b_encode(q[0], q[1], q[2], q[3], D, H, W);
            if (key != INSMOS_INVALID_KEY) {
                const int64_t w = (int64_t)(key >> 6), blk = w >> 2;
                const int jw = (int)(w & 3);
                const ulonglong4 qq = *(const ulonglong4*)(bits + 4 * blk);
                const uint64_t mine = jw == 0 ? qq.x : jw == 1 ? qq.y : jw == 2 ? qq.z : qq.w;
                if ((mine >> (key & 63)) & 1ull) {
                    int pos = blk ? incl[blk - 1] : 0;
                    pos += (jw > 0 ? __popcll(qq.x) : 0) + (jw > 1 ? __popcll(qq.y) : 0) + (jw > 2 ? __popcll(qq.z) : 0);
                    pos += __popcll(mine & ((1ull << (key & 63)) - 1ull));
                    r = perm ? perm[pos] : pos;
                }
            }
        }
        const unsigned long long bal = __ballot(r >= 0);
        // (sparse stores when asked for: the 16 entries of a tap none of the group's rows has are never read through the mask)
        if (k_ok && row_ok && (!sparse || ((bal >> (16 * g)) & 0xFFFFull))) nbr[(int64_t)k * n_out + o] = r;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kt = kk + t;   // (taps past K never have r >= 0)
            if ((bal >> (16 * t)) & 0xFFFFull) mk[(kt >> 5) & 3] |= 1u << (kt & 31);
        }
    }
    if (mask16 && lane == 0) *(uint4*)(mask16 + grp * 4) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
}

// k_build_nbr_rank for SEVERAL tables in one launch (round 5: the 13 kernel maps of the 3D branch were 13 launches of 5-55 us, the
// level-4 ones a few hundred blocks each): a block finds its job in a prefix table of block counts, then runs the body above --
// same entries, same masks (tests/test_gpu_coords.py).  Tap offsets are int8 (|delta| <= 127) so that 16 jobs and 4 offset sets
// fit the 4 KiB of kernel arguments.
struct RankJobD {
    const int32_t* out_coords;
    const uint64_t* bits;
    const int32_t* incl;
    const int32_t* perm;
    int32_t* nbr;
    uint32_t* mask16;
    int64_t n_out;
    int shape[3], mul[4], dv[4];
    int K, dset, blk_end, pad;
};
constexpr int kRankJobsMax = 16, kRankDsetsMax = 4, kRankDsetTaps = 32;
struct RankJobs {
    RankJobD job[kRankJobsMax];
    int8_t delta[kRankDsetsMax][kRankDsetTaps][4];
    int n_jobs, sparse;
};
__global__ void __launch_bounds__(256) k_build_nbr_rank_multi(RankJobs J) {
    int ji = 0;
    while (ji + 1 < J.n_jobs && (int)blockIdx.x >= J.job[ji].blk_end) ++ji;   // (block-uniform, <= 15 steps)
    const RankJobD& Q = J.job[ji];
    const int blk = (int)blockIdx.x - (ji ? J.job[ji - 1].blk_end : 0);
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int64_t grp = (int64_t)blk * 4 + (threadIdx.x >> 6);
    const int64_t n_out = Q.n_out;
    if (grp * 16 >= n_out) return;   // (wave-uniform)
    const int64_t o = grp * 16 + j;
    const bool row_ok = o < n_out;
    const int4 c = *(const int4*)(Q.out_coords + (row_ok ? o : n_out - 1) * 4);
    const int D = Q.shape[0], H = Q.shape[1], W = Q.shape[2], K = Q.K;
    const uint64_t* __restrict__ bits = Q.bits;
    const int32_t* __restrict__ incl = Q.incl;
    const int32_t* __restrict__ perm = Q.perm;
    int32_t* __restrict__ nbr = Q.nbr;
    uint32_t mk[4] = {0u, 0u, 0u, 0u};
    for (int kk = 0; kk < K; kk += 4) {
        const int k = kk + g;
        const bool k_ok = k < K;
        const int kc = k_ok ? k : K - 1;
        const int8_t* dl = J.delta[Q.dset][kc];
        int q[4] = {c.x * Q.mul[0] + dl[0], c.y * Q.mul[1] + dl[1], c.z * Q.mul[2] + dl[2], c.w * Q.mul[3] + dl[3]};
        bool ok = k_ok && row_ok;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int dv = Q.dv[d];
            if (dv > 1) {
                if (q[d] % dv != 0) ok = false;
                q[d] = q[d] / dv;
            }
        }
        int32_t r = -1;
        if (ok) {
            const uint64_t key = key3b_encode(q[0], q[1], q[2], q[3], D, H, W);
            if (key != INSMOS_INVALID_KEY) {
                const int64_t w = (int64_t)(key >> 6), bq = w >> 2;
                const int jw = (int)(w & 3);
                const ulonglong4 qq = *(const ulonglong4*)(bits + 4 * bq);
                const uint64_t mine = jw == 0 ? qq.x : jw == 1 ? qq.y : jw == 2 ? qq.z : qq.w;
                if ((mine >> (key & 63)) & 1ull) {
                    int pos = bq ? incl[bq - 1] : 0;
                    pos += (jw > 0 ? __popcll(qq.x) : 0) + (jw > 1 ? __popcll(qq.y) : 0) + (jw > 2 ? __popcll(qq.z) : 0);
                    pos += __popcll(mine & ((1ull << (key & 63)) - 1ull));
                    r = perm ? perm[pos] : pos;
                }
            }
        }
        const unsigned long long bal = __ballot(r >= 0);
        if (k_ok && row_ok && (!J.sparse || ((bal >> (16 * g)) & 0xFFFFull))) nbr[(int64_t)k * n_out + o] = r;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kt = kk + t;
            if ((bal >> (16 * t)) & 0xFFFFull) mk[(kt >> 5) & 3] |= 1u << (kt & 31);
        }
    }
    if (Q.mask16 && lane == 0) *(uint4*)(Q.mask16 + grp * 4) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
}

// ---------------------------------------------------------------------------------------------------
// per-voxel tap resolver: one thread = one fine voxel.  The <= 27 coarse neighbour blocks a voxel's taps
// can fall into are fetched ONCE (independent loads, entries cached in LDS as (child_start, child_mask)),
// then every tap of the (2*RANGE+1)^3 x NDT kernel is resolved with bit arithmetic only.
//   MODE 0: write the neighbour table column-by-column (+ active-tap masks by a 16-lane OR reduction)
//   MODE 2: the same table with SPARSE stores -- entries of (16-row group, tap) pairs that are not in the group's mask stay unwritten
//   MODE 1: constant-input convolution -- out[o] = relu(bias + sum_{valid taps} w[k]) -- for the first
//           MotionNet layer, whose input is 0.5 on every voxel (motionnet.py:29-32): no table, no gathers.
// Tap order = ME kernel-region order (x fastest, then y, z, t).
// ---------------------------------------------------------------------------------------------------
template <int RANGE, int NDT, int MODE>
__global__ void __launch_bounds__(256) k_resolve_taps(const int32_t* __restrict__ coords, int64_t n_f,
                                                      const int32_t* __restrict__ parent, int L,
                                                      const int32_t* __restrict__ cnbr, int64_t n_c,
                                                      const int32_t* __restrict__ child_start,
                                                      const uint32_t* __restrict__ child_mask,
                                                      int32_t* __restrict__ nbr, uint32_t* __restrict__ mask16,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ out, int ld_out, int relu, int64_t row0,
                                                      const uint32_t* __restrict__ cmask16) {
    // cmask16: the COARSE table's active-tap masks when that table was written with sparse stores (its entries outside a group's
    // mask are unwritten memory: they are "no neighbour" and must not be read); null = a fully written table
    constexpr int NB = RANGE == 1 ? 2 : 3;       // candidate coarse blocks per axis
    constexpr int E = NB * NB * NB * NDT;        // cached coarse entries per voxel
    constexpr int W1 = 2 * RANGE + 1;
    __shared__ uint32_t sl[E][256];  // child_start (24 bits) << 8 | child_mask (8 bits)
    const int tid = threadIdx.x;
    const int64_t o_raw = row0 + (int64_t)blockIdx.x * blockDim.x + tid;  // rows [row0, n_f); row0 is a multiple of 16
    const bool live = o_raw < n_f;
    const int64_t o = live ? o_raw : n_f - 1;
    const int4 c = *(const int4*)(coords + o * 4);
    const int px = (c.x >> L) & 1, py = (c.y >> L) & 1, pz = (c.z >> L) & 1;
    const int p = parent[o];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int ibx = e % NB, iby = (e / NB) % NB, ibz = (e / (NB * NB)) % NB, idt = e / (NB * NB * NB);
        const int bx = RANGE == 1 ? px - 1 + ibx : ibx - 1;
        const int by = RANGE == 1 ? py - 1 + iby : iby - 1;
        const int bz = RANGE == 1 ? pz - 1 + ibz : ibz - 1;
        const int dt = NDT == 3 ? idt - 1 : 0;
        const int ctap = (bx + 1) + 3 * (by + 1) + 9 * (bz + 1) + 27 * (dt + 1);
        int q = -1;
        if (!cmask16 || ((cmask16[(int64_t)(p >> 4) * 4 + (ctap >> 5)] >> (ctap & 31)) & 1u)) q = cnbr[(int64_t)ctap * n_c + p];
        uint32_t v = 0u;
        // (fewer than 2^24 fine rows: the mask word carries child_start above its low byte -- one gather, not two)
        if (q >= 0) v = n_f < (1ll << 24) ? child_mask[q] : (((uint32_t)child_start[q] << 8) | (child_mask[q] & 0xFFu));
        sl[e][tid] = v;
    }
    // (each thread reads back only its own column: no barrier needed)
    uint32_t mw[4] = {0u, 0u, 0u, 0u};
    float acc[8];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;  // bias is added last, like the generic kernel's epilogue
    }
    int k = 0;
#pragma unroll
    for (int idt = 0; idt < NDT; ++idt)
#pragma unroll
        for (int dz = -RANGE; dz <= RANGE; ++dz)
#pragma unroll
            for (int dy = -RANGE; dy <= RANGE; ++dy)
#pragma unroll
                for (int dx = -RANGE; dx <= RANGE; ++dx, ++k) {
                    const int vx = px + dx, vy = py + dy, vz = pz + dz;
                    const int ibx = RANGE == 1 ? (vx >> 1) - (px - 1) : (vx >> 1) + 1;
                    const int iby = RANGE == 1 ? (vy >> 1) - (py - 1) : (vy >> 1) + 1;
                    const int ibz = RANGE == 1 ? (vz >> 1) - (pz - 1) : (vz >> 1) + 1;
                    const int e = ((idt * NB + ibz) * NB + iby) * NB + ibx;
                    const unsigned oct = (vx & 1) | ((vy & 1) << 1) | ((vz & 1) << 2);
                    const uint32_t ent = sl[e][tid];
                    const bool hit = (ent >> oct) & 1u;
                    if (MODE == 0 || MODE == 2) {
                        const int32_t r = hit ? (int32_t)(ent >> 8) + __popc(ent & ((1u << oct) - 1u)) : -1;
                        if (MODE == 2) {
                            // sparse stores: a (16-row group, tap) none of whose rows has the tap is never read by the convolution
                            // kernels (they walk the group's active-tap mask), so its 16 entries are not written -- half of the table
                            const unsigned long long bal = __ballot(hit && live);
                            if (live && ((bal >> (tid & 48)) & 0xFFFFull)) nbr[(int64_t)k * n_f + o] = r;
                        } else if (live) {
                            nbr[(int64_t)k * n_f + o] = r;
                        }
                        if (hit && live) mw[k >> 5] |= 1u << (k & 31);
                    } else {
                        // a tap none of the wave's 64 (Morton-consecutive) voxels has is skipped wave-uniformly: its
                        // products are exact zeros, the sums do not change (LiDAR shells are thin: most |dz| = 2 taps)
                        if (__ballot(hit) != 0ull) {
                            const float sel = hit ? 1.0f : 0.0f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[i] = fmaf(sel, w[k * 8 + i], acc[i]);
                        }
                    }
                }
    static_assert(W1 * W1 * W1 * NDT <= 128, "tap count");
    if (MODE == 0 || MODE == 2) {
        if (mask16) {
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                uint32_t v = mw[wd];
                v |= __shfl_xor(v, 8); v |= __shfl_xor(v, 4); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 1);
                if ((tid & 15) == 0 && live) mask16[(o >> 4) * 4 + wd] = v;
            }
        }
    } else if (live) {
        float* op = out + o * ld_out;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = acc[i] + bias[i];
            op[i] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// The constant-input first layer (MODE 1 above) on an OCCUPANCY CUBE instead of per-tap table arithmetic.  The 5^3 taps of a voxel
// with parities (px, py, pz) inside its parent are cells [p, p + 4] per axis of the 6^3 fine cells covered by the parent's 3^3
// coarse neighbourhood; the 27 neighbours' 8-bit child masks are spread into six z-planes of 6 x 6 occupancy bits (static
// shifts), the plane of tap offset dz is picked by pz and shifted by 6 * py + px once -- after which tap (dz, dy, dx) is bit
// 6 * (dy + 2) + (dx + 2) of word T[dz + 2]: a compile-time bit.  The planes are built once per COARSE voxel (k_parent_cubes: all
// children of a parent see the same cube).  A wave-wide OR of the five words tells, in scalar registers,
// which taps any of the wave's 64 voxels has; the others are skipped without a vector instruction.  Same sums, same order
// (k ascending, fmaf(sel, w[k], acc), bias last): the same bits as k_resolve_taps<2, 1, 1>.
// ---------------------------------------------------------------------------------------------------
// Part 1, one thread per COARSE voxel: its 27 neighbours' child masks spread into the six occupancy planes (rows y' 0..2 in lo,
// 3..5 in hi: 18 bits each), 12 words per coarse voxel.  The cube is the same for all of a parent's children.
__global__ void __launch_bounds__(256) k_parent_cubes(const int32_t* __restrict__ cnbr, int64_t n_c,
                                                      const uint32_t* __restrict__ child_mask, uint32_t* __restrict__ cubes,
                                                      const uint32_t* __restrict__ cmask16) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_c) return;
    // (cmask16 as in k_resolve_taps: taps 27..53, the dt = 0 slab, are bits 27..31 of word 0 and 0..21 of word 1)
    uint32_t mw0 = ~0u, mw1 = ~0u;
    if (cmask16) { mw0 = cmask16[(p >> 4) * 4 + 0]; mw1 = cmask16[(p >> 4) * 4 + 1]; }
    uint32_t lo[6] = {0u, 0u, 0u, 0u, 0u, 0u}, hi[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int ibz = 0; ibz < 3; ++ibz)
#pragma unroll
        for (int iby = 0; iby < 3; ++iby)
#pragma unroll
            for (int ibx = 0; ibx < 3; ++ibx) {
                const int ctap = ibx + 3 * iby + 9 * ibz + 27;   // (dt = 0 slab of the coarse 81-tap table)
                const bool written = ((ctap < 32 ? mw0 >> ctap : mw1 >> (ctap - 32)) & 1u) != 0u;
                const int q = written ? cnbr[(int64_t)ctap * n_c + p] : -1;
                const uint32_t m = q >= 0 ? (child_mask[q] & 0xFFu) : 0u;   // bit (x | y << 1 | z << 2)
#pragma unroll
                for (int oz = 0; oz < 2; ++oz)
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        const uint32_t two = (m >> (oz * 4 + oy * 2)) & 3u;
                        const int zz = 2 * ibz + oz, yy = 2 * iby + oy;
                        if (yy < 3) lo[zz] |= two << (6 * yy + 2 * ibx);
                        else hi[zz] |= two << (6 * (yy - 3) + 2 * ibx);
                    }
            }
    uint4* cp = (uint4*)(cubes + p * 12);
    cp[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    cp[1] = make_uint4(lo[4], lo[5], hi[0], hi[1]);
    cp[2] = make_uint4(hi[2], hi[3], hi[4], hi[5]);
}
// Part 2, one thread per fine voxel: its parent's cube (48 contiguous bytes, shared by the siblings), planes picked by pz and
// shifted by 6 * py + px, taps as compile-time bits.
__global__ void __launch_bounds__(256) k_const_conv125(const int32_t* __restrict__ coords, int64_t n_f,
                                                       const int32_t* __restrict__ parent, int L,
                                                       const uint32_t* __restrict__ cubes, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int ld_out,
                                                       int relu) {
    const int64_t o_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = o_raw < n_f;
    const int64_t o = live ? o_raw : n_f - 1;
    const int4 c = *(const int4*)(coords + o * 4);
    const int px = (c.x >> L) & 1, py = (c.y >> L) & 1, pz = (c.z >> L) & 1;
    const uint4* cp = (const uint4*)(cubes + (int64_t)parent[o] * 12);
    const uint4 c0 = cp[0], c1 = cp[1], c2 = cp[2];
    const uint32_t lo[6] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y}, hi[6] = {c1.z, c1.w, c2.x, c2.y, c2.z, c2.w};
    uint32_t T[5], U[5];
    const int sh = 6 * py + px;
#pragma unroll
    for (int d = 0; d < 5; ++d) {   // tap offset dz = d - 2 reads plane pz + d
        const uint32_t l = pz ? lo[d + 1] : lo[d], h = pz ? hi[d + 1] : hi[d];
        const uint64_t plane = (uint64_t)l | ((uint64_t)h << 18);
        T[d] = (uint32_t)(plane >> sh);
        uint32_t u = T[d];
#pragma unroll
        for (int x = 1; x < 64; x <<= 1) u |= (uint32_t)__shfl_xor((int)u, x);
        U[d] = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
    }
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    // (one scalar branch per x-ROW of five taps, not per tap: the row's 40 weights arrive with one batch of scalar loads; a tap the
    //  wave does not have inside a live row adds exact zeros)
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            if ((U[d] >> (6 * dy)) & 31u) {   // (wave-uniform)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) {
                    const int k = (d * 5 + dy) * 5 + dx;
                    const float sel = ((T[d] >> (6 * dy + dx)) & 1u) ? 1.0f : 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(sel, w[k * 8 + i], acc[i]);
                }
            }
        }
    if (live) {
        float* op = out + o * ld_out;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = acc[i] + bias[i];
            op[i] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

// first row of each trailing time slice of a sorted 4D key array: starts[d] = first row with t >= t_last - d
__global__ void k_tslice_starts(const uint64_t* __restrict__ keys, int64_t n, int max_d, int32_t* __restrict__ starts) {
    const int d = threadIdx.x;
    if (d >= max_d) return;
    const uint64_t t_last = keys[n - 1] >> 48;
    const uint64_t want = t_last >= (uint64_t)d ? (t_last - (uint64_t)d) << 48 : 0ull;
    int64_t lo = 0, hi = n;  // lower_bound(keys, want)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    starts[d] = (int32_t)lo;
}

// batched form: the time field is t' = tq * B + b; starts[d] = first row with tq >= tq_last - d
__global__ void k_tslice_starts_b(const uint64_t* __restrict__ keys, int64_t n, int max_d, int B,
                                  int32_t* __restrict__ starts) {
    const int d = threadIdx.x;
    if (d >= max_d) return;
    const int tp_last = (int)(keys[n - 1] >> 48) - (int)INSMOS_KEY_BIAS;
    const int tq_last = tp_last >= 0 ? tp_last / B : -((-tp_last + B - 1) / B);   // floor division
    const long long want_tp = (long long)(tq_last - d) * B + (long long)INSMOS_KEY_BIAS;
    const uint64_t want = want_tp > 0 ? (uint64_t)want_tp << 48 : 0ull;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    starts[d] = (int32_t)lo;
}


// ---- the level-down chain (insmos_level_down4d_chain): the same three kernels with the row count read from DEVICE memory, so that
// levels 1..3 follow each other without a host round trip.  Grids are sized for the finest level (an upper bound of every count).
__global__ void k_head_flags_dn(const uint64_t* __restrict__ keys, int64_t n_cap, const int32_t* __restrict__ n_dev, int64_t n_host,
                                int shift_bits, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cap) return;
    const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
    int f = 0;
    if (i < n) {
        const uint64_t k = keys[i];
        if (k != INSMOS_INVALID_KEY) f = (i == 0) || ((k >> shift_bits) != (keys[i - 1] >> shift_bits));
    }
    flag[i] = f;   // (rows past the count flag 0: the scan over n_cap rows then ends on the level's count)
}
__global__ void k_level_down_scatter_dn(const uint64_t* __restrict__ keys, const int32_t* __restrict__ flag,
                                        const int32_t* __restrict__ scan, const int32_t* __restrict__ n_dev, int64_t n_host,
                                        int shift_bits, uint64_t* __restrict__ okeys, int32_t* __restrict__ ocoords,
                                        int32_t* __restrict__ parent, int32_t* __restrict__ child_start,
                                        uint32_t* __restrict__ child_mask, int32_t* __restrict__ count_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
    if (i >= n) return;
    const int vid = scan[i] - 1;
    if (flag[i]) {   // (the body of k_level_down_scatter; the chain is only used below 2^24 rows: child_start always rides in the mask)
        const uint64_t k = (keys[i] >> shift_bits) << shift_bits;
        okeys[vid] = k;
        int x, y, z, t;
        key4_decode(k, x, y, z, t);
        *(int4*)(ocoords + (int64_t)vid * 4) = make_int4(x, y, z, t);
        child_start[vid] = (int32_t)i;
        uint32_t m = 0;
        for (int j = 0; j < 8 && i + j < n; ++j) {
            const uint64_t kj = keys[i + j];
            if ((kj >> shift_bits) != (k >> shift_bits)) break;
            m |= 1u << (unsigned)((kj >> (shift_bits - 3)) & 7ull);
        }
        child_mask[vid] = m | ((uint32_t)i << 8);
    }
    parent[i] = vid;
    if (i == n - 1) *count_out = scan[i];
}
// k_tslice_starts_b with the row count on the device (null: n_host)
__global__ void k_tslice_starts_bd(const uint64_t* __restrict__ keys, const int32_t* __restrict__ n_dev, int64_t n_host, int max_d, int B,
                                   int32_t* __restrict__ starts) {
    const int d = threadIdx.x;
    if (d >= max_d) return;
    const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
    if (n <= 0) { starts[d] = 0; return; }
    const int tp_last = (int)(keys[n - 1] >> 48) - (int)INSMOS_KEY_BIAS;
    const int tq_last = tp_last >= 0 ? tp_last / B : -((-tp_last + B - 1) / B);   // floor division
    const long long want_tp = (long long)(tq_last - d) * B + (long long)INSMOS_KEY_BIAS;
    const uint64_t want = want_tp > 0 ? (uint64_t)want_tp << 48 : 0ull;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    starts[d] = (int32_t)lo;
}

}  // namespace insmos

using namespace insmos;
static const int TPB = 256;

// ===================================================================================================
extern "C" size_t insmos_quantize4d_ws_bytes(int64_t n) {
    size_t N = (size_t)n;
    size_t st = sort_pairs_u64_u32_temp(N), sc = scan_i32_temp(N), sk = sort_keys_u64_temp(N);
    if (sk > st) st = sk;
    return pad256(N * 8) * 2 + pad256(N * 4) * 6 + (st > sc ? st : sc) + 1024;
}

extern "C" int insmos_quantize4d_ex(const float* points, int64_t n, int ld_pts, const float* quant_host, uint64_t* keys,
                                    int32_t* coords, int32_t* inverse, int32_t* cur_index, int32_t* counts, void* ws,
                                    size_t ws_bytes, int compact_keys, void* stream) {
    if (n <= 0 || ld_pts < 5 || !points || !quant_host) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    size_t N = (size_t)n;
    uint64_t* k_in = b.take<uint64_t>(N);
    uint64_t* k_s = b.take<uint64_t>(N);
    uint32_t* i_in = b.take<uint32_t>(N);
    uint32_t* i_s = b.take<uint32_t>(N);
    int32_t* flag = b.take<int32_t>(N);
    int32_t* scan = b.take<int32_t>(N);
    int32_t* tflag = b.take<int32_t>(N);
    int32_t* tscan = b.take<int32_t>(N);
    size_t st = sort_pairs_u64_u32_temp(N), sc = scan_i32_temp(N);
    size_t tb = st > sc ? st : sc;
    char* tmp = b.take<char>(tb);
    if (!b.ok) return INSMOS_EWORKSPACE;
    HIP_TRY(hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), s));
    unsigned g = cdiv(n, TPB);
    {
        ProfScope ps(KK_QUANT_KEYS, s);
        if (compact_keys)
            INSMOS_LAUNCH(k_quant_keys<true>, dim3(g), dim3(TPB), 0, s, points, n, ld_pts, quant_host[0], quant_host[1],
                          quant_host[2], quant_host[3], k_in, i_in, tflag, counts);
        else
            INSMOS_LAUNCH(k_quant_keys<false>, dim3(g), dim3(TPB), 0, s, points, n, ld_pts, quant_host[0], quant_host[1],
                          quant_host[2], quant_host[3], k_in, i_in, tflag, counts);
    }
    // (invalid keys are all-ones: they sort last in either width as long as no valid compact key is 2^40 - 1, and a
    //  window with invalid keys is rejected by the caller anyway)
    int rc = sort_pairs_u64_u32(tmp, st, k_in, k_s, i_in, i_s, N, 0, compact_keys ? 40 : 64, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_QUANT_SCATTER, s);
        INSMOS_LAUNCH(k_head_flags, dim3(g), dim3(TPB), 0, s, k_s, n, 0, flag);
    }
    rc = inclusive_scan_i32(tmp, sc, flag, scan, N, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_QUANT_SCATTER, s);
        if (compact_keys)
            INSMOS_LAUNCH(k_quant_scatter<true>, dim3(g), dim3(TPB), 0, s, k_s, i_s, flag, scan, n, keys, coords, inverse,
                          counts);
        else
            INSMOS_LAUNCH(k_quant_scatter<false>, dim3(g), dim3(TPB), 0, s, k_s, i_s, flag, scan, n, keys, coords, inverse,
                          counts);
    }
    rc = inclusive_scan_i32(tmp, sc, tflag, tscan, N, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_QUANT_SCATTER, s);
        INSMOS_LAUNCH(k_compact_index, dim3(g), dim3(TPB), 0, s, tflag, tscan, n, cur_index, counts + 1);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

int insmos::make_win_pts(const float* const* pts_host, const int64_t* n_pts_host, int B, WinPts* out, int64_t* total) {
    if (!pts_host || !n_pts_host || B < 1 || B > INSMOS_MAX_BATCH) return INSMOS_EINVAL;
    memset(out, 0, sizeof(*out));
    out->B = B;
    int64_t acc = 0;
    for (int b = 0; b < B; ++b) {
        if (!pts_host[b] || n_pts_host[b] <= 0) return INSMOS_EINVAL;
        out->p[b] = pts_host[b];
        out->start[b] = acc;
        acc += n_pts_host[b];
    }
    for (int b = B; b <= INSMOS_MAX_BATCH; ++b) out->start[b] = acc;
    *total = acc;
    return INSMOS_OK;
}

// B windows in one coordinate set (k_quant_keys_b): pts_host[b] / n_pts_host[b] = device pointer and point count of window
// b (same leading dimension).  Outputs as insmos_quantize4d_ex over the concatenation of the windows' points (window-major
// point index); the t column of `coords` and the time field of `keys` hold t' = floor(t / dt) * B + b.
// counts: [0] voxels, [1] current points, [2] outside the key window, [3] outside the compact-key box, [4 .. 4+B] start
// of each window's current points in `cur_index` ([4+B] = all of them): 5 + B int32 slots.
extern "C" int insmos_quantize4d_windows(const float* const* pts_host, const int64_t* n_pts_host, int B, int ld_pts,
                                         const float* quant_host, uint64_t* keys, int32_t* coords, int32_t* inverse,
                                         int32_t* cur_index, int32_t* counts, void* ws, size_t ws_bytes, int compact_keys,
                                         void* stream) {
    WinPts W;
    int64_t n = 0;
    int rc = make_win_pts(pts_host, n_pts_host, B, &W, &n);
    if (rc) return rc;
    if (n >= (1ll << 31) || ld_pts < 5 || !quant_host) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    size_t N = (size_t)n;
    uint64_t* k_in = b.take<uint64_t>(N);
    uint64_t* k_s = b.take<uint64_t>(N);
    uint32_t* i_in = b.take<uint32_t>(N);
    uint32_t* i_s = b.take<uint32_t>(N);
    int32_t* flag = b.take<int32_t>(N);
    int32_t* scan = b.take<int32_t>(N);
    int32_t* tflag = b.take<int32_t>(N);
    int32_t* tscan = b.take<int32_t>(N);
    size_t st = sort_pairs_u64_u32_temp(N), sc = scan_i32_temp(N), sk = sort_keys_u64_temp(N);
    size_t tb = st > sc ? st : sc;
    if (sk > tb) tb = sk;
    char* tmp = b.take<char>(tb);
    if (!b.ok) return INSMOS_EWORKSPACE;
    HIP_TRY(hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), s));
    unsigned g = cdiv(n, TPB);
    if (compact_keys == 2) {
        // packed keys: the point index rides in the high 24 bits above the 40-bit sort key (keys-only sort, 5 byte passes)
        if (n >= (1ll << PK_IDX_BITS) || 16 * B > 128) return INSMOS_EINVAL;
        {
            ProfScope ps(KK_QUANT_KEYS, s);
            INSMOS_LAUNCH(k_quant_keys_p, dim3(g), dim3(TPB), 0, s, W, n, ld_pts, quant_host[0], quant_host[1], quant_host[2],
                          quant_host[3], k_in, tflag, counts);
        }
        rc = sort_keys_u64(tmp, tb, k_in, k_s, N, 0, PK_KEY_BITS, s);   // stable: ties stay in point order
        if (rc) return rc;
        {
            ProfScope ps(KK_QUANT_SCATTER, s);
            INSMOS_LAUNCH(k_head_flags_p, dim3(g), dim3(TPB), 0, s, k_s, n, flag);
        }
        rc = inclusive_scan_i32(tmp, sc, flag, scan, N, s);
        if (rc) return rc;
        {
            ProfScope ps(KK_QUANT_SCATTER, s);
            INSMOS_LAUNCH(k_quant_scatter_p, dim3(g), dim3(TPB), 0, s, k_s, flag, scan, n, B, keys, coords, inverse, counts);
        }
    } else {
        {
            ProfScope ps(KK_QUANT_KEYS, s);
            if (compact_keys)
                INSMOS_LAUNCH(k_quant_keys_b<true>, dim3(g), dim3(TPB), 0, s, W, n, ld_pts, quant_host[0], quant_host[1],
                              quant_host[2], quant_host[3], k_in, i_in, tflag, counts);
            else
                INSMOS_LAUNCH(k_quant_keys_b<false>, dim3(g), dim3(TPB), 0, s, W, n, ld_pts, quant_host[0], quant_host[1],
                              quant_host[2], quant_host[3], k_in, i_in, tflag, counts);
        }
        const int end_bit = compact_keys ? 36 + bits_for((uint64_t)(16 * B - 1)) : 64;
        rc = sort_pairs_u64_u32(tmp, st, k_in, k_s, i_in, i_s, N, 0, end_bit, s);
        if (rc) return rc;
        {
            ProfScope ps(KK_QUANT_SCATTER, s);
            INSMOS_LAUNCH(k_head_flags, dim3(g), dim3(TPB), 0, s, k_s, n, 0, flag);
        }
        rc = inclusive_scan_i32(tmp, sc, flag, scan, N, s);
        if (rc) return rc;
        {
            ProfScope ps(KK_QUANT_SCATTER, s);
            if (compact_keys)
                INSMOS_LAUNCH(k_quant_scatter_b<true>, dim3(g), dim3(TPB), 0, s, k_s, i_s, flag, scan, n, B, keys, coords, inverse,
                              counts);
            else
                INSMOS_LAUNCH(k_quant_scatter_b<false>, dim3(g), dim3(TPB), 0, s, k_s, i_s, flag, scan, n, B, keys, coords, inverse,
                              counts);
        }
    }
    rc = inclusive_scan_i32(tmp, sc, tflag, tscan, N, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_QUANT_SCATTER, s);
        INSMOS_LAUNCH(k_compact_index, dim3(g), dim3(TPB), 0, s, tflag, tscan, n, cur_index, counts + 1);
        INSMOS_LAUNCH(k_cur_starts, dim3(1), dim3(64), 0, s, W, tscan, counts + 4);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_tslice_starts_batched(const uint64_t* keys, int64_t n, int max_d, int B, int32_t* starts,
                                            void* stream) {
    if (!keys || n <= 0 || max_d <= 0 || max_d > 64 || B < 1 || !starts) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    INSMOS_LAUNCH(k_tslice_starts_b, dim3(1), dim3(64), 0, s, keys, n, max_d, B, starts);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_quantize4d(const float* points, int64_t n, int ld_pts, const float* quant_host, uint64_t* keys,
                                 int32_t* coords, int32_t* inverse, int32_t* cur_index, int32_t* counts, void* ws,
                                 size_t ws_bytes, void* stream) {
    return insmos_quantize4d_ex(points, n, ld_pts, quant_host, keys, coords, inverse, cur_index, counts, ws, ws_bytes, 0,
                                stream);
}

extern "C" size_t insmos_level_down4d_ws_bytes(int64_t n) {
    return pad256((size_t)n * 4) * 2 + scan_i32_temp((size_t)n) + 1024;
}

extern "C" int insmos_level_down4d(const uint64_t* keys, int64_t n, int shift, uint64_t* out_keys, int32_t* out_coords,
                                   int32_t* parent, int32_t* child_start, uint32_t* child_mask, int32_t* counts,
                                   void* ws, size_t ws_bytes, void* stream) {
    if (n <= 0 || shift < 1 || shift > 15) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    int32_t* flag = b.take<int32_t>((size_t)n);
    int32_t* scan = b.take<int32_t>((size_t)n);
    size_t sc = scan_i32_temp((size_t)n);
    char* tmp = b.take<char>(sc);
    if (!b.ok) return INSMOS_EWORKSPACE;
    unsigned g = cdiv(n, TPB);
    {
        ProfScope ps(KK_LEVEL_DOWN, s);
        INSMOS_LAUNCH(k_head_flags, dim3(g), dim3(TPB), 0, s, keys, n, 3 * shift, flag);
    }
    int rc = inclusive_scan_i32(tmp, sc, flag, scan, (size_t)n, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_LEVEL_DOWN, s);
        INSMOS_LAUNCH(k_level_down_scatter, dim3(g), dim3(TPB), 0, s, keys, flag, scan, n, 3 * shift, out_keys,
                           out_coords, parent, child_start, child_mask, counts);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// Levels 1 .. n_levels of the MinkUNet coordinate hierarchy (minkunet.py:139-160: conv1p1s2, conv2p2s2, conv3p4s2 create them one
// from the other) in ONE chain of launches: level l's row count stays on the device (chain[l - 1]) and is what level l + 1's kernels
// read, so the host waits once -- for all counts and the time-slice starts -- instead of once per level (a single window's forward
// is a chain of such round trips: DESIGN.md section 6).  Arrays of level l (1-based; index l - 1 of the pointer lists) need room for
// n0 rows each (every level's count is <= n0).  chain (device, 4 + 16 * (n_levels + 1) int32): [l - 1] = rows of level l,
// [4 + 16 * l + d] = first row of level l (0 = the given one) whose scan index is >= last - d (insmos_tslice_starts_batched,
// max_d = 16).  n0 < 2^24 (EINVAL otherwise: the per-level call handles those).  Same outputs as n_levels calls of
// insmos_level_down4d + insmos_tslice_starts_batched on the rows below each count.
extern "C" int insmos_level_down4d_chain(const uint64_t* keys0, int64_t n0, int n_levels, int B, uint64_t* const* out_keys,
                                         int32_t* const* out_coords, int32_t* const* parent, int32_t* const* child_start,
                                         uint32_t* const* child_mask, int32_t* chain, void* ws, size_t ws_bytes, void* stream) {
    if (!keys0 || n0 <= 0 || n0 >= (1ll << 24) || n_levels < 1 || n_levels > 3 || B < 1 || !out_keys || !out_coords || !parent ||
        !child_start || !child_mask || !chain || !ws)
        return INSMOS_EINVAL;
    for (int l = 0; l < n_levels; ++l)
        if (!out_keys[l] || !out_coords[l] || !parent[l] || !child_start[l] || !child_mask[l]) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    int32_t* flag = b.take<int32_t>((size_t)n0);
    int32_t* scan = b.take<int32_t>((size_t)n0);
    const size_t sc = scan_i32_temp((size_t)n0);
    char* tmp = b.take<char>(sc);
    if (!b.ok) return INSMOS_EWORKSPACE;
    const unsigned g = cdiv(n0, TPB);
    const uint64_t* kin = keys0;
    for (int l = 1; l <= n_levels; ++l) {
        const int32_t* n_dev = l == 1 ? nullptr : chain + (l - 2);
        {
            ProfScope ps(KK_LEVEL_DOWN, s);
            INSMOS_LAUNCH(k_head_flags_dn, dim3(g), dim3(TPB), 0, s, kin, n0, n_dev, n0, 3 * l, flag);
        }
        int rc = inclusive_scan_i32(tmp, sc, flag, scan, (size_t)n0, s);
        if (rc) return rc;
        {
            ProfScope ps(KK_LEVEL_DOWN, s);
            INSMOS_LAUNCH(k_level_down_scatter_dn, dim3(g), dim3(TPB), 0, s, kin, flag, scan, n_dev, n0, 3 * l, out_keys[l - 1],
                          out_coords[l - 1], parent[l - 1], child_start[l - 1], child_mask[l - 1], chain + (l - 1));
            INSMOS_LAUNCH(k_tslice_starts_bd, dim3(1), dim3(64), 0, s, kin, n_dev, n0, 16, B, chain + 4 + 16 * (l - 1));
        }
        kin = out_keys[l - 1];
    }
    INSMOS_LAUNCH(k_tslice_starts_bd, dim3(1), dim3(64), 0, s, kin, (const int32_t*)(chain + (n_levels - 1)), n0, 16, B,
                  chain + 4 + 16 * n_levels);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_build_nbr(const int32_t* out_coords, int64_t n_out, const uint64_t* in_keys,
                                const int32_t* in_perm, int64_t n_in, int key_mode, const int32_t* in_shape_host,
                                const int32_t* delta_host, int K, const int32_t* mul_host, const int32_t* div_host,
                                int32_t* nbr, uint32_t* mask16, void* stream) {
    if (n_out <= 0 || K <= 0 || K > 128 || n_in < 0 || (key_mode != 0 && key_mode != 1)) return INSMOS_EINVAL;
    if (key_mode == 1 && !in_shape_host) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    NbrParams P;
    memset(&P, 0, sizeof(P));
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < 4; ++d) P.delta[k][d] = delta_host[k * 4 + d];
    for (int d = 0; d < 4; ++d) {
        P.mul[d] = mul_host ? mul_host[d] : 1;
        P.dv[d] = div_host ? div_host[d] : 1;
    }
    if (in_shape_host)
        for (int d = 0; d < 3; ++d) P.shape[d] = in_shape_host[d];
    P.K = K;
    dim3 grid(cdiv(n_out, TPB), (unsigned)K);
    ProfScope ps(KK_BUILD_NBR, s);
    if (mask16) HIP_TRY(hipMemsetAsync(mask16, 0, (size_t)((n_out + 15) / 16) * 4 * sizeof(uint32_t), s));
    // taps in x-triples (d, d+1, d+2 on x, everything else equal, x undivided): one search per triple
    bool x3 = key_mode == 1 && K % 3 == 0 && P.dv[3] == 1 && P.mul[3] >= 1;
    for (int k = 0; x3 && k < K; k += 3)
        for (int t = 1; t < 3; ++t)
            x3 = x3 && P.delta[k + t][0] == P.delta[k][0] && P.delta[k + t][1] == P.delta[k][1] &&
                 P.delta[k + t][2] == P.delta[k][2] && P.delta[k + t][3] == P.delta[k][3] + t;
    if (key_mode == 0)
        INSMOS_LAUNCH(k_build_nbr<0>, grid, dim3(TPB), 0, s, out_coords, n_out, in_keys, in_perm, n_in, P, nbr,
                           mask16);
    else if (x3)
        INSMOS_LAUNCH(k_build_nbr_x3, dim3(cdiv(n_out, TPB), (unsigned)(K / 3)), dim3(TPB), 0, s, out_coords, n_out, in_keys,
                      in_perm, n_in, P, nbr, mask16);
    else
        INSMOS_LAUNCH(k_build_nbr<1>, grid, dim3(TPB), 0, s, out_coords, n_out, in_keys, in_perm, n_in, P, nbr,
                           mask16);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" size_t insmos_voxelize_mean_ws_bytes(int64_t n) {
    size_t N = (size_t)n;
    size_t st = sort_keys_u64_temp(N), sc = scan_i32_temp(N), sr = sort_keys_u64_radix_temp(N);
    if (sr > st) st = sr;
    return pad256(N * 8) * 2 + pad256((N + 1) * 4) * 6 + (st > sc ? st : sc) + 2048;
}

// win_start (device, B + 1 int32, or null for one window): first point of each window in the window-major point array.
// Voxel rows are window-major: window b owns rows [counts[4+b], counts[4+b+1]), each window in its own first-seen order and
// capped at max_voxels on its own; coords column 0 = b; pc_voxel_id holds batch-wide rows; ukeys = b * key_cells + cell,
// key_cells = cells per window of the keys the level-1 tables are searched with (the spconv spatial shape is one cell
// deeper than the voxel grid, spconv_unet.py:114; 0 = the voxel grid's own cell count).
// counts: [0] voxel rows, [1] occupied cells, [2] in-range points, and when win_start is given [4 .. 4+B] row starts
// (5 + B int32 slots).
// phase 0: everything (insmos_voxelize_mean_windows).  phase 1: everything EXCEPT the feature means -- voxel coordinates,
// num_points, pc_voxel_id, ukeys / uperm, counts depend on the points' positions only, so the 3D coordinate sets and kernel maps
// can be built while the motion features are still being computed.  phase 2 (same arguments, same workspace, untouched in
// between): the feature means.  phase 1 + phase 2 == phase 0 bit for bit.
extern "C" int insmos_voxelize_windows_phased(const float* points, int64_t n, int ld_pts, int n_feat, const int32_t* win_start,
                                              int B, int64_t key_cells, const float* range_host, const float* vsize_host, int max_voxels,
                                              int max_pts, float* feat, int ld_feat, int32_t* coords, int32_t* num_points,
                                              int64_t* pc_voxel_id, uint64_t* ukeys, int32_t* uperm, int32_t* counts, void* ws,
                                              size_t ws_bytes, int phase, void* stream) {
    if (phase < 0 || phase > 2) return INSMOS_EINVAL;
    if (n <= 0 || n >= (1ll << VOX_IDX_BITS) || n_feat < 3 || n_feat > 8 || ld_pts < n_feat || ld_feat < n_feat ||
        max_voxels <= 0 || max_pts <= 0 || B < 1 || B > INSMOS_MAX_BATCH || (B > 1 && !win_start))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // grid = round((hi - lo) / vsize), as models.py:281-282 / spconv compute it (float64 on the host)
    int g3[3];
    for (int d = 0; d < 3; ++d)
        g3[d] = (int)llround(((double)range_host[3 + d] - (double)range_host[d]) / (double)vsize_host[d]);
    const uint64_t grid_cells = (uint64_t)g3[0] * g3[1] * g3[2];
    if (key_cells == 0) key_cells = (int64_t)grid_cells;
    if ((uint64_t)key_cells < grid_cells) return INSMOS_EINVAL;
    uint64_t max_lin = (uint64_t)key_cells * (uint64_t)B;
    int end_bit = VOX_IDX_BITS + bits_for(max_lin);
    if (end_bit > 63) return INSMOS_EINVAL;
    Bump b(ws, ws_bytes);
    size_t N = (size_t)n;
    uint64_t* k_in = b.take<uint64_t>(N);
    uint64_t* k_s = b.take<uint64_t>(N);
    int32_t* flag = b.take<int32_t>(N + 1);
    int32_t* sid_scan = b.take<int32_t>(N + 1);
    int32_t* mark = b.take<int32_t>(N + 1);
    int32_t* rank_scan = b.take<int32_t>(N + 1);
    int32_t* seg_start = b.take<int32_t>(N + 1);
    int32_t* seg_first = b.take<int32_t>(N + 1);
    int32_t* woff = b.take<int32_t>(2 * (INSMOS_MAX_BATCH + 1) + 4);   // (+ [.. + 0] = occupied cells, kept for phase 2)
    int32_t* kept_state = woff + 2 * (INSMOS_MAX_BATCH + 1);
    size_t st = sort_keys_u64_temp(N), sc = scan_i32_temp(N);
    {
        const size_t sr = sort_keys_u64_radix_temp(N);
        if (sr > st) st = sr;
    }
    char* tmp = b.take<char>(st > sc ? st : sc);
    if (!b.ok) return INSMOS_EWORKSPACE;
    unsigned g = cdiv(n, TPB);
    if (phase == 2) {
        ProfScope ps(KK_VOX_MEAN, s);
        INSMOS_LAUNCH(k_vox_features, dim3(g), dim3(TPB), 0, s, points, ld_pts, n_feat, k_s, seg_start, kept_state, uperm, num_points,
                      feat, ld_feat);
        HIP_TRY(hipGetLastError());
        return INSMOS_OK;
    }
    HIP_TRY(hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), s));
    {
        ProfScope ps(KK_VOX_KEYS, s);
        INSMOS_LAUNCH(k_vox_keys, dim3(g), dim3(TPB), 0, s, points, n, ld_pts, win_start, B, (uint64_t)key_cells,
                      range_host[0], range_host[1], range_host[2], vsize_host[0], vsize_host[1], vsize_host[2], g3[0], g3[1],
                      g3[2], k_in, pc_voxel_id, mark, counts);
    }
    // The keys come in point order with the point index in their low bits, so a STABLE sort of the cell field alone -- bits
    // [VOX_IDX_BITS, end_bit): 4 radix passes for a launch set of 8 windows instead of 64 bits' worth of merging -- gives the same
    // order as sorting whole keys.  Invalid keys are all-ones: their cell field, 2^bits - 1 >= max_lin, is above every valid cell
    // (<= max_lin - 1), so they still sort last.  INSMOS_VOX_SORT_FULL=1: the whole-key sort (A/B; same result).
    static const bool full_sort = [] { const char* e = getenv("INSMOS_VOX_SORT_FULL"); return e && e[0] == '1'; }();
    int rc = full_sort ? sort_keys_u64(tmp, st, k_in, k_s, N, 0, 64, s) : sort_keys_u64_radix(tmp, st, k_in, k_s, N, VOX_IDX_BITS, end_bit, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_VOX_SEGMENTS, s);
        INSMOS_LAUNCH(k_head_flags, dim3(g), dim3(TPB), 0, s, k_s, n, VOX_IDX_BITS, flag);
    }
    rc = inclusive_scan_i32(tmp, sc, flag, sid_scan, N, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_VOX_SEGMENTS, s);
        INSMOS_LAUNCH(k_vox_heads, dim3(g), dim3(TPB), 0, s, k_s, flag, sid_scan, n, seg_start, seg_first, mark, counts);
    }
    rc = inclusive_scan_i32(tmp, sc, mark, rank_scan, N, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_VOX_MEAN, s);
        INSMOS_LAUNCH(k_vox_offsets, dim3(1), dim3(64), 0, s, win_start, B, n, rank_scan, sid_scan,
                      max_voxels, woff, counts, kept_state);
        if (phase == 0)
            INSMOS_LAUNCH(k_vox_segments<true>, dim3(g), dim3(TPB), 0, s, points, ld_pts, n_feat, k_s, sid_scan, n, seg_start,
                          seg_first, rank_scan, woff, B, (uint64_t)key_cells, g3[0], g3[1], max_voxels, max_pts, feat, ld_feat, coords,
                          num_points, ukeys, uperm, counts);
        else
            INSMOS_LAUNCH(k_vox_segments<false>, dim3(g), dim3(TPB), 0, s, points, ld_pts, n_feat, k_s, sid_scan, n, seg_start,
                          seg_first, rank_scan, woff, B, (uint64_t)key_cells, g3[0], g3[1], max_voxels, max_pts, feat, ld_feat, coords,
                          num_points, ukeys, uperm, counts);
        INSMOS_LAUNCH(k_vox_pcid, dim3(g), dim3(TPB), 0, s, k_s, sid_scan, n, uperm, pc_voxel_id);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_voxelize_mean_windows(const float* points, int64_t n, int ld_pts, int n_feat, const int32_t* win_start,
                                            int B, int64_t key_cells, const float* range_host, const float* vsize_host, int max_voxels,
                                            int max_pts, float* feat, int ld_feat, int32_t* coords, int32_t* num_points,
                                            int64_t* pc_voxel_id, uint64_t* ukeys, int32_t* uperm, int32_t* counts, void* ws,
                                            size_t ws_bytes, void* stream) {
    return insmos_voxelize_windows_phased(points, n, ld_pts, n_feat, win_start, B, key_cells, range_host, vsize_host, max_voxels, max_pts,
                                          feat, ld_feat, coords, num_points, pc_voxel_id, ukeys, uperm, counts, ws, ws_bytes, 0, stream);
}

// one window (the reference's single VoxelGenerate call): counts = 4 int32 slots ([0] voxels, [1] cells, [2] points)
extern "C" int insmos_voxelize_mean(const float* points, int64_t n, int ld_pts, int n_feat, const float* range_host,
                                    const float* vsize_host, int max_voxels, int max_pts, float* feat, int ld_feat,
                                    int32_t* coords, int32_t* num_points, int64_t* pc_voxel_id, uint64_t* ukeys,
                                    int32_t* uperm, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    return insmos_voxelize_mean_windows(points, n, ld_pts, n_feat, nullptr, 1, 0, range_host, vsize_host, max_voxels, max_pts, feat,
                                        ld_feat, coords, num_points, pc_voxel_id, ukeys, uperm, counts, ws, ws_bytes, stream);
}

extern "C" size_t insmos_down_coords3d_ws_bytes_b(const int32_t* out_shape_host, int B) {
    size_t nwords = ((size_t)out_shape_host[0] * out_shape_host[1] * out_shape_host[2] * (size_t)(B < 1 ? 1 : B) + 31) / 32;
    return pad256(nwords * 4) * 3 + scan_i32_temp(nwords) + 1024;
}
extern "C" size_t insmos_down_coords3d_ws_bytes(const int32_t* out_shape_host) {
    return insmos_down_coords3d_ws_bytes_b(out_shape_host, 1);
}

// B windows stacked along the batch column of the indices (in_coords[:, 0] in [0, B)): one occupancy bitmap of
// B * cells bits, rows come out in ascending (b, z, y, x) order -- window-major, each window exactly as it comes out alone.
extern "C" int insmos_down_coords3d_b(const int32_t* in_coords, int64_t n_in, const int32_t* ksize_host,
                                      const int32_t* stride_host, const int32_t* pad_host, const int32_t* out_shape_host,
                                      int B, uint64_t* out_keys, int32_t* out_coords, int32_t* counts, void* ws,
                                      size_t ws_bytes, void* stream) {
    if (n_in <= 0 || B < 1 || B > INSMOS_MAX_BATCH) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    DownParams P;
    int K = 1;
    for (int d = 0; d < 3; ++d) {
        P.ks[d] = ksize_host[d]; P.st[d] = stride_host[d]; P.pd[d] = pad_host[d]; P.oshape[d] = out_shape_host[d];
        K *= ksize_host[d];
    }
    const int64_t cells = (int64_t)P.oshape[0] * P.oshape[1] * P.oshape[2] * B;
    if (cells <= 0 || cells >= (1ll << 36)) return INSMOS_EINVAL;
    const int64_t N = n_in * K;
    const int64_t cap = N < cells ? N : cells;
    const int64_t nwords = (cells + 31) / 32;
    Bump b(ws, ws_bytes);
    uint32_t* bitmap = b.take<uint32_t>((size_t)nwords);
    int32_t* cnt = b.take<int32_t>((size_t)nwords);
    int32_t* scan = b.take<int32_t>((size_t)nwords);
    size_t sc = scan_i32_temp((size_t)nwords);
    char* tmp = b.take<char>(sc);
    if (!b.ok) return INSMOS_EWORKSPACE;
    {
        ProfScope ps(KK_DOWN_CAND, s);
        HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)nwords * 4, s));
        INSMOS_LAUNCH(k_down_mark, dim3(cdiv(n_in, TPB)), dim3(TPB), 0, s, in_coords, n_in, P, bitmap);
        INSMOS_LAUNCH(k_word_popc, dim3(cdiv(nwords, TPB)), dim3(TPB), 0, s, bitmap, nwords, cnt);
    }
    int rc = inclusive_scan_i32(tmp, sc, cnt, scan, (size_t)nwords, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_DOWN_UNIQUE, s);
        INSMOS_LAUNCH(k_down_expand, dim3(cdiv(nwords, TPB)), dim3(TPB), 0, s, bitmap, scan, nwords, P.oshape[0], P.oshape[1],
                           P.oshape[2], cap, out_keys, out_coords, counts);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// ---- rank maps (k_build_nbr_rank) ----------------------------------------------------------------------------------------
static inline int64_t rank_words(int64_t cells) { return ((cells + 255) / 256) * 4; }
extern "C" size_t insmos_rankmap_words(const int32_t* shape_host, int B) {
    return (size_t)rank_words((int64_t)shape_host[0] * shape_host[1] * shape_host[2] * (B < 1 ? 1 : B));
}
extern "C" size_t insmos_rankmap_ws_bytes(const int32_t* shape_host, int B) {
    const size_t nblk = insmos_rankmap_words(shape_host, B) / 4;
    return pad256(nblk * 4) + scan_i32_temp(nblk) + 1024;
}
static int rank_finish(uint64_t* bits, int64_t nblk, int32_t* incl, void* ws, size_t ws_bytes, hipStream_t s) {
    Bump b(ws, ws_bytes);
    int32_t* cnt = b.take<int32_t>((size_t)nblk);
    const size_t sc = scan_i32_temp((size_t)nblk);
    char* tmp = b.take<char>(sc);
    if (!b.ok) return INSMOS_EWORKSPACE;
    INSMOS_LAUNCH(k_blk_popc, dim3(cdiv(nblk, TPB)), dim3(TPB), 0, s, bits, nblk, cnt);
    return inclusive_scan_i32(tmp, sc, cnt, incl, (size_t)nblk, s);
}
extern "C" int insmos_rankmap_from_keys(const uint64_t* keys, int64_t n, const int32_t* shape_host, int B, uint64_t* bits,
                                        int32_t* blk_incl, void* ws, size_t ws_bytes, void* stream) {
    if (!keys || n < 0 || !shape_host || B < 1 || !bits || !blk_incl || !ws) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nw = (int64_t)insmos_rankmap_words(shape_host, B);
    ProfScope ps(KK_BUILD_NBR, s);
    HIP_TRY(hipMemsetAsync(bits, 0, (size_t)nw * 8, s));
    if (n > 0) INSMOS_LAUNCH(k_rank_mark_keys, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, keys, n, (unsigned long long*)bits);
    int rc = rank_finish(bits, nw / 4, blk_incl, ws, ws_bytes, s);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// insmos_down_coords3d_b with the occupancy bitmap in CALLER memory, in rank-map form, valid afterwards: the output level's
// rank map for insmos_build_nbr_rank.  bits: insmos_rankmap_words(out_shape, B) u64, blk_incl: a quarter as many i32.
extern "C" int insmos_down_coords3d_rank(const int32_t* in_coords, int64_t n_in, const int32_t* ksize_host,
                                         const int32_t* stride_host, const int32_t* pad_host, const int32_t* out_shape_host,
                                         int B, uint64_t* out_keys, int32_t* out_coords, int32_t* counts, uint64_t* bits,
                                         int32_t* blk_incl, void* ws, size_t ws_bytes, void* stream) {
    if (n_in <= 0 || B < 1 || B > INSMOS_MAX_BATCH || !bits || !blk_incl) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    DownParams P;
    int K = 1;
    for (int d = 0; d < 3; ++d) {
        P.ks[d] = ksize_host[d]; P.st[d] = stride_host[d]; P.pd[d] = pad_host[d]; P.oshape[d] = out_shape_host[d];
        K *= ksize_host[d];
    }
    const int64_t cells = (int64_t)P.oshape[0] * P.oshape[1] * P.oshape[2] * B;
    if (cells <= 0 || cells >= (1ll << 36)) return INSMOS_EINVAL;
    const int64_t N = n_in * K;
    const int64_t cap = N < cells ? N : cells;
    const int64_t nw = rank_words(cells);
    {
        ProfScope ps(KK_DOWN_CAND, s);
        HIP_TRY(hipMemsetAsync(bits, 0, (size_t)nw * 8, s));
        INSMOS_LAUNCH(k_down_mark64, dim3(cdiv(n_in, TPB)), dim3(TPB), 0, s, in_coords, n_in, P, (unsigned long long*)bits);
    }
    int rc = rank_finish(bits, nw / 4, blk_incl, ws, ws_bytes, s);
    if (rc) return rc;
    {
        ProfScope ps(KK_DOWN_UNIQUE, s);
        INSMOS_LAUNCH(k_down_expand64, dim3(cdiv(nw, TPB)), dim3(TPB), 0, s, bits, blk_incl, nw, P.oshape[0], P.oshape[1], P.oshape[2],
                      cap, out_keys, out_coords, counts);
    }
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// insmos_build_nbr (key_mode 1) over a rank map instead of a sorted key array: the same table, the same masks.
static int build_nbr_rank_impl(const int32_t* out_coords, int64_t n_out, const uint64_t* bits, const int32_t* blk_incl,
                               const int32_t* in_perm, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                               const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16, bool sparse_stores,
                               void* stream);
extern "C" int insmos_build_nbr_rank(const int32_t* out_coords, int64_t n_out, const uint64_t* bits, const int32_t* blk_incl,
                                     const int32_t* in_perm, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                                     const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16,
                                     void* stream) {
    return build_nbr_rank_impl(out_coords, n_out, bits, blk_incl, in_perm, in_shape_host, delta_host, K, mul_host, div_host, nbr, mask16,
                               false, stream);
}
// the same with sparse stores (see insmos_nbr81_from_coarse_rows_sparse): mask16 required
extern "C" int insmos_build_nbr_rank_sparse(const int32_t* out_coords, int64_t n_out, const uint64_t* bits, const int32_t* blk_incl,
                                            const int32_t* in_perm, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                                            const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16,
                                            void* stream) {
    if (!mask16) return INSMOS_EINVAL;
    return build_nbr_rank_impl(out_coords, n_out, bits, blk_incl, in_perm, in_shape_host, delta_host, K, mul_host, div_host, nbr, mask16,
                               true, stream);
}
static int build_nbr_rank_impl(const int32_t* out_coords, int64_t n_out, const uint64_t* bits, const int32_t* blk_incl,
                               const int32_t* in_perm, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                               const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16, bool sparse_stores,
                               void* stream) {
    if (n_out <= 0 || K <= 0 || K > 128 || !out_coords || !bits || !blk_incl || !in_shape_host || !delta_host || !nbr)
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    NbrParams P;
    memset(&P, 0, sizeof(P));
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < 4; ++d) P.delta[k][d] = delta_host[k * 4 + d];
    for (int d = 0; d < 4; ++d) {
        P.mul[d] = mul_host ? mul_host[d] : 1;
        P.dv[d] = div_host ? div_host[d] : 1;
    }
    for (int d = 0; d < 3; ++d) P.shape[d] = in_shape_host[d];
    P.K = K;
    ProfScope ps(KK_BUILD_NBR, s);
    INSMOS_LAUNCH(k_build_nbr_rank, dim3(cdiv((n_out + 15) / 16, 4)), dim3(256), 0, s, out_coords, n_out, bits, blk_incl, in_perm, P,
                  nbr, mask16, sparse_stores ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// Several kernel maps in ONE launch (k_build_nbr_rank_multi): what the native runner uses for the 13 maps of the 3D branch.
// Jobs with n_out == 0 are skipped; <= 16 jobs, K <= 32 taps each, <= 4 distinct offset sets, |offset| <= 127.
extern "C" int insmos_build_nbr_rank_multi(const InsmosRankJob* jobs_host, int n_jobs, int sparse_stores, void* stream) {
    if (!jobs_host || n_jobs < 0 || n_jobs > kRankJobsMax) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RankJobs J;
    memset(&J, 0, sizeof(J));
    int n_sets = 0, nj = 0;
    long blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const InsmosRankJob& h = jobs_host[i];
        if (h.n_out == 0) continue;
        if (h.n_out < 0 || h.K <= 0 || h.K > kRankDsetTaps || !h.out_coords || !h.bits || !h.blk_incl || !h.in_shape || !h.delta || !h.nbr ||
            (sparse_stores && !h.mask16))
            return INSMOS_EINVAL;
        int8_t d8[kRankDsetTaps][4];
        memset(d8, 0, sizeof(d8));
        for (int k = 0; k < h.K; ++k)
            for (int d = 0; d < 4; ++d) {
                const int v = h.delta[k * 4 + d];
                if (v < -127 || v > 127) return INSMOS_EINVAL;
                d8[k][d] = (int8_t)v;
            }
        int ds = -1;
        for (int q = 0; q < n_sets && ds < 0; ++q)
            if (memcmp(J.delta[q], d8, sizeof(d8)) == 0) ds = q;
        if (ds < 0) {
            if (n_sets == kRankDsetsMax) return INSMOS_EINVAL;
            memcpy(J.delta[n_sets], d8, sizeof(d8));
            ds = n_sets++;
        }
        RankJobD& Q = J.job[nj++];
        Q.out_coords = h.out_coords; Q.bits = h.bits; Q.incl = h.blk_incl; Q.perm = h.in_perm; Q.nbr = h.nbr; Q.mask16 = h.mask16;
        Q.n_out = h.n_out;
        for (int d = 0; d < 3; ++d) Q.shape[d] = h.in_shape[d];
        for (int d = 0; d < 4; ++d) {
            Q.mul[d] = h.mul ? h.mul[d] : 1;
            Q.dv[d] = h.div ? h.div[d] : 1;
        }
        Q.K = h.K;
        Q.dset = ds;
        blocks += (long)cdiv((h.n_out + 15) / 16, 4);
        if (blocks >= (1l << 31)) return INSMOS_EINVAL;
        Q.blk_end = (int)blocks;
    }
    if (nj == 0) return INSMOS_OK;
    J.n_jobs = nj;
    J.sparse = sparse_stores ? 1 : 0;
    ProfScope ps(KK_BUILD_NBR, s);
    INSMOS_LAUNCH(k_build_nbr_rank_multi, dim3((unsigned)blocks), dim3(256), 0, s, J);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// ---- row regrouping ------------------------------------------------------------------------------------------------------
// The convolution kernels walk, per 16-row group, the UNION of the taps its rows have: a (group, tap) slot costs a full 16-row
// MFMA pass whether one row or sixteen use it (~30-40 % of the executed passes multiply absent rows).
// Which rows share a group is free: a level's row order is private to the runner (features, tables and rank lookups all go
// through it).  Rows are therefore re-ordered inside blocks of NB consecutive rows (spatially close: the gathers keep their
// locality) by their submanifold TAP SIGNATURE -- bit k = the row has the k-th of its 27 neighbours -- so that a group's rows want
// the same taps (the window index leads the sort key: rows of different windows are never exchanged).  Measured on the S0 windows: executed (group, tap) slots of the submanifold maps -19...-24 %, of the strided
// maps -21...-29 %, of the (cheap) inverse maps +12...+23 %.
// Signatures from the level's rank-map bitmap (one thread per row), then one workgroup = one block: a bitonic sort of
// (window, signature, local row) in LDS, the permuted coordinates and new_of_old / old_of_new.  Deterministic (the local row breaks
// ties).
// 27-bit submanifold tap signature of every row (bit k = the k-th neighbour, (dz, dy, dx) raster order, exists)
__device__ __forceinline__ uint32_t tap_signature(const int4 c, const uint64_t* __restrict__ bits, int D, int H, int W) {
    uint32_t sig = 0u;
    int k = 0;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx, ++k) {
                const uint64_t q = key3b_encode(c.x, c.y + dz, c.z + dy, c.w + dx, D, H, W);
                if (q != INSMOS_INVALID_KEY && ((bits[q >> 6] >> (q & 63)) & 1ull)) sig |= 1u << k;
            }
    return sig;
}
__global__ void k_regroup_sig(const int32_t* __restrict__ coords, int64_t n, const uint64_t* __restrict__ bits, int D, int H, int W,
                              uint32_t* __restrict__ sig) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    sig[o] = tap_signature(*(const int4*)(coords + o * 4), bits, D, H, W);
}
// sort key of a row inside its block: window first (rows stay window-major), then the signature, then the coordinate PARITY class
// -- rows of equal signature come out grouped by parity, which is what decides a row's valid taps in the strided / inverse maps
// (tap k of an inverse 3^3 stride-2 map exists only where (coordinate + pad - k) is even: <= 8 of 27 taps per parity class); free
// for the submanifold maps.  PARITY_FIRST: parity above the signature -- the inverse map of the level becomes near-dense (3-4
// active taps per group instead of 9-14) at the price of ~11 % more active slots in its submanifold maps.
__device__ __forceinline__ uint64_t regroup_key(const int4 c, uint32_t sig, bool parity_first) {
    const uint64_t par = (uint64_t)(((c.y & 1) << 2) | ((c.z & 1) << 1) | (c.w & 1));
    const uint64_t body = parity_first ? (par << 27) | (uint64_t)sig : ((uint64_t)sig << 3) | par;
    return ((uint64_t)c.x << 30) | body;   // 4 + 30 bits
}
template <int NB>
__global__ void __launch_bounds__(1024) k_regroup_rows(const int32_t* __restrict__ coords, int64_t n, const uint32_t* __restrict__ sig,
                                                       int parity_first, int32_t* __restrict__ new_coords,
                                                       int32_t* __restrict__ new_of_old, int32_t* __restrict__ old_of_new) {
    __shared__ uint64_t key[NB];
    constexpr int NT = NB < 2048 ? NB / 2 : 1024;   // threads: one comparator each, or two (NB = 4096)
    const int64_t base = (int64_t)blockIdx.x * NB;
    for (int i = threadIdx.x; i < NB; i += NT) {
        const int64_t o = base + i;
        // (rows past the end sort last)
        key[i] = o < n ? (regroup_key(*(const int4*)(coords + o * 4), sig[o], parity_first != 0) << 16) | (uint64_t)i : ~0ull;
    }
    __syncthreads();
    for (int size = 2; size <= NB; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < NB / 2; t += NT) {
                const int lo = 2 * t - (t & (stride - 1));   // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < NB; i += NT) {
        const int64_t nw = base + i;
        if (nw >= n) break;
        const int64_t old = base + (int64_t)(key[i] & 0xFFFFull);
        new_of_old[old] = (int32_t)nw;
        if (old_of_new) old_of_new[nw] = (int32_t)old;
        *(int4*)(new_coords + nw * 4) = *(const int4*)(coords + old * 4);
    }
}

// The 4096-row block on 1024 threads with four keys per thread in REGISTERS (element e = q * 1024 + thread): of the 78
// compare-exchange steps of the bitonic network only the 18 whose partner sits in another wave (strides 64..512) go through LDS;
// strides 1..32 are lane exchanges (57 steps, no barrier), strides 1024 / 2048 stay inside the thread.  Keys are distinct (the
// local row is part of the key), so compare-exchange is min / max.  Same result as k_regroup_rows<4096>; 47 -> 42 us per
// launch on the S0 sets: lane exchanges of 64-bit keys are ds_bpermute pairs, i.e. the LDS pipe again (one block per CU).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}
__global__ void __launch_bounds__(1024) k_regroup_rows4096(const int32_t* __restrict__ coords, int64_t n, const uint32_t* __restrict__ sig,
                                                           int parity_first, int32_t* __restrict__ new_coords,
                                                           int32_t* __restrict__ new_of_old, int32_t* __restrict__ old_of_new) {
    __shared__ uint64_t key[4096];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 4096;
    uint64_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 1024 + t;
        const int64_t o = base + e;
        v[q] = o < n ? (regroup_key(*(const int4*)(coords + o * 4), sig[o], parity_first != 0) << 16) | (uint64_t)e : ~0ull;
    }
    for (int size = 2; size <= 4096; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 1024) {           // partner = another register of this thread
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sq = stride >> 10;   // 1 or 2
                    if (q & sq) continue;
                    const bool up = ((q * 1024 + t) & size) == 0;
                    // (static register indices: both candidate partners are named, the stride picks)
                    const int qp = q | sq;
                    uint64_t a = v[q], b = qp == 1 ? v[1] : qp == 2 ? v[2] : v[3];
                    const bool sw = (a > b) == up;
                    const uint64_t na = sw ? b : a, nb = sw ? a : b;
                    v[q] = na;
                    if (qp == 1) v[1] = nb; else if (qp == 2) v[2] = nb; else v[3] = nb;
                }
            } else if (stride < 64) {       // partner = another lane of this wave
                const bool lower = (t & stride) == 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool up = ((q * 1024 + t) & size) == 0;
                    const uint64_t p = shfl_xor_u64(v[q], stride);
                    const uint64_t mn = v[q] < p ? v[q] : p, mx = v[q] < p ? p : v[q];
                    v[q] = (lower == up) ? mn : mx;
                }
            } else {                        // partner = another wave: through LDS
#pragma unroll
                for (int q = 0; q < 4; ++q) key[q * 1024 + t] = v[q];
                __syncthreads();
                const bool lower = (t & stride) == 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = q * 1024 + t;
                    const bool up = (e & size) == 0;
                    const uint64_t p = key[e ^ stride];
                    const uint64_t mn = v[q] < p ? v[q] : p, mx = v[q] < p ? p : v[q];
                    v[q] = (lower == up) ? mn : mx;
                }
                __syncthreads();
            }
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t nw = base + q * 1024 + t;
        if (nw < n) {
            const int64_t old = base + (int64_t)(v[q] & 0xFFFFull);
            new_of_old[old] = (int32_t)nw;
            if (old_of_new) old_of_new[nw] = (int32_t)old;
            *(int4*)(new_coords + nw * 4) = *(const int4*)(coords + old * 4);
        }
    }
}

// coords (n, 4) int32 (b, z, y, x) of one level, bits = that level's rank-map bitmap (insmos_rankmap_from_keys /
// insmos_down_coords3d_rank), shape = its (D, H, W).  block_rows in {256, 1024, 4096}.  Writes new_coords (n, 4) = the rows in
// their new order, new_of_old (n): the new row of each old row, and (optional) old_of_new, its inverse.  Rows of different
// windows (coords column 0) never change their relative order: window-major rows stay window-major.
extern "C" size_t insmos_regroup_ws_bytes(int64_t n) {
    if (n <= 0) return 0;
    return pad256((size_t)n * 8) * 2 + sort_keys_u64_temp((size_t)n) + 1024;
}
extern "C" int insmos_regroup_rows3d(const int32_t* coords, int64_t n, const uint64_t* bits, const int32_t* shape_host, int block_rows,
                                     int32_t* new_coords, int32_t* new_of_old, int32_t* old_of_new, void* ws, size_t ws_bytes,
                                     void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!coords || !bits || !shape_host || !new_coords || !new_of_old || !ws || n >= (1ll << 31)) return INSMOS_EINVAL;
    const int parity_first = block_rows < 0 ? 1 : 0;   // (negative block size: parity class above the signature)
    if (block_rows < 0) block_rows = -block_rows;
    if (block_rows != 256 && block_rows != 1024 && block_rows != 4096) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    uint32_t* sig = b.take<uint32_t>((size_t)n);
    if (!b.ok) return INSMOS_EWORKSPACE;
    ProfScope ps(KK_BUILD_NBR, s);
    INSMOS_LAUNCH(k_regroup_sig, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, coords, n, bits, shape_host[0], shape_host[1], shape_host[2], sig);
    if (block_rows == 256)
        INSMOS_LAUNCH(k_regroup_rows<256>, dim3(cdiv(n, 256)), dim3(128), 0, s, coords, n, sig, parity_first, new_coords, new_of_old,
                      old_of_new);
    else if (block_rows == 1024)
        INSMOS_LAUNCH(k_regroup_rows<1024>, dim3(cdiv(n, 1024)), dim3(512), 0, s, coords, n, sig, parity_first, new_coords, new_of_old,
                      old_of_new);
    else
        INSMOS_LAUNCH(k_regroup_rows4096, dim3(cdiv(n, 4096)), dim3(1024), 0, s, coords, n, sig, parity_first, new_coords, new_of_old,
                      old_of_new);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// The same over WHOLE WINDOWS (block_rows = 1): one stable radix sort of (window, signature) keys -- packed above the row index like
// the 4D quantiser's keys -- instead of block-local sorts.  Longer runs of equal signatures: consecutive tiles walk the same tap
// list at the same time, so the wide layers' weight fragments are re-used out of L1/L2 by neighbouring waves.
__global__ void k_regroup_keys(const int32_t* __restrict__ coords, int64_t n, const uint64_t* __restrict__ bits, int D, int H, int W,
                               uint64_t* __restrict__ keys) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const int4 c = *(const int4*)(coords + o * 4);
    keys[o] = ((uint64_t)o << PK_KEY_BITS) | regroup_key(c, tap_signature(c, bits, D, H, W), false);
}
__global__ void k_regroup_scatter(const uint64_t* __restrict__ keys_s, const int32_t* __restrict__ coords, int64_t n,
                                  int32_t* __restrict__ new_coords, int32_t* __restrict__ new_of_old, int32_t* __restrict__ old_of_new) {
    const int64_t nw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nw >= n) return;
    const int64_t old = (int64_t)(keys_s[nw] >> PK_KEY_BITS);
    new_of_old[old] = (int32_t)nw;
    if (old_of_new) old_of_new[nw] = (int32_t)old;
    *(int4*)(new_coords + nw * 4) = *(const int4*)(coords + old * 4);
}
extern "C" int insmos_regroup_rows3d_global(const int32_t* coords, int64_t n, const uint64_t* bits, const int32_t* shape_host,
                                            int32_t* new_coords, int32_t* new_of_old, int32_t* old_of_new, void* ws, size_t ws_bytes,
                                            void* stream) {
    if (n <= 0) return INSMOS_OK;
    if (!coords || !bits || !shape_host || !new_coords || !new_of_old || !ws || n >= (1ll << PK_IDX_BITS)) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    uint64_t* k_in = b.take<uint64_t>((size_t)n);
    uint64_t* k_s = b.take<uint64_t>((size_t)n);
    const size_t st = sort_keys_u64_temp((size_t)n);
    char* tmp = b.take<char>(st);
    if (!b.ok) return INSMOS_EWORKSPACE;
    ProfScope ps(KK_BUILD_NBR, s);
    INSMOS_LAUNCH(k_regroup_keys, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, coords, n, bits, shape_host[0], shape_host[1], shape_host[2], k_in);
    int rc = sort_keys_u64(tmp, st, k_in, k_s, (size_t)n, 0, 34, s);   // (regroup_key: window, signature, parity); stable
    if (rc) return rc;
    INSMOS_LAUNCH(k_regroup_scatter, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, k_s, coords, n, new_coords, new_of_old, old_of_new);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// the level-1 (voxeliser) arrays under a row re-ordering: num_points moves with its row, the cell -> row map (uperm, n_cells
// entries, -1 = dropped cell) and the point -> row map (pc_voxel_id, n_points entries, -1 = no voxel) are renamed in place
__global__ void k_regroup_apply(const int32_t* __restrict__ new_of_old, int64_t n_rows, const int32_t* __restrict__ num_old,
                                int32_t* __restrict__ num_new, int32_t* __restrict__ uperm, int64_t n_cells,
                                int64_t* __restrict__ pcid, int64_t n_points) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) num_new[new_of_old[i]] = num_old[i];
    if (i < n_cells) {
        const int v = uperm[i];
        if (v >= 0) uperm[i] = new_of_old[v];
    }
    if (i < n_points) {
        const int64_t v = pcid[i];
        if (v >= 0) pcid[i] = (int64_t)new_of_old[v];
    }
}
extern "C" int insmos_regroup_apply_voxels(const int32_t* new_of_old, int64_t n_rows, const int32_t* num_points_old,
                                           int32_t* num_points_new, int32_t* uperm, int64_t n_cells, int64_t* pc_voxel_id,
                                           int64_t n_points, void* stream) {
    if (n_rows <= 0) return INSMOS_OK;
    if (!new_of_old || !num_points_old || !num_points_new || !uperm || !pc_voxel_id || n_cells < 0 || n_points < 0) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int64_t m = n_rows > n_cells ? n_rows : n_cells;
    if (n_points > m) m = n_points;
    ProfScope ps(KK_BUILD_NBR, s);
    INSMOS_LAUNCH(k_regroup_apply, dim3(cdiv(m, TPB)), dim3(TPB), 0, s, new_of_old, n_rows, num_points_old, num_points_new, uperm,
                  n_cells, pc_voxel_id, n_points);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_down_coords3d(const int32_t* in_coords, int64_t n_in, const int32_t* ksize_host,
                                    const int32_t* stride_host, const int32_t* pad_host,
                                    const int32_t* out_shape_host, uint64_t* out_keys, int32_t* out_coords,
                                    int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
    return insmos_down_coords3d_b(in_coords, n_in, ksize_host, stride_host, pad_host, out_shape_host, 1, out_keys, out_coords,
                                  counts, ws, ws_bytes, stream);
}

extern "C" int insmos_nbr_from_coarse(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                                      const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                                      const uint32_t* child_mask, const int32_t* delta_host, int K, int32_t* nbr,
                                      uint32_t* mask16, void* stream) {
    if (n_f <= 0 || n_c <= 0 || K <= 0 || K > 128 || !fine_coords || !parent || !coarse_nbr81 || !child_start ||
        !child_mask || !delta_host || !nbr || fine_shift < 0 || fine_shift > 14)
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // The 5^3 table of the stride-2^fine_shift level (x fastest, no time extent) -- what the TRAINING step needs for the first
    // layer's d/dW -- through the per-voxel resolver (k_resolve_taps: the 27 coarse neighbour blocks fetched once per voxel, every
    // tap by bit arithmetic) instead of one thread per (voxel, tap): the same entries and masks (tests/test_gpu_coords.py), the
    // 785 MB table of a four-window step in a third of the time.  INSMOS_NBR125_RESOLVER=0 keeps the generic kernel.
    static const bool fast125 = [] { const char* e = getenv("INSMOS_NBR125_RESOLVER"); return !(e && e[0] == '0'); }();
    // (n_f < 2^24: the resolver reads the first child row out of bits 8..31 of child_mask -- the packed form k_level_down_scatter writes
    //  only below that row count; larger sets take the generic kernel, which reads child_start itself and masks child_mask with 0xFF)
    if (fast125 && K == 125 && n_f < (1ll << 24)) {
        bool is5 = true;
        const int st = 1 << fine_shift;
        for (int k = 0; k < 125 && is5; ++k)
            is5 = delta_host[k * 4 + 0] == (k % 5 - 2) * st && delta_host[k * 4 + 1] == ((k / 5) % 5 - 2) * st &&
                  delta_host[k * 4 + 2] == (k / 25 - 2) * st && delta_host[k * 4 + 3] == 0;
        if (is5) {
            ProfScope ps(KK_BUILD_NBR, s);
            INSMOS_LAUNCH((k_resolve_taps<2, 1, 0>), dim3(cdiv(n_f, TPB)), dim3(TPB), 0, s, fine_coords, n_f, parent, fine_shift,
                          coarse_nbr81, n_c, child_start, child_mask, nbr, mask16, (const float*)nullptr, (const float*)nullptr,
                          (float*)nullptr, 0, 0, (int64_t)0, (const uint32_t*)nullptr);
            HIP_TRY(hipGetLastError());
            return INSMOS_OK;
        }
    }
    TapList T;
    memset(&T, 0, sizeof(T));
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < 4; ++d) T.delta[k][d] = delta_host[k * 4 + d];
    T.K = K;
    ProfScope ps(KK_BUILD_NBR, s);
    if (mask16) HIP_TRY(hipMemsetAsync(mask16, 0, (size_t)((n_f + 15) / 16) * 4 * sizeof(uint32_t), s));
    INSMOS_LAUNCH(k_nbr_from_coarse, dim3(cdiv(n_f, TPB), (unsigned)K), dim3(TPB), 0, s, fine_coords, n_f, parent,
                       fine_shift, coarse_nbr81, n_c, child_start, child_mask, T, nbr, mask16);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_nbr_down_up(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                                  int64_t n_c, const int32_t* child_start, const uint32_t* child_mask, int32_t* dn,
                                  uint32_t* dn_mask16, int32_t* up, uint32_t* up_mask16, void* stream) {
    if (n_f <= 0 || n_c <= 0 || !fine_coords || !parent || !child_start || !child_mask || !dn || !up) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_BUILD_NBR, s);
    INSMOS_LAUNCH(k_nbr_down, dim3(cdiv(n_c, TPB)), dim3(TPB), 0, s, n_c, child_start, child_mask, dn, dn_mask16);
    INSMOS_LAUNCH(k_nbr_up, dim3(cdiv(n_f, TPB)), dim3(TPB), 0, s, fine_coords, n_f, parent, fine_shift, up,
                       up_mask16);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

static int nbr81_rows_impl(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent, int fine_shift,
                           const int32_t* coarse_nbr81, const uint32_t* coarse_mask16, int64_t n_c, const int32_t* child_start,
                           const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16, bool sparse, void* stream);
extern "C" int insmos_nbr81_from_coarse_rows(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent,
                                             int fine_shift, const int32_t* coarse_nbr81, int64_t n_c,
                                             const int32_t* child_start, const uint32_t* child_mask, int32_t* nbr,
                                             uint32_t* mask16, void* stream) {
    return nbr81_rows_impl(fine_coords, n_f, row0, parent, fine_shift, coarse_nbr81, nullptr, n_c, child_start, child_mask, nbr, mask16,
                           false, stream);
}
// The same table with the entries of inactive (16-row group, tap) pairs left UNWRITTEN (mask16 is required and complete): for
// consumers that walk a group's taps through its mask only -- insmos_sparse_conv with 16-row tiles -- half the table's bytes.
extern "C" int insmos_nbr81_from_coarse_rows_sparse(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent,
                                                    int fine_shift, const int32_t* coarse_nbr81, int64_t n_c,
                                                    const int32_t* child_start, const uint32_t* child_mask, int32_t* nbr,
                                                    uint32_t* mask16, void* stream) {
    if (!mask16) return INSMOS_EINVAL;
    return nbr81_rows_impl(fine_coords, n_f, row0, parent, fine_shift, coarse_nbr81, nullptr, n_c, child_start, child_mask, nbr, mask16,
                           true, stream);
}
// the same from a coarse table that was itself written with sparse stores: coarse_mask16 = ITS mask array (entries outside a
// group's mask are unwritten memory and count as "no neighbour"); sparse_stores: write this table sparsely too (mask16 required)
extern "C" int insmos_nbr81_from_coarse_rows_masked(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent,
                                                    int fine_shift, const int32_t* coarse_nbr81, const uint32_t* coarse_mask16,
                                                    int64_t n_c, const int32_t* child_start, const uint32_t* child_mask,
                                                    int32_t* nbr, uint32_t* mask16, int sparse_stores, void* stream) {
    if (sparse_stores && !mask16) return INSMOS_EINVAL;
    return nbr81_rows_impl(fine_coords, n_f, row0, parent, fine_shift, coarse_nbr81, coarse_mask16, n_c, child_start, child_mask, nbr,
                           mask16, sparse_stores != 0, stream);
}
static int nbr81_rows_impl(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent, int fine_shift,
                           const int32_t* coarse_nbr81, const uint32_t* coarse_mask16, int64_t n_c, const int32_t* child_start,
                           const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16, bool sparse, void* stream) {
    if (n_f <= 0 || n_f >= (1 << 24) || n_c <= 0 || !fine_coords || !parent || !coarse_nbr81 || !child_start ||
        !child_mask || !nbr || fine_shift < 0 || fine_shift > 14 || row0 < 0)
        return INSMOS_EINVAL;
    row0 &= ~(int64_t)15;
    if (row0 >= n_f) return INSMOS_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_BUILD_NBR, s);
    if (sparse)
        INSMOS_LAUNCH((k_resolve_taps<1, 3, 2>), dim3(cdiv(n_f - row0, TPB)), dim3(TPB), 0, s, fine_coords, n_f, parent, fine_shift,
                      coarse_nbr81, n_c, child_start, child_mask, nbr, mask16, (const float*)nullptr, (const float*)nullptr,
                      (float*)nullptr, 0, 0, row0, coarse_mask16);
    else
    INSMOS_LAUNCH((k_resolve_taps<1, 3, 0>), dim3(cdiv(n_f - row0, TPB)), dim3(TPB), 0, s, fine_coords, n_f, parent,
                       fine_shift, coarse_nbr81, n_c, child_start, child_mask, nbr, mask16, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, 0, 0, row0, coarse_mask16);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_nbr81_from_coarse(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                                        const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                                        const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16, void* stream) {
    return insmos_nbr81_from_coarse_rows(fine_coords, n_f, 0, parent, fine_shift, coarse_nbr81, n_c, child_start, child_mask,
                                         nbr, mask16, stream);
}

extern "C" int insmos_const_conv125_from_coarse(const int32_t* fine_coords, int64_t n_f, const int32_t* parent,
                                                int fine_shift, const int32_t* coarse_nbr81, int64_t n_c,
                                                const int32_t* child_start, const uint32_t* child_mask,
                                                const float* w125x8, const float* bias8, float* out, int ld_out, int relu,
                                                void* stream) {
    return insmos_const_conv125_cubes(fine_coords, n_f, parent, fine_shift, coarse_nbr81, nullptr, n_c, child_start, child_mask, w125x8,
                                      bias8, out, ld_out, relu, nullptr, stream);
}
// the same with 48 bytes of scratch per COARSE voxel (cubes_ws, 16-byte aligned; null = the per-tap resolver): occupancy cubes
extern "C" int insmos_const_conv125_cubes(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                                          const int32_t* coarse_nbr81, const uint32_t* coarse_mask16, int64_t n_c,
                                          const int32_t* child_start, const uint32_t* child_mask, const float* w125x8,
                                          const float* bias8, float* out, int ld_out, int relu, void* cubes_ws, void* stream) {
    if (n_f <= 0 || n_c <= 0 || !fine_coords || !parent || !coarse_nbr81 || !child_start || !child_mask || !w125x8 ||
        !bias8 || !out || ld_out < 8 || fine_shift < 0 || fine_shift > 14 || ((uintptr_t)cubes_ws & 15))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = 125; ps.meta[1] = 1; ps.meta[2] = 8; ps.meta[3] = n_f;
    if (cubes_ws) {
        INSMOS_LAUNCH(k_parent_cubes, dim3(cdiv(n_c, TPB)), dim3(TPB), 0, s, coarse_nbr81, n_c, child_mask, (uint32_t*)cubes_ws,
                      coarse_mask16);
        INSMOS_LAUNCH(k_const_conv125, dim3(cdiv(n_f, TPB)), dim3(TPB), 0, s, fine_coords, n_f, parent, fine_shift,
                      (const uint32_t*)cubes_ws, w125x8, bias8, out, ld_out, relu);
    } else   // (the per-tap resolver: same bits)
        INSMOS_LAUNCH((k_resolve_taps<2, 1, 1>), dim3(cdiv(n_f, TPB)), dim3(TPB), 0, s, fine_coords, n_f, parent,
                           fine_shift, coarse_nbr81, n_c, child_start, child_mask, (int32_t*)nullptr, (uint32_t*)nullptr,
                           w125x8, bias8, out, ld_out, relu, (int64_t)0, coarse_mask16);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_tslice_starts(const uint64_t* keys, int64_t n, int max_d, int32_t* starts, void* stream) {
    if (!keys || n <= 0 || max_d <= 0 || max_d > 64 || !starts) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    INSMOS_LAUNCH(k_tslice_starts, dim3(1), dim3(64), 0, s, keys, n, max_d, starts);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
