// insmos_amd/csrc/spconv_tapc.hip -- TAP-COMPACTED sparse convolution for the 81-tap 4D layers of MotionNet (round 6).
//
// Why.  The output-stationary 16-row tiles of spconv.hip pay a full MFMA pass (and a 16-row gather, and a weight fragment) for
// every (16-row group, tap) slot ANY row of the group uses.  On the 4D levels a row has 16-21 of the 81 taps but a group of 16
// (t, Morton)-consecutive rows uses 31-47 of them: half of the issued passes multiply absent rows (oracle tables of an S0 window:
// issued / useful = 1.90 / 1.95 / 2.26 at levels 1 / 2 / 3).  Sorting rows by signature recovers 12-22 % of that; compacting the rows
// PER TAP recovers most of it: inside a block of 128 consecutive output rows, the rows that HAVE tap k are packed into dense groups
// of 16 -- issued / useful = 1.20 / 1.22 / 1.22 (tools/tapc_probe.py) -- and a tap's weight fragments are shared by its 2-3 dense
// groups instead of being re-fetched per 16-row group.
//
// How.  Two kernels:
//   * k_tapc_build (once per neighbour table and class count; a table serves 2-4 layers): one wave per (128-row block, tap class)
//     walks the class's taps in ascending order, reads the dense table's column pieces (two coalesced 256-byte loads), compacts the
//     present rows with a ballot + prefix popcount and writes ITEMS of 16 entries: entry = [31] one bit of the tap id (entry j < 7
//     carries bit j) | [30:23] the row's index inside its block (128 = padding: a scratch accumulator row) | [22:0] the neighbour's
//     row (all ones = none).  An item is one dense MFMA group of ONE tap.
//   * k_conv_tapc: one wave per (block, class) streams its items through a counted, branch-free software pipeline (entries three
//     items ahead, gathers + weight fragments two, MFMAs now).  The accumulators of the block's 128 rows cannot stay in MFMA
//     registers (column j of a dense group is a different row for every tap): they are PARKED IN LDS, row-major, and an item reads
//     its 16 rows' accumulators as the MFMA's C operand (lane (g, j): channels 4g..4g+3 of row j: one ds_read_b128 per channel tile),
//     runs the tap's chain on them and writes them back.  Per output element that is the same fmaf chain as in spconv.hip -- taps
//     ascending, chunks ascending, MFMA steps 0..3, the accumulator carried from tap to tap -- so the results are THE SAME BITS
//     (adding a tap the row does not have adds +-0 to a value that started at +0: skipping it changes nothing;
//     tests/test_gpu_conv.py::test_tap_compacted_kernel_is_bitwise_the_tile_kernels).
//   * classes: a layer spconv.hip runs on TAP-SPLIT tiles (four partial chains over the taps k % 4 == 0..3, summed ((c0 + c1) + c2) + c3)
//     runs here with four waves per block, one class each, each with its own LDS accumulators, summed in that order in the epilogue;
//     an unsplit layer (one chain over all taps) runs one wave per block on a table built with ONE class.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"

namespace insmos {
namespace {

constexpr int kRW = 128;                 // output rows per block
constexpr uint32_t kNoNbr = 0x7FFFFFu;   // entry bits [22:0]: no neighbour
constexpr int kItemsPerTap = kRW / 16;   // capacity: dense groups one tap can have in a block

struct TapcP {
    ConvP c;
    const uint32_t* tc;       // [block][class][items_cap][16] entries
    const int32_t* n_items;   // [block][class]
    uint32_t blk0;            // first block computed (row0 / 128)
    uint32_t items_cap;       // items per (block, class) region
    uint32_t row_lo;          // rows below it are not stored (row0 rounded down to 16, as the tile kernels do)
};

// ---------------------------------------------------------------------------------------------------------------------
// builder: NCLS waves of a 256-thread block = the NCLS classes of one row block (NCLS = 4) or four row blocks (NCLS = 1)
template <int NCLS>
__global__ void __launch_bounds__(256) k_tapc_build(const int32_t* __restrict__ nbr, const uint32_t* __restrict__ mask16, uint32_t n_out,
                                                    int K, uint32_t blk0, uint32_t n_blk, uint32_t items_cap, uint32_t* __restrict__ tc,
                                                    int32_t* __restrict__ n_items) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = blk0 + (NCLS == 4 ? blockIdx.x : blockIdx.x * 4u + w);
    const uint32_t cls = NCLS == 4 ? w : 0u;
    if (blk >= n_blk) return;   // (wave-uniform)
    const uint32_t r0 = blk * (uint32_t)kRW + (uint32_t)lane, r1 = r0 + 64u;
    uint32_t* out = tc + ((size_t)blk * NCLS + cls) * items_cap * 16u;
    uint32_t it_off = 0;
    // a SPARSE table (mask16: the active-tap bits of every 16-row group) holds unwritten memory in the entries of (group, tap) pairs
    // outside the group's mask: those are "no neighbour" and are not read
    uint4 mk0 = make_uint4(~0u, ~0u, ~0u, ~0u), mk1 = mk0;
    if (mask16) {
        if (r0 < n_out) mk0 = *(const uint4*)(mask16 + (size_t)(r0 >> 4) * 4);
        if (r1 < n_out) mk1 = *(const uint4*)(mask16 + (size_t)(r1 >> 4) * 4);
    }
    auto has = [](const uint4& m, int k) -> bool {
        const uint32_t w = (k >> 5) == 0 ? m.x : (k >> 5) == 1 ? m.y : (k >> 5) == 2 ? m.z : m.w;
        return ((w >> (k & 31)) & 1u) != 0;
    };
    for (int k = (int)cls; k < K; k += NCLS) {
        const int32_t e0 = (r0 < n_out && has(mk0, k)) ? nbr[(size_t)k * n_out + r0] : -1;
        const int32_t e1 = (r1 < n_out && has(mk1, k)) ? nbr[(size_t)k * n_out + r1] : -1;
        const bool p0 = e0 >= 0, p1 = e1 >= 0;
        const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
        const uint32_t c0 = (uint32_t)__popcll(b0), cnt = c0 + (uint32_t)__popcll(b1);
        if (cnt == 0) continue;   // (wave-uniform)
        const uint32_t below0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
        const uint32_t below1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
        const uint32_t ng = (cnt + 15u) >> 4;
        uint32_t* o = out + (size_t)it_off * 16u;
        auto tapbit = [&](uint32_t pos) -> uint32_t {
            const uint32_t j = pos & 15u;
            return j < 7u ? (((uint32_t)k >> j) & 1u) << 31 : 0u;
        };
        if (p0) o[below0] = (uint32_t)e0 | ((uint32_t)lane << 23) | tapbit(below0);
        if (p1) o[c0 + below1] = (uint32_t)e1 | ((uint32_t)(64 + lane) << 23) | tapbit(c0 + below1);
        const uint32_t pad = cnt + (uint32_t)lane;   // (at most 15 padding entries close the tap's last item)
        if (pad < ng * 16u) o[pad] = kNoNbr | ((uint32_t)kRW << 23) | tapbit(pad);
        it_off += ng;
    }
    // the consumer's pipeline is unrolled three items deep and has no tail: a list is closed with all-padding items (no neighbour,
    // scratch row) up to a multiple of three (the capacity has room: insmos_tapc_words)
    const uint32_t n3 = (it_off + 2u) / 3u * 3u;
    for (uint32_t e = it_off * 16u + (uint32_t)lane; e < n3 * 16u; e += 64u) out[e] = kNoNbr | ((uint32_t)kRW << 23);
    if (lane == 0) n_items[(size_t)blk * NCLS + cls] = (int32_t)n3;
}

// ---------------------------------------------------------------------------------------------------------------------
// NCH: 16-channel chunks per tap (CK == 0) -- 1, 2 or 3; CK == 8: the whole contraction of a tap is ONE 8-channel chunk
// COT: channel tiles (all of the layer's: 1 or 2); NCLS: tap classes = waves per block (1 or 4)
template <int NCH, int COT, int NCLS, int CK>
__global__ void __launch_bounds__(NCLS * 64) k_conv_tapc(TapcP Q) {
    static_assert(CK == 0 || (CK == 8 && NCH == 1), "single-chunk layers: Cin = 8");
    constexpr int PITCH = COT * 64 + 16;   // bytes between accumulator rows (16 B of padding: spreads the ds_read_b128 bank slots)
    constexpr int REGION = (kRW + 1) * PITCH;   // row kRW = the scratch row of padding entries
    __shared__ __attribute__((aligned(16))) unsigned char accs[NCLS][REGION];
    const ConvP& P = Q.c;
    const int lane = threadIdx.x & 63;
    const uint32_t cls = NCLS == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t blk = Q.blk0 + blockIdx.x;
    const uint32_t n_out = P.n_out;
    const uint32_t ld4 = (uint32_t)P.ld_in * 4u;
    const uint32_t cout = P.cout;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    constexpr uint32_t LW = CK == 8 ? 8u : 16u;    // bytes per lane per gather
    constexpr uint32_t LWF = CK == 8 ? 2u : 4u;    // floats per lane per weight fragment
    constexpr uint32_t FR = 64u * LWF;
    const uint32_t goff = (uint32_t)g * LW;
    const uint32_t blk_stride = (uint32_t)COT * FR;              // floats between chunk blocks of one tap
    const uint32_t tap_stride = (uint32_t)NCH * blk_stride;      // floats between taps
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    const size_t region = ((size_t)blk * NCLS + cls) * Q.items_cap;   // first item of this wave
    const int n_it = Q.n_items[(size_t)blk * NCLS + cls];
    const __amdgpu_buffer_rsrc_t rs_tc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Q.tc + region * 16u), 0, (int)(Q.items_cap * 64u), 0x00020000);

    unsigned char* const my = &accs[cls][0];
    // zero this wave's accumulators (the scratch row too)
    for (int o = lane * 16; o < REGION; o += 64 * 16) *(f32x4*)(my + o) = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint32_t woffv[COT];   // lanes whose output channel lies beyond Cout read zeros from past the end of the buffer (spconv.hip)
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co = (uint32_t)it * 16u + (uint32_t)(lane & 15);
        woffv[it] = co < cout ? ((uint32_t)it * FR + (uint32_t)lane * LWF) * 4u : 0x7FFFFFF0u;
    }

    if (n_it > 0) {
        const int last = n_it - 1;
        uint32_t pr[3];              // entries of the items in flight (lane (g, j): entry j)
        f32x4 bs[3][NCH];            // gathered B fragments
        f32x4 as[3][NCH][COT];       // weight A fragments
        uint32_t la[3];              // LDS byte address of the lane's accumulator piece (row's region + 16 g)
        const uint32_t jo = (uint32_t)j * 4u;
#define TAPC_REQP(slot, item)                                                                                               \
    {                                                                                                                       \
        const int it_ = (item) < last ? (item) : last;   /* running off the end re-requests the last item: clamp, no guard */ \
        pr[slot] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_tc, jo, (uint32_t)it_ * 64u, 0);                       \
    }
#define TAPC_REQAB(slot)                                                                                                    \
    {                                                                                                                       \
        const uint32_t p_ = pr[slot];                                                                                       \
        const uint32_t tap_ = (uint32_t)__builtin_amdgcn_ballot_w64((p_ >> 31) != 0u) & 0x7Fu;                              \
        const uint32_t idx_ = p_ & kNoNbr;                                                                                  \
        const uint32_t off_ = idx_ == kNoNbr ? 0x7FFFFFF0u : idx_ * ld4 + goff;   /* no neighbour: out of range, the load returns 0 */ \
        la[slot] = ((p_ >> 23) & 0xFFu) * (uint32_t)PITCH + (uint32_t)g * 16u;                                              \
        asm volatile("" : "+v"(la[slot]));   /* computed HERE: sunk to its use, the raw entry would outlive the reload of its slot */ \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                                   \
            if constexpr (CK == 8) {                                                                                        \
                const f32x2 t_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, off_, 0, 0));        \
                bs[slot][c] = (f32x4){t_[0], t_[1], 0.f, 0.f};                                                              \
            } else {                                                                                                        \
                bs[slot][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off_, (uint32_t)c * 64u, 0)); \
            }                                                                                                               \
        }                                                                                                                   \
        const uint32_t sw_ = tap_ * tap_stride * 4u;                                                                        \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c)                                                                     \
            _Pragma("unroll") for (int it = 0; it < COT; ++it) {                                                            \
                if constexpr (CK == 8) {                                                                                    \
                    const f32x2 t_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, woffv[it], sw_, 0)); \
                    as[slot][c][it] = (f32x4){t_[0], t_[1], 0.f, 0.f};                                                      \
                } else {                                                                                                    \
                    as[slot][c][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw_ + (uint32_t)c * blk_stride * 4u, 0)); \
                }                                                                                                           \
            }                                                                                                               \
    }
#define TAPC_MMA(slot)                                                                                                      \
    {                                                                                                                       \
        f32x4 acc_[COT];                                                                                                    \
        _Pragma("unroll") for (int it = 0; it < COT; ++it) acc_[it] = *(const f32x4*)(my + la[slot] + it * 64);            \
        constexpr int NS_ = CK == 8 ? 2 : 4;                                                                                \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c)                                                                     \
            _Pragma("unroll") for (int s = 0; s < NS_; ++s)                                                                 \
                _Pragma("unroll") for (int it = 0; it < COT; ++it)                                                          \
                    acc_[it] = MFMA(as[slot][c][it][s], bs[slot][c][s], acc_[it]);                                          \
        _Pragma("unroll") for (int it = 0; it < COT; ++it) *(f32x4*)(my + la[slot] + it * 64) = acc_[it];                   \
    }
        // prologue: entries of items 0..2, operands of items 0 and 1
        TAPC_REQP(0, 0)
        TAPC_REQP(1, 1)
        TAPC_REQP(2, 2)
        TAPC_REQAB(0)
        TAPC_REQP(0, 3)
        TAPC_REQAB(1)
        TAPC_REQP(1, 4)
        int i = 0;
        // (no tail: the builder closes every list with padding items up to a multiple of three)
        // (sched_barrier: without it hipcc hoists the next step's loads INTO this step's MFMAs, i.e. into registers of their own, and
        //  pays for that with a block of copies behind an s_waitcnt vmcnt(0) at the loop head -- the whole pipeline drained every
        //  three items; ISA checked: with the barriers every wait in the loop is a counted vmcnt)
        do {
            TAPC_REQAB(2) TAPC_REQP(2, i + 5) TAPC_MMA(0) __builtin_amdgcn_sched_barrier(0);
            TAPC_REQAB(0) TAPC_REQP(0, i + 6) TAPC_MMA(1) __builtin_amdgcn_sched_barrier(0);
            TAPC_REQAB(1) TAPC_REQP(1, i + 7) TAPC_MMA(2) __builtin_amdgcn_sched_barrier(0);
            i += 3;
        } while (i < n_it);
#undef TAPC_REQP
#undef TAPC_REQAB
#undef TAPC_MMA
    }
    __syncthreads();

    // ---- epilogue: thread -> (row, four channels); the classes' partial sums meet in the tap-split tiles' order
    constexpr int QPR = COT * 4;                   // channel quads per row
    constexpr int RPP = NCLS * 64 / QPR;           // rows per pass
    const uint32_t q = threadIdx.x % QPR, co0 = q * 4u;
#pragma unroll 1
    for (int r = (int)(threadIdx.x / QPR); r < kRW; r += RPP) {
        const uint32_t o = blk * (uint32_t)kRW + (uint32_t)r;
        if (o >= n_out || o < Q.row_lo || co0 >= cout) continue;
        f32x4 v = *(const f32x4*)(&accs[0][0] + r * PITCH + q * 16);
#pragma unroll
        for (int p = 1; p < NCLS; ++p) v += *(const f32x4*)(&accs[p][0] + r * PITCH + q * 16);
        v += *(const f32x4*)(P.bias + co0);
        if (P.relu_pre) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (P.res_mode == 1) {
            const float* rp = P.res + (size_t)o * P.ld_res + co0;
            if (P.vec_store && co0 + 3 < cout) {
                v += *(const f32x4*)rp;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co0 + e < cout) v[e] += rp[e];
            }
        } else if (P.res_mode == 2) {
            const float* rp = P.res + (size_t)o * P.ld_res + 2 * co0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co0 + e < cout) v[e] += rp[2 * e] + rp[2 * e + 1];
        }
        if (P.relu_post) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        float* op = P.out + (size_t)o * P.ld_out + co0;
        if (P.vec_store && co0 + 3 < cout) {
            *(f32x4*)op = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co0 + e < cout) op[e] = v[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Cin = 32 -> Cout 16 with rows that ARE 128-byte lines (block7.conv1).  Probe builds of k_conv_tapc (profiles/r06_tapc_probe.txt: no
// MFMAs / no gathers / no weight loads / accumulators in registers) all land within 15 % of the full kernel: it waits for the vector
// memory path -- per item 2 chunk gathers of 16 rows x 64 B (a line each: 27 ns per CU) and a weight fragment per chunk and channel
// tile -- with only the few waves per SIMD the accumulators' LDS leaves.  Two changes:
//   * WHOLE-ROW gathers (spconv_row32.hip): lane (r = lane >> 3, q = lane & 7) reads piece q of the neighbour row of the item's row r,
//     then of row 8 + r -- 8 lines per instruction; the neighbour indices reach those lanes by ds_bpermute from the lanes that hold
//     the item's entries; the rows become B fragments through a wave-private 2 KiB staging buffer (rows of 128 B, 16-byte pieces
//     XOR-swizzled by the row: conflict-free for both the writes and the fragment reads);
//   * WEIGHT FRAGMENTS ONCE PER TAP: the items of a tap are consecutive, so an item LOADS its tap's fragments only when the tap id of
//     the entry stream changes (a wave-uniform branch) and otherwise takes them over from the item before it (register moves).
// Pipeline per item i: MFMAs of i | staging of i + 1 | gathers (+ weights) of i + 3 | entries of i + 6.  The accumulator rows in LDS
// are 64 * COT bytes without padding, their 16-byte pieces XOR-swizzled by the row.  Same chain per output element: same bits.
// Measured per layer, interleaved in one process on a launch set of 8 (profiles/r06_tapc_layers_ab.csv): block7.0.conv1 245 (tiles)
// / 262 (k_conv_tapc) -> 185 us; with two channel tiles (Cout 32: block3.conv2, block6.conv2) the staging's extra LDS traffic and
// the smaller occupancy lose against k_conv_tapc (119 / 193 vs 106 / 172 us), so the launcher takes it for Cout <= 16 only.
template <int COT, int NCLS>
__global__ void __launch_bounds__(NCLS * 64) k_conv_tapc32(TapcP Q) {
    constexpr int AROW = COT * 64;
    constexpr int REGION = (kRW + 1) * AROW;
    __shared__ __attribute__((aligned(128))) unsigned char accs[NCLS][REGION];
    __shared__ __attribute__((aligned(128))) unsigned char stg[NCLS][16 * 128];
    const ConvP& P = Q.c;
    const int lane = threadIdx.x & 63;
    const uint32_t cls = NCLS == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t blk = Q.blk0 + blockIdx.x;
    const uint32_t n_out = P.n_out;
    const uint32_t cout = P.cout;
    auto aswz = [](uint32_t row, uint32_t piece) -> uint32_t {   // byte offset of 16-byte piece `piece` of accumulator row `row`
        return COT == 2 ? row * 128u + ((piece ^ (row & 7u)) << 4) : row * 64u + ((piece ^ ((row >> 2) & 3u)) << 4);
    };

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)P.in_bytes, 0x00020000);
    constexpr uint32_t FR = 256u;
    const uint32_t blk_stride = (uint32_t)COT * FR;
    const uint32_t tap_stride = 2u * blk_stride;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((uint32_t)P.K * tap_stride * 4u), 0x00020000);
    const size_t region = ((size_t)blk * NCLS + cls) * Q.items_cap;
    const __amdgpu_buffer_rsrc_t rs_tc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Q.tc + region * 16u), 0, (int)(Q.items_cap * 64u), 0x00020000);
    const uint32_t jo = (uint32_t)j * 4u;
    // the first entries do not depend on the item count: requested before anything else (a list's capacity is >= 3 items)
    uint32_t pr[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) pr[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_tc, jo, (uint32_t)q * 64u, 0);
    const int n_it = Q.n_items[(size_t)blk * NCLS + cls];

    unsigned char* const my = &accs[cls][0];
    unsigned char* const st = &stg[cls][0];
    for (int o = lane * 16; o < REGION; o += 64 * 16) *(f32x4*)(my + o) = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint32_t woffv[COT];
#pragma unroll
    for (int it = 0; it < COT; ++it) {
        const uint32_t co = (uint32_t)it * 16u + (uint32_t)(lane & 15);
        woffv[it] = co < cout ? ((uint32_t)it * FR + (uint32_t)lane * 4u) * 4u : 0x7FFFFFF0u;
    }
    // whole-row gather lanes: row r8 (then 8 + r8) of the item, piece q8; staging addresses
    const uint32_t r8 = (uint32_t)lane >> 3, q8 = (uint32_t)lane & 7u;
    const uint32_t st_wa = r8 * 128u + ((q8 ^ (r8 & 7u)) << 4), st_wb = (r8 + 8u) * 128u + ((q8 ^ (r8 & 7u)) << 4);   // ((r8 + 8) & 7 == r8 & 7)
    const uint32_t st_r0 = (uint32_t)j * 128u + ((((uint32_t)g) ^ ((uint32_t)j & 7u)) << 4);
    const uint32_t st_r1 = (uint32_t)j * 128u + ((((uint32_t)g + 4u) ^ ((uint32_t)j & 7u)) << 4);

    if (n_it > 0) {
        const int last = n_it - 1;
        f32x4 ga[3], gb[3];          // raw row pieces of the items in flight
        f32x4 fr[3][2];              // B fragments [slot][chunk]
        // weight fragments [slot][chunk][channel tile]: an item either LOADS its tap's fragments (first item of a tap) or COPIES them
        // from the item before it (register moves, no memory traffic).  (Rotating register SETS picked by a wave-uniform index were
        // the first build: hipcc turns that into one load plus conditional copies behind an s_waitcnt vmcnt(0).)
        f32x4 as[3][2][COT];
        uint32_t la[3];              // LDS byte offset of the lane's first accumulator piece
        uint32_t cur_tap = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int it = 0; it < COT; ++it) as[q][c][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define TAPC32_REQP(slot, item)                                                                                             \
    {                                                                                                                       \
        const int it_ = (item) < last ? (item) : last;                                                                      \
        pr[slot] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_tc, jo, (uint32_t)it_ * 64u, 0);                       \
    }
#define TAPC32_LOADW(slot)                                                                                                  \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                           \
        _Pragma("unroll") for (int it = 0; it < COT; ++it) {                                                                \
            as[slot][c][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woffv[it], sw_ + (uint32_t)c * blk_stride * 4u, 0)); \
        }
#define TAPC32_REQAB(slot)                                                                                                  \
    {                                                                                                                       \
        const uint32_t p_ = pr[slot];                                                                                       \
        const uint32_t tap_ = (uint32_t)__builtin_amdgcn_ballot_w64((p_ >> 31) != 0u) & 0x7Fu;                              \
        const uint32_t idx_ = p_ & kNoNbr;                                                                                  \
        const uint32_t ia_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r8 * 4u), (int)idx_);                             \
        const uint32_t ib_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r8 + 8u) * 4u), (int)idx_);                      \
        const uint32_t row_ = (p_ >> 23) & 0xFFu;                                                                           \
        la[slot] = aswz(row_, (uint32_t)g);                                                                                 \
        asm volatile("" : "+v"(la[slot]));                                                                                  \
        if (tap_ != cur_tap) {   /* wave-uniform: the first item of a tap brings the tap's weight fragments ... */           \
            cur_tap = tap_;                                                                                                 \
            const uint32_t sw_ = tap_ * tap_stride * 4u;                                                                    \
            TAPC32_LOADW(slot)                                                                                              \
        } else {                 /* ... its other items take them over from the item before */                              \
            _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                   \
                _Pragma("unroll") for (int it = 0; it < COT; ++it) as[slot][c][it] = as[(slot + 2) % 3][c][it];             \
        }                                                                                                                   \
        const uint32_t oa_ = ia_ == kNoNbr ? 0x7FFFFFF0u : ia_ * 128u + q8 * 16u;                                           \
        const uint32_t ob_ = ib_ == kNoNbr ? 0x7FFFFFF0u : ib_ * 128u + q8 * 16u;                                           \
        ga[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, oa_, 0, 0));                      \
        gb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ob_, 0, 0));                      \
    }
#define TAPC32_STAGE(slot)                                                                                                  \
    {                                                                                                                       \
        __builtin_amdgcn_wave_barrier();   /* (scheduling only: the reads below are other lanes' writes) */                 \
        *(f32x4*)(st + st_wa) = ga[slot];                                                                                   \
        *(f32x4*)(st + st_wb) = gb[slot];                                                                                   \
        __builtin_amdgcn_wave_barrier();                                                                                    \
        fr[slot][0] = *(const f32x4*)(st + st_r0);                                                                          \
        fr[slot][1] = *(const f32x4*)(st + st_r1);                                                                          \
        __builtin_amdgcn_wave_barrier();                                                                                    \
    }
#define TAPC32_MMA(slot)                                                                                                    \
    {                                                                                                                       \
        f32x4 acc_[COT];                                                                                                    \
        /* channel tile 1 = pieces 4 + g: with the XOR swizzle that is the lane's address with bit 6 flipped */            \
        _Pragma("unroll") for (int it = 0; it < COT; ++it) acc_[it] = *(const f32x4*)(my + (la[slot] ^ (uint32_t)(it * 64))); \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                       \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                   \
                _Pragma("unroll") for (int it = 0; it < COT; ++it)                                                          \
                    acc_[it] = MFMA(as[slot][c][it][s], fr[slot][c][s], acc_[it]);                                          \
        _Pragma("unroll") for (int it = 0; it < COT; ++it) *(f32x4*)(my + (la[slot] ^ (uint32_t)(it * 64))) = acc_[it];    \
    }
        // prologue (entries of items 0..2 are in flight)
        TAPC32_REQAB(0) TAPC32_REQP(0, 3)
        TAPC32_REQAB(1) TAPC32_REQP(1, 4)
        TAPC32_REQAB(2) TAPC32_REQP(2, 5)
        TAPC32_STAGE(0)
        int i = 0;
        do {   // (no tail: item lists are closed with padding items up to a multiple of three)
            TAPC32_MMA(0) TAPC32_STAGE(1) TAPC32_REQAB(0) TAPC32_REQP(0, i + 6) __builtin_amdgcn_sched_barrier(0);
            TAPC32_MMA(1) TAPC32_STAGE(2) TAPC32_REQAB(1) TAPC32_REQP(1, i + 7) __builtin_amdgcn_sched_barrier(0);
            TAPC32_MMA(2) TAPC32_STAGE(0) TAPC32_REQAB(2) TAPC32_REQP(2, i + 8) __builtin_amdgcn_sched_barrier(0);
            i += 3;
        } while (i < n_it);
#undef TAPC32_REQP
#undef TAPC32_LOADW
#undef TAPC32_REQAB
#undef TAPC32_STAGE
#undef TAPC32_MMA
    }
    __syncthreads();

    constexpr int QPR = COT * 4;
    constexpr int RPP = NCLS * 64 / QPR;
    const uint32_t q = threadIdx.x % QPR, co0 = q * 4u;
#pragma unroll 1
    for (int r = (int)(threadIdx.x / QPR); r < kRW; r += RPP) {
        const uint32_t o = blk * (uint32_t)kRW + (uint32_t)r;
        if (o >= n_out || o < Q.row_lo || co0 >= cout) continue;
        const uint32_t ao = aswz((uint32_t)r, q);
        f32x4 v = *(const f32x4*)(&accs[0][0] + ao);
#pragma unroll
        for (int p = 1; p < NCLS; ++p) v += *(const f32x4*)(&accs[p][0] + ao);
        v += *(const f32x4*)(P.bias + co0);
        if (P.relu_pre) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (P.res_mode == 1) {
            const float* rp = P.res + (size_t)o * P.ld_res + co0;
            if (P.vec_store && co0 + 3 < cout) {
                v += *(const f32x4*)rp;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co0 + e < cout) v[e] += rp[e];
            }
        } else if (P.res_mode == 2) {
            const float* rp = P.res + (size_t)o * P.ld_res + 2 * co0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co0 + e < cout) v[e] += rp[2 * e] + rp[2 * e + 1];
        }
        if (P.relu_post) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        float* op = P.out + (size_t)o * P.ld_out + co0;
        if (P.vec_store && co0 + 3 < cout) {
            *(f32x4*)op = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (co0 + e < cout) op[e] = v[e];
        }
    }
}

typedef void (*TapcKernel)(TapcP);
TapcKernel pick_tapc(int ck, int nch, int cot, int ncls) {
#define CASE(NCH_, COT_, NCLS_, CK_) if (nch == NCH_ && cot == COT_ && ncls == NCLS_ && ck == CK_) return k_conv_tapc<NCH_, COT_, NCLS_, CK_>;
    CASE(1, 1, 1, 0) CASE(1, 2, 1, 0) CASE(2, 1, 1, 0) CASE(2, 2, 1, 0) CASE(3, 1, 1, 0) CASE(3, 2, 1, 0)
    CASE(1, 1, 4, 0) CASE(1, 2, 4, 0) CASE(2, 1, 4, 0) CASE(2, 2, 4, 0) CASE(3, 1, 4, 0) CASE(3, 2, 4, 0)
    CASE(1, 1, 1, 8) CASE(1, 2, 1, 8) CASE(1, 1, 4, 8) CASE(1, 2, 4, 8)
#undef CASE
    return nullptr;
}
}  // namespace
}  // namespace insmos

using namespace insmos;

extern "C" size_t insmos_tapc_blocks(int64_t n_out) { return n_out <= 0 ? 0 : (size_t)((n_out + kRW - 1) / kRW); }

// items a (block, class) region holds: every tap of the class with all 128 rows, rounded up to the pipeline's multiple of three
static uint32_t tapc_items_cap(int K, int ncls) { return ((uint32_t)((K + ncls - 1) / ncls) * (uint32_t)kItemsPerTap + 2u) / 3u * 3u; }

// uint32 words of the item table of a (K, n_out) neighbour table with `ncls` tap classes (its counts: insmos_tapc_blocks * ncls int32)
extern "C" size_t insmos_tapc_words(int K, int64_t n_out, int ncls) {
    if (K <= 0 || K > 128 || n_out <= 0 || (ncls != 1 && ncls != 4)) return 0;
    return (size_t)insmos_tapc_blocks(n_out) * (size_t)ncls * tapc_items_cap(K, ncls) * 16u;
}

// row0: only the blocks from row0 / 128 on are built (the rest of the item table stays unwritten) -- for layers that run on a row
// suffix (insmos_sparse_conv_tapc_rows with that row0 or a later one)
extern "C" int insmos_tapc_build_masked(const int32_t* nbr, const uint32_t* mask16, int K, int64_t n_out, int64_t row0, int ncls,
                                        uint32_t* tc, int32_t* n_items, void* stream) {
    if (!nbr || !tc || !n_items || K <= 0 || K > 128 || n_out <= 0 || row0 < 0 || (ncls != 1 && ncls != 4) ||
        n_out * (int64_t)K * 4 >= (1ll << 31))
        return INSMOS_EINVAL;
    if (row0 >= n_out) return INSMOS_OK;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_blk = (uint32_t)insmos_tapc_blocks(n_out);
    const uint32_t blk0 = (uint32_t)(row0 / kRW);
    const uint32_t cap = tapc_items_cap(K, ncls);
    ProfScope ps(KK_BUILD_NBR, s);
    if (ncls == 4)
        INSMOS_LAUNCH(k_tapc_build<4>, dim3(n_blk - blk0), dim3(256), 0, s, nbr, mask16, (uint32_t)n_out, K, blk0, n_blk, cap, tc, n_items);
    else
        INSMOS_LAUNCH(k_tapc_build<1>, dim3((n_blk - blk0 + 3) / 4), dim3(256), 0, s, nbr, mask16, (uint32_t)n_out, K, blk0, n_blk, cap, tc, n_items);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

extern "C" int insmos_tapc_build(const int32_t* nbr, int K, int64_t n_out, int64_t row0, int ncls, uint32_t* tc, int32_t* n_items,
                                 void* stream) {
    return insmos_tapc_build_masked(nbr, nullptr, K, n_out, row0, ncls, tc, n_items, stream);
}

extern "C" int insmos_sparse_conv_tapc_rows(const float* in, int64_t n_in, int ld_in, int cin, const uint32_t* tc, const int32_t* n_items,
                                            int ncls, int K, int64_t n_out, int64_t row0, const float* wpacked, const float* bias,
                                            float* out, int ld_out, int cout, const float* res, int ld_res, int res_mode, int relu_pre,
                                            int relu_post, void* stream) {
    if (n_out <= 0 || row0 >= n_out) return INSMOS_OK;
    if (row0 < 0) return INSMOS_EINVAL;
    row0 &= ~(int64_t)15;
    if (!in || n_in <= 0 || n_in >= (int64_t)kNoNbr || !tc || !n_items || !wpacked || !bias || !out || cin <= 0 || ld_in % 4 != 0 ||
        ld_in < cin || K <= 0 || K > 128 || cout <= 0 || cout > 32 || ld_out < cout || (res_mode != 0 && !res) || ((uintptr_t)in & 15) ||
        n_in * (int64_t)ld_in * 4 >= (1ll << 31) || (ncls != 1 && ncls != 4))
        return INSMOS_EINVAL;
    // the class count IS the summation order: it must be the one the tile kernels use for this layer shape (insmos_conv_tap_classes)
    if (insmos_conv_tap_classes(K, cin, cout, 1) != ncls) return INSMOS_EINVAL;
    const int ck = cin == 8 ? 8 : 0;
    if (!ck && (cin % 16 != 0 || cin > 48)) return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    TapcP Q;
    ConvP& P = Q.c;
    P.in = in; P.nbr = nullptr; P.mask16 = nullptr; P.w = wpacked; P.w_rl = nullptr; P.bias = bias; P.out = out; P.res = res;
    P.n_out = (uint32_t)n_out;
    P.row0 = (uint32_t)row0;
    P.in_bytes = (uint32_t)((n_in - 1) * (int64_t)ld_in * 4 + (int64_t)cin * 4);
    P.ld_in = ld_in; P.cin = cin; P.K = K; P.ld_out = ld_out; P.cout = cout; P.ld_res = ld_res; P.res_mode = res_mode;
    P.relu_pre = relu_pre; P.relu_post = relu_post;
    P.n16 = cin / 16; P.has8 = ck ? 1 : 0; P.has4 = 0; P.nblk = ck ? 1 : cin / 16; P.ntile_co = (cout + 15) / 16;
    P.n_otiles = 0; P.tap_mod = 1;
    P.vec_store = (cout % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
                   (res_mode != 1 || (ld_res % 4 == 0 && ((uintptr_t)res & 15) == 0)))
                      ? 1
                      : 0;
    Q.tc = tc;
    Q.n_items = n_items;
    Q.blk0 = (uint32_t)(row0 / kRW);
    Q.items_cap = tapc_items_cap(K, ncls);
    Q.row_lo = (uint32_t)row0;
    TapcKernel kern = pick_tapc(ck, ck ? 1 : cin / 16, P.ntile_co, ncls);
    if (!kern) return INSMOS_EINVAL;
    // Cin = 32 -> Cout <= 16 with rows that are whole 128-byte lines: whole-row gathers + weight fragments once per tap (same bits;
    // INSMOS_TAPC_ROW32: 0 = the chunk-gather kernel, 2 = the whole-row kernel for Cout 32 too -- A/B runs; read per call)
    const int row32 = [] { const char* e = getenv("INSMOS_TAPC_ROW32"); return e ? atoi(e) : 1; }();
    const bool lines32 = cin == 32 && ld_in == 32 && ((uintptr_t)in & 127) == 0;
    if (row32 && lines32 && (P.ntile_co == 1 || row32 == 2)) {
        if (P.ntile_co == 1) kern = ncls == 4 ? k_conv_tapc32<1, 4> : k_conv_tapc32<1, 1>;
        else kern = ncls == 4 ? k_conv_tapc32<2, 4> : k_conv_tapc32<2, 1>;
    }
    const uint32_t n_blk = (uint32_t)insmos_tapc_blocks(n_out) - Q.blk0;
    ProfScope ps(KK_SPARSE_CONV, s);
    ps.meta[0] = K; ps.meta[1] = cin; ps.meta[2] = cout; ps.meta[3] = n_out - row0;
    INSMOS_LAUNCH(kern, dim3(n_blk), dim3(64 * ncls), 0, s, Q);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
