// insmos_amd/csrc/forward.hip -- native host orchestration of the InsMOS forward (InsMOS_Model.forward, 'test') for a
// BATCH of B windows in ONE set of launches: the same sequence of C-ABI operator calls insmos_amd/engine.py issues step
// by step for one window, driven from C++ so that a batch costs ONE foreign call (no interpreter work, no GIL).
// Where the reference walks its batch list window by window (models/models.py:313), here the B windows share every
// launch: in the 4D branch the window index is folded into the time coordinate (t' = t * B + b: time has no bounds and
// no striding, so windows never meet and the rows of the newest scans of ALL windows stay one suffix), in the 3D branch
// it is spconv's batch column, the BEV images are stacked along the row axis, and the head / NMS / instance kernels
// run over a (window, .) grid.  Every window gets exactly the bits it gets alone (tests/test_gpu_batched.py).  Layer order and channel widths follow the reference modules
// (models/backbones_3d/motionnet.py:21-50, models/MinkowskiEngine/minkunet.py:139-181,
// models/backbones_3d/spconv_unet.py:267-416, models/backbones_2d/*.py, models/post_process.py:112-224);
// insmos_amd/engine.py carries the same graph in inspectable form and tests/test_gpu_model.py asserts that both
// produce identical bits.  All device memory comes from a caller-provided arena (bump-allocated per window).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"
#include "prec.h"

namespace {
using namespace insmos;

struct Ctx {
    InsmosNetCfg cfg;
    std::map<std::string, InsmosConvW> L;
    std::vector<int32_t> off81[4];
    std::vector<int32_t> d_subm, d_inv, d_down5, d_inv5;
    // launch sets whose points left the packed-key box (+-2048 voxels in x / y, +-256 in z: coords.hip k_quant_keys_p) pay a second
    // quantise + sort + read-back through the fallback; a context that has seen one starts at the pair sort from then on (far
    // returns are a property of the sensor / voxel size, not of one window).  Re-armed every 256 sets.
    mutable std::atomic<int> packed_overflow{0};
    // constant vectors of the dense BEV stack ((n_bev_layers + 1) x 128 floats; csrc/bev.hip, SKIP): functions of the weights only,
    // computed with the product kernel on the first forward of the context
    mutable std::mutex bev_mu;
    mutable float* bev_const = nullptr;
    mutable bool bev_chead = false;   // the 64 floats behind the layer constants hold the deblock + head constant (insmos_deconv_head_constant)
    ~Ctx() {
        if (bev_const) (void)hipFree(bev_const);
    }
};

// c_0 = relu(bias_0), c_l = layer l applied to a neighbourhood that is c_{l-1} everywhere (insmos_bev_constant)
// *chead (optional): the fused deblock + heads applied to the last constant (4 x 16 floats), null when that layer pair is not the
// fused kernel's shape
int ensure_bev_constants(const Ctx& C, hipStream_t s, const float** out, const float** chead = nullptr) {
    std::lock_guard<std::mutex> lk(C.bev_mu);
    if (!C.bev_const) {
        const int L = C.cfg.n_bev_layers + 1;
        float* cv = nullptr;
        float* ws = nullptr;
        HIP_TRY(hipMalloc(&cv, ((size_t)L * 128 + 64) * sizeof(float)));
        size_t wsf = 0;
        for (int l = 0; l < L; ++l) {
            const InsmosConvW& w = C.L.at("bev" + std::to_string(l));
            wsf = std::max(wsf, insmos_bev_constant_ws_floats(w.cin, w.cout));
        }
        const auto wd = C.L.find("deconv"), wh = C.L.find("head");
        const InsmosConvW& wl = C.L.at("bev" + std::to_string(L - 1));
        const bool fused = wd != C.L.end() && wh != C.L.end() && C.cfg.up_ch == 256 && wl.cout % 16 == 0 && wl.cout <= 128 &&
                           C.cfg.head_ld <= 16 && wd->second.cin == wl.cout;
        if (fused) wsf = std::max(wsf, insmos_deconv_head_constant_ws_floats(wl.cout));
        if (hipMalloc(&ws, wsf * sizeof(float)) != hipSuccess) { (void)hipFree(cv); return INSMOS_EHIP; }
        int rc = INSMOS_OK;
        // the constants are cached for the context's lifetime and must be the EXACT-fp32 kernel's: pin this thread to mode 0 while
        // they are computed (a forward under the split-bf16 experiment does not use them: run_bev_stack skips nothing in mode 3)
        struct PrecPin {
            int saved;
            PrecPin() : saved(insmos::conv_precision_thread_get()) { (void)insmos_conv_precision_thread(0); }
            ~PrecPin() { (void)insmos_conv_precision_thread(saved); }
        } pin;
        for (int l = 0; l < L && rc == INSMOS_OK; ++l) {
            const InsmosConvW& w = C.L.at("bev" + std::to_string(l));
            rc = insmos_bev_constant(w.w, w.b, w.cin, w.cout, 1, l ? cv + (size_t)(l - 1) * 128 : nullptr, cv + (size_t)l * 128, ws, s);
        }
        if (rc == INSMOS_OK && fused)
            rc = insmos_deconv_head_constant(wd->second.w, wd->second.b, wl.cout, C.cfg.up_ch, wh->second.w, wh->second.b, C.cfg.head_ld,
                                             cv + (size_t)(L - 1) * 128, cv + (size_t)L * 128, ws, s);
        if (rc == INSMOS_OK && hipStreamSynchronize(s) != hipSuccess) rc = INSMOS_EHIP;
        (void)hipFree(ws);
        if (rc != INSMOS_OK) { (void)hipFree(cv); return rc; }
        C.bev_chead = fused;
        C.bev_const = cv;
    }
    *out = C.bev_const;
    if (chead) *chead = C.bev_chead ? C.bev_const + (size_t)(C.cfg.n_bev_layers + 1) * 128 : nullptr;
    return INSMOS_OK;
}

struct Arena {
    char* base;
    size_t cap, off = 0;
    bool ok = true;
    template <class T>
    T* take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        if (off + bytes > cap) { ok = false; off += bytes; return (T*)base; }  // keep counting: `off` = bytes needed
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
};

// ME kernel-region tap order (x fastest; odd sizes centred), offsets scaled by the tensor stride -- engine.py
std::vector<int32_t> me_offsets(const int ks[4], const int ts[4]) {
    std::vector<int32_t> o;
    for (int it = 0; it < ks[3]; ++it)
        for (int iz = 0; iz < ks[2]; ++iz)
            for (int iy = 0; iy < ks[1]; ++iy)
                for (int ix = 0; ix < ks[0]; ++ix) {
                    const int idx[4] = {ix, iy, iz, it};
                    for (int d = 0; d < 4; ++d) o.push_back((ks[d] % 2 == 1 ? idx[d] - (ks[d] - 1) / 2 : idx[d]) * ts[d]);
                }
    return o;
}

// tc / ni: the table's tap-compacted item lists (csrc/spconv_tapc.hip) for one chain ([0]) and for four tap classes ([1]), or null
// (tc_row0: the first row the item lists are built from)
struct Table {
    int32_t* nbr; uint32_t* mask; int K; int64_t n;
    uint32_t* tc[2] = {nullptr, nullptr};
    int32_t* ni[2] = {nullptr, nullptr};
    int64_t tc_row0[2] = {0, 0};
};
// launch sets of this many windows or more walk compacted row-group lists in the skipping BEV layers (measured on sets of 8: the six
// layers 1 844 -> 1 676 us per set, bench 705-712 -> 714 scans/s; ONE window is 1.4 % slower that way -- few workgroups either way,
// and the list kernel's are heavier: profiles/r04_bev_list_ab.txt)
constexpr int kBevSkipListMinB = 4;
int g_regroup = -1;                 // row regrouping modes of the 3D levels set by insmos_forward_regroup; -1 = the default
constexpr int kRegroupDefault = 3553;   // 4096-row blocks; levels 2 and 3 with the parity class above the signature (their inverse maps feed 64- and 32-channel layers; measured)
inline bool regroup_modes_ok(int v) {
    if (v < 0 || v > 5555) return false;
    for (int l = 0; l < 4; ++l, v /= 10)
        if (v % 10 > 5) return false;
    return true;
}
int64_t g_table_limit = 1ll << 31;  // bytes a neighbour table may span (32-bit offsets); lowered by tests (insmos_debug_table_limit)

#define CK(expr)                                  \
    do {                                          \
        int rc__ = (expr);                        \
        if (rc__ != INSMOS_OK) return rc__;       \
    } while (0)
#define NEED_ARENA()                              \
    do {                                          \
        if (!A.ok) { out->arena_needed = (int64_t)A.off; return INSMOS_EWORKSPACE; } \
    } while (0)

// A second stream per HOST THREAD (the launch-set workers of InsMOS_Model are persistent threads) for the work that does not sit
// on the convolution chain: the 3D branch's coordinate sets and kernel maps depend on the points' positions only, so they are
// built on this stream WHILE MotionNet's convolutions run on the caller's; the level-0 81-tap table and the later one-hot
// passes go there too.  Cross-stream order is explicit (events); INSMOS_TWO_STREAMS=0 puts everything back on one stream.
struct Aux {
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    int dev = -1;   // the device s2 and the events belong to: a host thread that moves to another GPU gets new ones
    void release() {
        if (s2) (void)hipStreamDestroy(s2);
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        s2 = nullptr;
        ev.clear();
        used = 0;
        dev = -1;
    }
    int init() {
        int cur = -1;
        HIP_TRY(hipGetDevice(&cur));
        if (s2 && cur != dev) {   // (resources of the previous device are released from here: destroy calls take any current device)
            (void)hipStreamSynchronize(s2);
            release();
        }
        if (!s2) {
            HIP_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            dev = cur;
        }
        used = 0;
        return INSMOS_OK;
    }
    int next(hipEvent_t* e) {
        if (used == ev.size()) {
            hipEvent_t n;
            HIP_TRY(hipEventCreateWithFlags(&n, hipEventDisableTiming));
            ev.push_back(n);
        }
        *e = ev[used++];
        return INSMOS_OK;
    }
};
thread_local Aux tl_aux;
thread_local int tl_stream_mask = -1;   // per-host-thread override of INSMOS_TWO_STREAMS (insmos_forward_streams); -1 = none

// `to` waits for everything enqueued on `from` so far
int link_streams(hipStream_t from, hipStream_t to) {
    if (from == to) return INSMOS_OK;
    hipEvent_t e;
    int rc = tl_aux.next(&e);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e, from));
    HIP_TRY(hipStreamWaitEvent(to, e, 0));
    return INSMOS_OK;
}
// whatever path the function leaves by, the caller's stream ends up behind the second one (the arena is the caller's to reuse)
struct JoinGuard {
    hipStream_t s, s2;
    ~JoinGuard() { (void)link_streams(s2, s); }
};

// Count read-backs.  A launch set has about a dozen of them and, for ONE window per forward(), each is an idle gap on the GPU
// (profiles/r04_b1_trace.txt: ~27 us per read-back = copy kernel + host wake-up + next launch).  Two things shorten the gap:
// the copy lands in PINNED host memory of the calling thread (a pageable destination is staged by the runtime), and in latency mode
// (tl_spin: one launch set per forward(), set with the second-stream mask) the host polls the stream instead of sleeping in
// hipStreamSynchronize.  With several sets in flight the threads sleep as before (a spinning thread per set would take the cores the
// other sets' launch threads need).
struct PinnedCounts {
    int32_t* p = nullptr;
    ~PinnedCounts() {}   // (left to process teardown: freeing pinned memory after the runtime is gone crashes)
    int32_t* get() {
        if (!p && hipHostMalloc((void**)&p, 4096, hipHostMallocDefault) != hipSuccess) p = nullptr;
        return p;
    }
};
thread_local PinnedCounts tl_pinned;
thread_local bool tl_spin = false;
static const int kSpinBudgetUs = [] { const char* e = getenv("INSMOS_READBACK_SPIN_US"); return e ? std::max(0, atoi(e)) : 300; }();
static const bool kSpinEnv = [] { const char* e = getenv("INSMOS_READBACK_SPIN"); return !(e && e[0] == '0'); }();

int wait_stream(hipStream_t s) {
    if (tl_spin && kSpinEnv) {
        // bounded: poll for ~300 us (INSMOS_READBACK_SPIN_US) (a latency-mode read-back returns within tens of us; the final wait for the box counts is the
        // long one), then sleep in the runtime like everyone else -- a host thread per GPU must not burn a core for milliseconds
        hipError_t q;
        const auto t0 = std::chrono::steady_clock::now();
        int polls = 0;
        while ((q = hipStreamQuery(s)) == hipErrorNotReady) {
            if ((++polls & 15) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinBudgetUs)) {
                q = hipStreamSynchronize(s);
                break;
            }
        }
        if (q != hipSuccess) { insmos::g_last_hip_error = (int)q; return INSMOS_EHIP; }
        return INSMOS_OK;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return INSMOS_OK;
}

// Host timeline of the calling thread's last forward (INSMOS_HOST_MARKS=1; insmos_forward_host_marks): when the HOST passed each
// stage -- enqueue done / read-back returned -- in microseconds since the call began.  A single window's latency is a chain of host
// round trips and launch bursts; a GPU-side trace (tools/b1_trace.py) shows the gaps, this shows which side was waiting.
struct HostMarks {
    static constexpr int kMax = 40;
    const char* name[kMax];
    double us[kMax];
    int n = 0;
    std::chrono::steady_clock::time_point t0;
};
thread_local HostMarks tl_marks;
static const bool kMarksOn = [] { const char* e = getenv("INSMOS_HOST_MARKS"); return e && e[0] == '1'; }();
inline void host_mark(const char* what) {
    if (!kMarksOn) return;
    HostMarks& m = tl_marks;
    if (m.n == 0) m.t0 = std::chrono::steady_clock::now();
    if (m.n < HostMarks::kMax) {
        m.name[m.n] = what;
        m.us[m.n] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - m.t0).count();
        ++m.n;
    }
}

int read_counts(const int32_t* dev, int32_t* host, int n, hipStream_t s) {
    int32_t* pin = (size_t)n * sizeof(int32_t) <= 4096 ? tl_pinned.get() : nullptr;
    HIP_TRY(hipMemcpyAsync(pin ? pin : host, dev, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    int rc = wait_stream(s);
    if (rc != INSMOS_OK) return rc;
    if (pin) memcpy(host, pin, n * sizeof(int32_t));
    return INSMOS_OK;
}
}  // namespace

extern "C" int insmos_ctx_create(const InsmosNetCfg* cfg, const char* const* names, const InsmosConvW* layers,
                                 int n_layers, void** ctx_out) {
    if (!cfg || !names || !layers || n_layers <= 0 || !ctx_out) return INSMOS_EINVAL;
    Ctx* c = new Ctx();
    c->cfg = *cfg;
    for (int i = 0; i < n_layers; ++i) {
        if (!names[i] || !layers[i].w || !layers[i].b) { delete c; return INSMOS_EINVAL; }
        c->L[names[i]] = layers[i];
    }
    const int k3[4] = {3, 3, 3, 3};
    for (int l = 0; l < 4; ++l) {
        const int ts[4] = {1 << l, 1 << l, 1 << l, 1};
        c->off81[l] = me_offsets(k3, ts);
    }
    for (int kz = 0; kz < 3; ++kz)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int32_t a[4] = {0, kz - 1, ky - 1, kx - 1}, b[4] = {0, 1 - kz, 1 - ky, 1 - kx};
                c->d_subm.insert(c->d_subm.end(), a, a + 4);  // also the stride-2 table: i = 2*o - 1 + k
                c->d_inv.insert(c->d_inv.end(), b, b + 4);    // o = (i + 1 - k) / 2
            }
    for (int kz = 0; kz < 3; ++kz) {
        const int32_t a[4] = {0, kz, 0, 0}, b[4] = {0, -kz, 0, 0};
        c->d_down5.insert(c->d_down5.end(), a, a + 4);
        c->d_inv5.insert(c->d_inv5.end(), b, b + 4);
    }
    *ctx_out = c;
    return INSMOS_OK;
}

extern "C" int insmos_ctx_destroy(void* ctx) {
    delete (Ctx*)ctx;
    return INSMOS_OK;
}

static int forward_windows_impl(void* ctx, const float* const* pts_host, const int64_t* n_pts_host, int B, int ld,
                                void* arena, size_t arena_bytes, void* stream, InsmosForwardOut* outs) {
    if (!ctx || !pts_host || !n_pts_host || B < 1 || B > INSMOS_MAX_BATCH || ld < 5 || !arena || !outs) return INSMOS_EINVAL;
    int64_t N = 0;
    for (int b = 0; b < B; ++b) {
        if (!pts_host[b] || n_pts_host[b] <= 0) return INSMOS_EINVAL;
        N += n_pts_host[b];
    }
    const Ctx& C = *(const Ctx*)ctx;
    const InsmosNetCfg& g = C.cfg;
    hipStream_t s = (hipStream_t)stream;
    // INSMOS_TWO_STREAMS: bit 0 = level-0 table, bit 1 = the 3D coordinate phase, bit 2 = inv_conv_out, bit 3 = the finer one-hot
    // passes on the second stream (default 15 = all; 0 = everything on the caller's stream)
    static const int ts_env = [] { const char* e = getenv("INSMOS_TWO_STREAMS"); return e ? atoi(e) : -1; }();
    const int ts_mask = ts_env >= 0 ? ts_env : tl_stream_mask >= 0 ? tl_stream_mask : 15;
    tl_spin = ts_mask != 0;   // latency mode: one launch set per forward() (insmos_amd/models.py passes mask 15 then, 0 otherwise)
    hipStream_t s2 = s;
    if (ts_mask) {
        CK(tl_aux.init());
        s2 = tl_aux.s2;
    }
    JoinGuard join_guard{s, s2};
    const hipStream_t s2_tab0 = (ts_mask & 1) ? s2 : s, s2_coords = (ts_mask & 2) ? s2 : s, s2_inv = (ts_mask & 4) ? s2 : s,
                      s2_oh = (ts_mask & 8) ? s2 : s;
    tl_marks.n = 0;
    host_mark("begin");
    memset(outs, 0, sizeof(*outs) * (size_t)B);
    InsmosForwardOut* out = outs;  // batch-wide figures and error details go to the first entry
    for (int b = 0; b < B; ++b) outs[b].batch = B;
    Arena A{(char*)arena, arena_bytes};
    // INSMOS_CONV_TAPC (read per call: tools flip it inside one process): which 81-tap layers of MotionNet's levels 1..3 run in their
    // tap-compacted form (csrc/spconv_tapc.hip; same bits): bit 0 = the tap-split layers (block3.*, block6.*, block7.conv1), bit 1 = the
    // one-chain 16-channel layers (block2.conv2, block7.conv2), bit 2 = the one-chain 8-channel layers (block1.*, block2.conv1); 0 = off;
    // bit 3 (opt-in, NOT in the default: measured 81 -> 92 us per layer and set, bench +-0 -- 16-24 MFMAs per item do not carry the
    // item pipeline's overhead; profiles/r06_tapc27_layers_ab.csv) = the 27-tap tap-split layers of the 3D UNet's level 2 (conv2.1 /
    // conv2.2, conv_up_t2.*, conv_up_instance_block_up3: 32 / 48 -> 32 channels on the SubMConv3d table subm2, spconv_unet.py:138-144)
    const int tapc_mode = [] { const char* e = getenv("INSMOS_CONV_TAPC"); return e ? atoi(e) : 1; }();
    auto Lr = [&](const std::string& name) -> const InsmosConvW* {
        auto it = C.L.find(name);
        return it == C.L.end() ? nullptr : &it->second;
    };
    // out[:, col_out : col_out + cout] = epilogue(conv(x[:, col_in : col_in + cin]))  (Engine.conv)
    auto conv = [&](const char* name, const float* x, int64_t n_in, int ld_in, int col_in, const Table* t, int64_t n_out,
                    float* o, int ld_out, int col_out, const float* res, int ld_res, int col_res, int res_mode,
                    int relu_pre, int relu_post, int64_t row0 = 0, hipStream_t cs = nullptr) -> int {
        const InsmosConvW* w = Lr(name);
        if (!w) return INSMOS_EINVAL;
        if (n_out == 0) return INSMOS_OK;
        if (t && (t->K != w->K || t->n != n_out)) return INSMOS_EINVAL;
        if (t && (t->tc[0] || t->tc[1]) && insmos::conv_precision() == 0) {
            // the tap-compacted form of the layer, when the table carries the item lists of the layer's summation order (same bits)
            const int ncls = insmos_conv_tap_classes(w->K, w->cin, w->cout, t->mask ? 1 : 0);
            const int slot = ncls == 4 ? 1 : 0;
            const int bit = w->K == 27 ? 8 : ncls == 4 ? 1 : w->cin == 8 ? 4 : 2;
            if ((ncls == 1 || ncls == 4) && (tapc_mode & bit) && t->tc[slot] && row0 >= t->tc_row0[slot] && w->cout <= 32 && (w->cin == 8 || (w->cin % 16 == 0 && w->cin <= 48)) &&
                n_in < (1ll << 23) - 1)
                return insmos_sparse_conv_tapc_rows(x + col_in, n_in, ld_in, w->cin, t->tc[slot], t->ni[slot], ncls, w->K, n_out, row0, w->w,
                                                    w->b, o + col_out, ld_out, w->cout, res ? res + col_res : nullptr, ld_res, res_mode,
                                                    relu_pre, relu_post, cs ? cs : s);
        }
        return insmos_sparse_conv_rows(x + col_in, n_in, ld_in, w->cin, t ? t->nbr : nullptr, t ? t->mask : nullptr, w->K,
                                       n_out, row0, w->w, w->b, o + col_out, ld_out, w->cout, res ? res + col_res : nullptr,
                                       ld_res, res_mode, relu_pre, relu_post, cs ? cs : s);
    };
    auto table = [&](int K, int64_t n) {
        Table t;
        t.nbr = A.take<int32_t>((size_t)K * n);
        t.mask = A.take<uint32_t>((size_t)((n + 15) / 16) * 4);
        t.K = K;
        t.n = n;
        return t;
    };
    int32_t hc[8 + INSMOS_MAX_BATCH];
    int32_t* counts = A.take<int32_t>(8 + INSMOS_MAX_BATCH);
    NEED_ARENA();

    // =============================== MotionNet (4D) ===============================
    int64_t n[4];
    uint64_t* keys[4];
    int32_t* coords[4];
    int32_t *parent[3], *cstart[3];
    uint32_t* cmask[3];
    keys[0] = A.take<uint64_t>(N);
    coords[0] = A.take<int32_t>(4 * N);
    int32_t* inverse = A.take<int32_t>(N);
    int32_t* cur_index = A.take<int32_t>(N);
    {
        const size_t wsb = insmos_quantize4d_ws_bytes(N);
        const size_t mark = A.off;
        void* ws = A.take<char>(wsb);
        NEED_ARENA();
        const float quant[4] = {g.vs[0], g.vs[0], g.vs[0], g.dt};
        // packed 40-bit sort keys with the point index in the low bits first (sets of <= 8 windows, < 2^24 points, z within
        // +-256 voxels), then the 40(+)-bit pair sort, then the full-width sort (windows wider than +-2048 voxels / 16 time steps)
        static const bool packed_keys = [] { const char* e = getenv("INSMOS_PACKED_KEYS"); return !(e && e[0] == '0'); }();
        int skip_packed = C.packed_overflow.load(std::memory_order_relaxed);
        if (skip_packed > 0) C.packed_overflow.store(skip_packed - 1, std::memory_order_relaxed);
        for (int mode = (packed_keys && !skip_packed && B <= 8 && N < (1ll << 24)) ? 2 : 1; mode >= 0; --mode) {
            CK(insmos_quantize4d_windows(pts_host, n_pts_host, B, ld, quant, keys[0], coords[0], inverse, cur_index, counts, ws, wsb,
                                         mode, s));
            host_mark("quantize enqueued");
            CK(read_counts(counts, hc, 5 + B, s));
            host_mark("quantize counts back");
            if (hc[3] == 0) break;
            if (mode == 2) C.packed_overflow.store(256, std::memory_order_relaxed);
        }
        A.off = mark;  // the sort workspace is dead once the counts are back
    }
    n[0] = hc[0];
    const int64_t ncur = hc[1];
    int64_t cur_start[INSMOS_MAX_BATCH + 1];
    for (int b = 0; b <= B; ++b) cur_start[b] = hc[4 + b];
    for (int b = 0; b < B; ++b) outs[b].n_cur = cur_start[b + 1] - cur_start[b];
    // the windows' current-point starts stay on the device for the voxeliser (the counts slots are reused below)
    int32_t* cur_start_dev = A.take<int32_t>(INSMOS_MAX_BATCH + 1);
    NEED_ARENA();
    HIP_TRY(hipMemcpyAsync(cur_start_dev, counts + 4, (size_t)(B + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // (a batch narrows the time field by B: a window that is valid alone may not fit beside B - 1 others -- the caller
    //  splits the set down to single windows, where the full +-32768 range applies, before giving up)
    if (hc[2] != 0) { out->n_out_of_window = hc[2]; return B > 1 ? INSMOS_EBATCH : INSMOS_EINVAL; }
    out->me_voxels[0] = n[0];
    if (n[0] == 0) return INSMOS_EINVAL;
    for (int b = 0; b < B; ++b)
        if (outs[b].n_cur == 0) return INSMOS_EINVAL;  // (a window without t == 0 points; the caller reads n_cur)
    // Dead-row elimination (same as Engine.motionnet): starts[l][d] = first level-l row with
    // t >= t_last - d; a layer whose output is needed d scans back computes rows [starts[l][d], n[l]) only.
    int32_t starts[4][16];
    // INSMOS_LEVEL_CHAIN: 1 (default) = a single window's three level-downs and the four time-slice searches run as ONE chain of
    // launches with the counts kept on the device (insmos_level_down4d_chain) and ONE read-back instead of four -- each read-back of
    // this section is an idle gap on the GPU, nothing else is queued yet; 2 = launch sets too (their scans then run over the finest
    // level's row count three times: not worth it with other sets in flight); 0 = level by level.
    // (read per call, not cached: tools/b1_ab.py flips it inside one process)
    const int chain_env = [] { const char* e = getenv("INSMOS_LEVEL_CHAIN"); return e ? atoi(e) : 1; }();
    if (((chain_env == 1 && B == 1) || chain_env >= 2) && n[0] < (1ll << 24)) {
        uint64_t* ok[3];
        int32_t* oc[3];
        for (int l = 1; l <= 3; ++l) {   // (room for n[0] rows each: every level's count is bounded by the finest one's)
            keys[l] = ok[l - 1] = A.take<uint64_t>(n[0]);
            coords[l] = oc[l - 1] = A.take<int32_t>(4 * n[0]);
            parent[l - 1] = A.take<int32_t>(n[0]);
            cstart[l - 1] = A.take<int32_t>(n[0]);
            cmask[l - 1] = A.take<uint32_t>(n[0]);
        }
        int32_t* chain = A.take<int32_t>(4 + 64);
        const size_t wsb = insmos_level_down4d_ws_bytes(n[0]);
        const size_t mark = A.off;
        void* ws = A.take<char>(wsb);
        NEED_ARENA();
        CK(insmos_level_down4d_chain(keys[0], n[0], 3, B, ok, oc, parent, cstart, cmask, chain, ws, wsb, s));
        int32_t hch[4 + 64];
        host_mark("level chain enqueued");
        CK(read_counts(chain, hch, 4 + 64, s));
        host_mark("level counts back");
        A.off = mark;
        for (int l = 1; l <= 3; ++l) n[l] = hch[l - 1];
        memcpy(&starts[0][0], hch + 4, sizeof(starts));
    } else {
        for (int l = 1; l <= 3; ++l) {
            const int64_t np = n[l - 1];
            keys[l] = A.take<uint64_t>(np);
            coords[l] = A.take<int32_t>(4 * np);
            parent[l - 1] = A.take<int32_t>(np);
            cstart[l - 1] = A.take<int32_t>(np);
            cmask[l - 1] = A.take<uint32_t>(np);
            const size_t wsb = insmos_level_down4d_ws_bytes(np);
            const size_t mark = A.off;
            void* ws = A.take<char>(wsb);
            NEED_ARENA();
            CK(insmos_level_down4d(keys[l - 1], np, l, keys[l], coords[l], parent[l - 1], cstart[l - 1], cmask[l - 1], counts, ws,
                                   wsb, s));
            CK(read_counts(counts, hc, 1, s));
            A.off = mark;
            n[l] = hc[0];
        }
        int32_t* sd = A.take<int32_t>(64);
        NEED_ARENA();
        for (int l = 0; l < 4; ++l) CK(insmos_tslice_starts_batched(keys[l], n[l], 16, B, sd + 16 * l, s));
        CK(read_counts(sd, &starts[0][0], 64, s));
    }
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < 4; ++l) outs[b].me_voxels[l] = n[l];  // batch totals
    auto row_from = [&](int l, int d) -> int64_t { return d < 16 ? starts[l][d] : 0; };

    // kernels address a table with 32-bit byte offsets: a batch whose finest 81-tap table would pass 2 GiB is refused
    // (the caller runs it as two smaller launch sets); a single window of that size is simply too large
    if ((int64_t)81 * n[0] * 4 >= g_table_limit) return B > 1 ? INSMOS_EBATCH : INSMOS_EINVAL;
    // only the coarsest level is searched; every finer table is derived through the Morton hierarchy
    Table nbr81[4];
    nbr81[3] = table(81, n[3]);
    NEED_ARENA();
    // the time offsets of a SEARCHED table move by B (t' = t * B + b): the only arithmetic on t in the whole branch
    auto search81 = [&](int lvl, const int32_t* cds, const uint64_t* kys, int64_t nn, Table& t) -> int {
        const int k3[4] = {3, 3, 3, 3};
        const int ts[4] = {1 << lvl, 1 << lvl, 1 << lvl, 1};
        std::vector<int32_t> off = me_offsets(k3, ts);
        for (size_t k = 0; k < off.size() / 4; ++k) off[4 * k + 3] *= B;
        const int32_t one[4] = {1, 1, 1, 1};
        return insmos_build_nbr(cds, nn, kys, nullptr, nn, 0, nullptr, off.data(), 81, one, one, t.nbr, t.mask, s);
    };
    // One more level of the hierarchy for launch sets (B >= 2): searching 81 taps of the stride-16 voxels (a third of the stride-8
    // ones) and deriving the stride-8 table from it costs a third of searching stride 8; a single window gains nothing (one more
    // count read-back), so it searches stride 8 directly.
    static const bool deep_tables = [] { const char* e = getenv("INSMOS_TABLES_LEVEL4"); return !(e && e[0] == '0'); }();
    // Sparse stores for the FINEST table (read by the convolutions only): an entry of a (16-row group, tap) slot outside the
    // group's active-tap mask is never read by the 16-row-tile kernels, so it is not written.  The coarser tables are written in
    // full -- measured: with sparse stores at every level (the derivations then read the coarse table through its mask array,
    // insmos_nbr81_from_coarse_rows_masked, tests/test_gpu_coords.py) the resolver's per-tap ballots and the mask loads in front of
    // every coarse entry cost more than the stores they save (2.39 -> 2.61 ms per 7 launch sets; masks on fully written tables:
    // 2.95).  The first layer's cubes do take the mask array of the level-1 table (a slot outside the mask is "no neighbour" for
    // all 16 rows, known without loading the entry: 118 -> 106 us).  INSMOS_TABLES_DENSE=1 writes everything.
    static const bool sparse_tab = [] {
        const char* e = getenv("INSMOS_TABLES_DENSE");
        const char* jt = getenv("INSMOS_CK_JT");
        return !(e && e[0] == '1') && !(jt && atoi(jt) > 1);
    }();
    if (deep_tables && B >= 2) {
        const int64_t np = n[3];
        uint64_t* keys4 = A.take<uint64_t>(np);
        int32_t* coords4 = A.take<int32_t>(4 * np);
        int32_t* parent3 = A.take<int32_t>(np);
        int32_t* cstart3 = A.take<int32_t>(np);
        uint32_t* cmask3 = A.take<uint32_t>(np);
        const size_t wsb = insmos_level_down4d_ws_bytes(np);
        void* ws = A.take<char>(wsb);   // (not rewound: the level-0 table below is written on the second stream)
        NEED_ARENA();
        CK(insmos_level_down4d(keys[3], np, 4, keys4, coords4, parent3, cstart3, cmask3, counts, ws, wsb, s));
        CK(read_counts(counts, hc, 1, s));
        const int64_t n4 = hc[0];
        Table t4 = table(81, n4);
        NEED_ARENA();
        CK(search81(4, coords4, keys4, n4, t4));
        CK(insmos_nbr81_from_coarse_rows(coords[3], n[3], 0, parent3, 3, t4.nbr, n4, cstart3, cmask3, nbr81[3].nbr, nbr81[3].mask, s));
    } else {
        CK(search81(3, coords[3], keys[3], n[3], nbr81[3]));
    }
    for (int l = 2; l >= 0; --l) {
        nbr81[l] = table(81, n[l]);
        NEED_ARENA();
        // (the level-0 table is read by block8 only -- rows of the last two scans, at the very end of the branch: it is built on
        //  the second stream, off the convolution chain)
        if (l == 0) CK(link_streams(s, s2_tab0));
        CK(((sparse_tab && l == 0) ? insmos_nbr81_from_coarse_rows_sparse : insmos_nbr81_from_coarse_rows)(
            coords[l], n[l], l == 0 ? row_from(0, 1) : 0, parent[l], l, nbr81[l + 1].nbr, n[l + 1], cstart[l], cmask[l], nbr81[l].nbr,
            nbr81[l].mask, l == 0 ? s2_tab0 : s));
    }
    hipEvent_t ev_tab0 = nullptr;
    if (s2_tab0 != s) {
        CK(tl_aux.next(&ev_tab0));
        HIP_TRY(hipEventRecord(ev_tab0, s2_tab0));
    }
    // Tap-compacted item lists of the level 1..3 tables (csrc/spconv_tapc.hip): the 81-tap layers of those levels run on dense
    // per-tap groups of 16 rows instead of 16-row tiles (half of whose MFMA passes multiply absent rows) -- tapc_mode above.
    {
        const int tapc = tapc_mode;
        for (int l = 1; l <= 3 && tapc; ++l) {
            // classes the level's layers need (forward order below): [0] one chain, [1] four classes
            const bool need[2] = {((tapc & 2) && (l == 1 || l == 2)) || ((tapc & 4) && (l == 1 || l == 2)), (tapc & 1) != 0};
            if (n[l] >= (1ll << 23) - 1) continue;
            // first row any tap-compacted layer of the level computes (the blocks below are not built): the one-chain layers are
            // the encoder's (every row) and block7.conv2 (depth 2); the tap-split ones block3.conv1 (7), block6.conv1 (5), block7.conv1 (3)
            const int64_t first_row[2] = {l == 1 && !(tapc & 4) ? row_from(1, 2) : 0, row_from(l, l == 3 ? 7 : l == 2 ? 5 : 3)};
            for (int c = 0; c < 2; ++c) {
                if (!need[c]) continue;
                const int ncls = c ? 4 : 1;
                nbr81[l].tc[c] = A.take<uint32_t>(insmos_tapc_words(81, n[l], ncls));
                nbr81[l].ni[c] = A.take<int32_t>((size_t)insmos_tapc_blocks(n[l]) * ncls);
                nbr81[l].tc_row0[c] = first_row[c] & ~(int64_t)127;
                NEED_ARENA();
                CK(insmos_tapc_build(nbr81[l].nbr, 81, n[l], nbr81[l].tc_row0[c], ncls, nbr81[l].tc[c], nbr81[l].ni[c], s));
            }
        }
    }
    Table dn[3], up[3];
    for (int l = 0; l < 3; ++l) {
        dn[l] = table(8, n[l + 1]);
        up[l] = table(8, n[l]);
        NEED_ARENA();
        CK(insmos_nbr_down_up(coords[l], n[l], parent[l], l, n[l + 1], cstart[l], cmask[l], dn[l].nbr, dn[l].mask, up[l].nbr,
                              up[l].mask, s));
    }
    float* cat8 = A.take<float>(n[0] * 16);  // [convtr7 (8) | out_p1 (8)]
    float* cat7 = A.take<float>(n[1] * 32);  // [convtr6 (16) | out_b1p2 (8) | zero pad (8)]
    float* cat6 = A.take<float>(n[2] * 48);  // [convtr5 (32) | out_b2p4 (16)]
    float* x1 = A.take<float>(n[1] * 8);
    float* x2 = A.take<float>(n[2] * 8);
    float* x3 = A.take<float>(n[3] * 16);
    float* b3 = A.take<float>(n[3] * 32);
    float* b6 = A.take<float>(n[2] * 32);
    float* b7 = A.take<float>(n[1] * 16);
    float* b8 = A.take<float>(n[0] * 8);
    float* motion = A.take<float>(n[0] * 4);
    float* tmp_t = A.take<float>(std::max<int64_t>(std::max(n[0] * 8, n[1] * 16), std::max(n[2] * 32, n[3] * 32)));
    float* tmp_r = A.take<float>(std::max<int64_t>(std::max(n[0] * 8, n[1] * 16), std::max(n[2] * 32, n[3] * 32)));
    float* cur = A.take<float>(ncur * 8);
    NEED_ARENA();
    CK(insmos_fill_cols(cat7, n[1], 32, 24, 8, 0.0f, s));
    if (!g.w0_const || !g.b0_const) return INSMOS_EINVAL;
    // motionnet.py:29-32: every point carries the feature 0.5 -> conv0 needs no table and no gathers
    {
        // (occupancy cubes, 48 B per level-1 voxel; INSMOS_CONV0_CUBE=0 keeps the per-tap resolver for A/B runs: same bits)
        static const bool cube = [] { const char* e = getenv("INSMOS_CONV0_CUBE"); return !(e && e[0] == '0'); }();
        // (not handed back: what is allocated next is written on the SECOND stream, which is not ordered behind these kernels)
        void* cubes = cube ? (void*)A.take<uint4>((size_t)n[1] * 3) : nullptr;
        NEED_ARENA();
        CK(insmos_const_conv125_cubes(coords[0], n[0], parent[0], 0, nbr81[1].nbr, nbr81[1].mask, n[1], cstart[0], cmask[0], g.w0_const,
                                      g.b0_const, cat8 + 8, 16, 1, cubes, s));
    }
    CK(conv("conv1p1s2", cat8, n[0], 16, 8, &dn[0], n[1], x1, 8, 0, nullptr, 0, 0, 0, 0, 1));
    // BasicBlock (minkunet.py:63-124): conv1-bn-relu, conv2-bn, (+ downsample(x) | x), relu
    // (output needed `depth` scans back: conv2 / downsample on those rows, conv1 one scan further; depth 99 = all rows)
    auto block = [&](const std::string& name, const float* x, int64_t nn, int ld_x, int col_x, const Table* nb, int cout,
                     float* o, int ld_out, int col_out, int lvl, int depth) -> int {
        const int64_t r2 = row_from(lvl, depth), r1 = row_from(lvl, depth + 1);
        CK(conv((name + ".conv1").c_str(), x, nn, ld_x, col_x, nb, nn, tmp_t, cout, 0, nullptr, 0, 0, 0, 0, 1, r1));
        if (Lr(name + ".ds")) {
            CK(conv((name + ".ds").c_str(), x, nn, ld_x, col_x, nullptr, nn, tmp_r, cout, 0, nullptr, 0, 0, 0, 0, 0, r2));
            CK(conv((name + ".conv2").c_str(), tmp_t, nn, cout, 0, nb, nn, o, ld_out, col_out, tmp_r, cout, 0, 1, 0, 1, r2));
        } else {
            CK(conv((name + ".conv2").c_str(), tmp_t, nn, cout, 0, nb, nn, o, ld_out, col_out, x, ld_x, col_x, 1, 0, 1, r2));
        }
        return INSMOS_OK;
    };
    CK(block("block1.0", x1, n[1], 8, 0, &nbr81[1], 8, cat7, 32, 16, 1, 99));
    CK(conv("conv2p2s2", cat7, n[1], 32, 16, &dn[1], n[2], x2, 8, 0, nullptr, 0, 0, 0, 0, 1));
    CK(block("block2.0", x2, n[2], 8, 0, &nbr81[2], 16, cat6, 48, 32, 2, 99));
    CK(conv("conv3p4s2", cat6, n[2], 48, 32, &dn[2], n[3], x3, 16, 0, nullptr, 0, 0, 0, 0, 1));
    // decoder side: needed time depth = 0 at `final`, +1 per 3^4 conv on the way back (a block holds two)
    CK(block("block3.0", x3, n[3], 16, 0, &nbr81[3], 32, b3, 32, 0, 3, 6));
    CK(conv("convtr5p8s2", b3, n[3], 32, 0, &up[2], n[2], cat6, 48, 0, nullptr, 0, 0, 0, 0, 1, row_from(2, 6)));
    CK(block("block6.0", cat6, n[2], 48, 0, &nbr81[2], 32, b6, 32, 0, 2, 4));
    CK(conv("convtr6p4s2", b6, n[2], 32, 0, &up[1], n[1], cat7, 32, 0, nullptr, 0, 0, 0, 0, 1, row_from(1, 4)));
    CK(block("block7.0", cat7, n[1], 32, 0, &nbr81[1], 16, b7, 16, 0, 1, 2));
    CK(conv("convtr7p2s2", b7, n[1], 16, 0, &up[0], n[0], cat8, 16, 0, nullptr, 0, 0, 0, 0, 1, row_from(0, 2)));
    if (ev_tab0) HIP_TRY(hipStreamWaitEvent(s, ev_tab0, 0));
    CK(block("block8.0", cat8, n[0], 16, 0, &nbr81[0], 8, b8, 8, 0, 0, 0));
    CK(conv("final", b8, n[0], 8, 0, nullptr, n[0], motion, 4, 0, nullptr, 0, 0, 0, 0, 0, row_from(0, 0)));
    // MotionNet is now ENQUEUED on the caller's stream (no host synchronisation above this line since the level counts).  The 3D
    // branch's coordinate work -- voxel cells, strided coordinate sets, 13 kernel maps: sorts, scans and table kernels with five
    // count read-backs -- needs the current points' POSITIONS only, so it runs on the second stream while those convolutions
    // execute; the motion columns and the MeanVFE feature means follow on the caller's stream once both are done.
    host_mark("MotionNet enqueued");
    hipStream_t sm = s;   // the stream the rest of the function calls `s`: swapped for the 3D coordinate phase below
    CK(insmos_build_current_points_part(pts_host, n_pts_host, B, ld, nullptr, 0, inverse, cur_index, ncur, cur, 8, 1, s2_coords));

    // =============================== UNetV2 (3D) ===============================
    const int ncls = g.ncls;
    const int64_t Vcap = (int64_t)g.max_voxels * B;
    float* feat = A.take<float>(Vcap * 8);
    int32_t* coords1 = A.take<int32_t>(Vcap * 4);
    int32_t* num_points = A.take<int32_t>(Vcap);
    int64_t* pcid = A.take<int64_t>(ncur);
    uint64_t* ukeys = A.take<uint64_t>(ncur);
    int32_t* uperm = A.take<int32_t>(ncur);
    const size_t vox_wsb = insmos_voxelize_mean_ws_bytes(ncur);
    void* vox_ws = A.take<char>(vox_wsb);   // (kept: phase 2 reads the sorted cell keys and segment starts phase 1 leaves here)
    NEED_ARENA();
    const int64_t key_cells1 = (int64_t)g.shape[1][0] * g.shape[1][1] * g.shape[1][2];
    s = s2_coords;   // ---- from here to the join below every launch and read-back goes to the second stream
    CK(insmos_voxelize_windows_phased(cur, ncur, 8, g.in_ch, cur_start_dev, B, key_cells1, g.range, g.vs, g.max_voxels, g.max_points,
                                      feat, 8, coords1, num_points, pcid, ukeys, uperm, counts, vox_ws, vox_wsb, 1, s));
    host_mark("voxeliser enqueued");
    CK(read_counts(counts, hc, 5 + B, s));
    host_mark("voxel counts back");
    for (int b = 0; b < B; ++b) outs[b].unet_voxels[0] = hc[4 + b + 1] - hc[4 + b];
    int64_t nv[6] = {0}, nkeys[6] = {0};
    const int32_t* co[6] = {nullptr};       // a level's coordinates in the row order its features are stored in ...
    const int32_t* co_ref[6] = {nullptr};   // ... and in the order they were generated in (what the next level's bitmap is marked from)
    const uint64_t* ky[6] = {nullptr};
    const int32_t* pm[6] = {nullptr};
    nv[1] = hc[0];
    nkeys[1] = hc[1];
    co[1] = co_ref[1] = coords1;
    ky[1] = ukeys;
    pm[1] = uperm;
    // rank maps (occupancy bitmap + block prefix counts) of the five levels: what the 13 kernel maps are read off instead of
    // binary searches (coords.hip: k_build_nbr_rank); INSMOS_TABLES3D_SEARCH=1 keeps the searched builder (A/B, same tables)
    static const bool rank_tables = [] { const char* e = getenv("INSMOS_TABLES3D_SEARCH"); return !(e && e[0] == '1'); }();
    uint64_t* rbits[6] = {nullptr};
    int32_t* rincl[6] = {nullptr};
    if (rank_tables) {
        for (int l = 1; l <= 5; ++l) {
            const size_t nw = insmos_rankmap_words(g.shape[l], B);
            rbits[l] = A.take<uint64_t>(nw);
            rincl[l] = A.take<int32_t>(nw / 4);
        }
        const size_t wsb = insmos_rankmap_ws_bytes(g.shape[1], B);
        const size_t mark = A.off;
        void* ws = A.take<char>(wsb);
        NEED_ARENA();
        CK(insmos_rankmap_from_keys(ukeys, nkeys[1], g.shape[1], B, rbits[1], rincl[1], ws, wsb, s));
        A.off = mark;   // (stream-ordered: the next user of this scratch is enqueued behind it)
    }
    // row regrouping (coords.hip: k_regroup_rows): every level's rows are re-ordered (inside blocks, or over whole windows) by tap
    // signature, so that the 16 rows of a group want the same taps; the row order of a level is private to this function
    // (features, kernel maps, rank lookups and the point -> voxel map all follow).  INSMOS_REGROUP_ROWS=0 keeps the plain order.
    // (default: launch sets only -- for ONE window the four block sorts cost more latency, ~0.16 ms, than the convolutions save)
    static const int regroup_env = [] {
        const char* e = getenv("INSMOS_REGROUP_ROWS");
        const int v = e ? atoi(e) : -1;
        return (e && regroup_modes_ok(v)) ? v : -1;
    }();
    const int regroup = g_regroup >= 0 ? g_regroup : regroup_env >= 0 ? regroup_env : (B == 1 ? 0 : kRegroupDefault);
    // level l's mode = the l-th decimal digit (insmos_forward_regroup): 0 off, 1 / 2 / 3 = blocks of 256 / 1024 / 4096 rows, 4 = windows,
    // 5 = blocks of 4096 rows with the parity class above the signature
    auto regroup_level = [&](int lvl, const int32_t* c_old, int64_t nrows, const int32_t* shp, int32_t* c_new, int32_t* n2o,
                             int32_t* o2n) -> int {
        int m = regroup;
        for (int l = 1; l < lvl; ++l) m /= 10;
        m %= 10;
        const size_t wsb = insmos_regroup_ws_bytes(nrows);
        const size_t mark = A.off;
        void* ws = A.take<char>(wsb);
        NEED_ARENA();
        if (m == 4) CK(insmos_regroup_rows3d_global(c_old, nrows, rbits[lvl], shp, c_new, n2o, o2n, ws, wsb, s));
        else CK(insmos_regroup_rows3d(c_old, nrows, rbits[lvl], shp, m == 1 ? 256 : m == 2 ? 1024 : m == 3 ? 4096 : -4096, c_new, n2o, o2n,
                                      ws, wsb, s));
        A.off = mark;
        return INSMOS_OK;
    };
    auto regroup_on = [&](int lvl) {
        int m = regroup;
        for (int l = 1; l < lvl; ++l) m /= 10;
        return m % 10 != 0;
    };
    const int32_t* row_new[6] = {nullptr};    // per level: reference row -> row here, and back (null: the reference's order)
    const int32_t* row_orig[6] = {nullptr};
    const int32_t* num_points_r = num_points;   // level-1 arrays in their final row order (what phase 2 of the voxeliser reads)
    const int32_t* coords1_r = coords1;
    if (rank_tables && regroup_on(1) && nv[1] > 0) {
        int32_t* c1n = A.take<int32_t>(nv[1] * 4);
        int32_t* np1n = A.take<int32_t>(nv[1]);
        int32_t* n2o = A.take<int32_t>(nv[1]);
        int32_t* o2n = A.take<int32_t>(nv[1]);
        NEED_ARENA();
        CK(regroup_level(1, coords1, nv[1], g.shape[1], c1n, n2o, o2n));
        row_new[1] = n2o;
        row_orig[1] = o2n;
        CK(insmos_regroup_apply_voxels(n2o, nv[1], num_points, np1n, uperm, nkeys[1], pcid, ncur, s));
        co[1] = coords1_r = c1n;
        num_points_r = np1n;
    }
    auto down_coords = [&](int lvl_in, const int32_t ks[3], const int32_t st[3], const int32_t pd[3], const int32_t* oshape,
                           int lvl_out) -> int {
        const int64_t n_in = nv[lvl_in];
        const int64_t K = (int64_t)ks[0] * ks[1] * ks[2];
        const int64_t cells = (int64_t)oshape[0] * oshape[1] * oshape[2] * B;
        const int64_t cap = std::max<int64_t>(std::min(n_in * K, cells), 1);
        uint64_t* ok = A.take<uint64_t>(cap);
        int32_t* oc = A.take<int32_t>(cap * 4);
        co[lvl_out] = co_ref[lvl_out] = oc;
        ky[lvl_out] = ok;
        pm[lvl_out] = nullptr;
        if (n_in == 0) {
            nv[lvl_out] = nkeys[lvl_out] = 0;
            NEED_ARENA();
            if (rank_tables) HIP_TRY(hipMemsetAsync(rbits[lvl_out], 0, insmos_rankmap_words(oshape, B) * 8, s));
            return INSMOS_OK;
        }
        const size_t wsb = rank_tables ? insmos_rankmap_ws_bytes(oshape, B) : insmos_down_coords3d_ws_bytes_b(oshape, B);
        const size_t mark = A.off;
        void* ws = A.take<char>(wsb);
        NEED_ARENA();
        if (rank_tables)
            CK(insmos_down_coords3d_rank(co_ref[lvl_in], n_in, ks, st, pd, oshape, B, ok, oc, counts, rbits[lvl_out], rincl[lvl_out], ws,
                                         wsb, s));
        else
            CK(insmos_down_coords3d_b(co_ref[lvl_in], n_in, ks, st, pd, oshape, B, ok, oc, counts, ws, wsb, s));
        CK(read_counts(counts, hc, 1, s));
        host_mark("strided level counts back");
        A.off = mark;
        nv[lvl_out] = nkeys[lvl_out] = hc[0];
        if (rank_tables && lvl_out <= 4 && regroup_on(lvl_out) && hc[0] > 0) {   // (level 5 feeds the dense BEV scatter only)
            int32_t* ocn = A.take<int32_t>((int64_t)hc[0] * 4);
            int32_t* n2o = A.take<int32_t>(hc[0]);
            int32_t* o2n = A.take<int32_t>(hc[0]);
            NEED_ARENA();
            CK(regroup_level(lvl_out, oc, hc[0], oshape, ocn, n2o, o2n));
            co[lvl_out] = ocn;
            pm[lvl_out] = n2o;   // (rows came out in ascending key order: sorted position == old row)
            row_new[lvl_out] = n2o;
            row_orig[lvl_out] = o2n;
        }
        return INSMOS_OK;
    };
    {
        const int32_t k333[3] = {3, 3, 3}, s222[3] = {2, 2, 2}, p111[3] = {1, 1, 1};
        for (int l = 2; l <= 4; ++l) CK(down_coords(l - 1, k333, s222, p111, g.shape[l], l));
        const int32_t k311[3] = {3, 1, 1}, s211[3] = {2, 1, 1}, p000[3] = {0, 0, 0};
        CK(down_coords(4, k311, s211, p000, g.shape[5], 5));
    }
    for (int b = 0; b < B; ++b)
        for (int l = (B == 1 ? 1 : 2); l <= 5; ++l) outs[b].unet_voxels[l - 1] = nv[l];  // levels 2..5: batch totals
    const int64_t V = nv[1];
    auto build = [&](Table& t, int lvl_out, int lvl_in, const std::vector<int32_t>& delta, const int32_t* mul,
                     const int32_t* div) -> int {
        const int K = (int)(delta.size() / 4);
        t = table(K, nv[lvl_out]);
        NEED_ARENA();
        if (nv[lvl_out] == 0) return INSMOS_OK;
        if (rank_tables) {
            static const bool sparse_tab = [] {
                const char* e = getenv("INSMOS_TABLES_DENSE");
                const char* jt = getenv("INSMOS_CK_JT");
                return !(e && e[0] == '1') && !(jt && atoi(jt) > 1);
            }();
            return (sparse_tab ? insmos_build_nbr_rank_sparse : insmos_build_nbr_rank)(
                co[lvl_out], nv[lvl_out], rbits[lvl_in], rincl[lvl_in], pm[lvl_in], g.shape[lvl_in], delta.data(), K, mul, div, t.nbr,
                t.mask, s);
        }
        return insmos_build_nbr(co[lvl_out], nv[lvl_out], ky[lvl_in], pm[lvl_in], nkeys[lvl_in], 1, g.shape[lvl_in],
                                delta.data(), K, mul, div, t.nbr, t.mask, s);
    };
    const int32_t one4[4] = {1, 1, 1, 1}, two3[4] = {1, 2, 2, 2}, two1[4] = {1, 2, 1, 1};
    Table subm[5], down[5], inv[5], down5, inv5;
    // the 13 kernel maps: ONE launch over all of them (insmos_build_nbr_rank_multi; INSMOS_TABLES3D_MULTI=0: a launch per map --
    // same tables, tests/test_gpu_coords.py); the searched builder (INSMOS_TABLES3D_SEARCH=1) stays a launch per map
    static const bool multi_tab = [] { const char* e = getenv("INSMOS_TABLES3D_MULTI"); return !(e && e[0] == '0'); }();
    if (rank_tables && multi_tab) {
        static const bool sparse_tab3 = [] {
            const char* e = getenv("INSMOS_TABLES_DENSE");
            const char* jt = getenv("INSMOS_CK_JT");
            return !(e && e[0] == '1') && !(jt && atoi(jt) > 1);
        }();
        InsmosRankJob jobs[13];
        int nj = 0;
        auto add = [&](Table& t, int lvl_out, int lvl_in, const std::vector<int32_t>& delta, const int32_t* mul, const int32_t* div) {
            const int K = (int)(delta.size() / 4);
            t = table(K, nv[lvl_out]);
            InsmosRankJob& j = jobs[nj++];
            j.out_coords = co[lvl_out]; j.bits = rbits[lvl_in]; j.blk_incl = rincl[lvl_in]; j.in_perm = pm[lvl_in];
            j.nbr = t.nbr; j.mask16 = t.mask; j.in_shape = g.shape[lvl_in]; j.delta = delta.data(); j.mul = mul; j.div = div;
            j.n_out = nv[lvl_out]; j.K = K; j.reserved = 0;
        };
        for (int l = 1; l <= 4; ++l) add(subm[l], l, l, C.d_subm, one4, one4);
        for (int l = 2; l <= 4; ++l) add(down[l], l, l - 1, C.d_subm, two3, one4);
        for (int l = 2; l <= 4; ++l) add(inv[l], l - 1, l, C.d_inv, one4, two3);
        add(down5, 5, 4, C.d_down5, two1, one4);
        add(inv5, 4, 5, C.d_inv5, one4, two1);
        NEED_ARENA();
        CK(insmos_build_nbr_rank_multi(jobs, nj, sparse_tab3 ? 1 : 0, s));
    } else {
        for (int l = 1; l <= 4; ++l) CK(build(subm[l], l, l, C.d_subm, one4, one4));
        for (int l = 2; l <= 4; ++l) CK(build(down[l], l, l - 1, C.d_subm, two3, one4));
        for (int l = 2; l <= 4; ++l) CK(build(inv[l], l - 1, l, C.d_inv, one4, two3));
        CK(build(down5, 5, 4, C.d_down5, two1, one4));
        CK(build(inv5, 4, 5, C.d_inv5, one4, two1));
    }
    if ((tapc_mode & 8) && nv[2] > 0 && insmos_conv_tap_classes(27, 32, 32, 1) == 4) {
        // item lists of the level-2 SubMConv3d table for its four tap classes (csrc/spconv_tapc.hip; the table is sparse: mask16)
        subm[2].tc[1] = A.take<uint32_t>(insmos_tapc_words(27, nv[2], 4));
        subm[2].ni[1] = A.take<int32_t>((size_t)insmos_tapc_blocks(nv[2]) * 4);
        subm[2].tc_row0[1] = 0;
        NEED_ARENA();
        CK(insmos_tapc_build_masked(subm[2].nbr, subm[2].mask, 27, nv[2], 0, 4, subm[2].tc[1], subm[2].ni[1], s));
    }
    host_mark("3D kernel maps enqueued");
    // ---- join: the caller's stream (MotionNet done) fills in the motion columns, waits for the coordinate phase, and averages
    s = sm;
    CK(insmos_build_current_points_part(pts_host, n_pts_host, B, ld, motion, 4, inverse, cur_index, ncur, cur, 8, 2, s));
    CK(link_streams(s2_coords, s));
    CK(insmos_voxelize_windows_phased(cur, ncur, 8, g.in_ch, cur_start_dev, B, key_cells1, g.range, g.vs, g.max_voxels, g.max_points,
                                      feat, 8, const_cast<int32_t*>(coords1_r), const_cast<int32_t*>(num_points_r), pcid, ukeys, uperm,
                                      counts, vox_ws, vox_wsb, 2, s));

    // ---- encoder (spconv_unet.py:297-306)
    float* x0 = A.take<float>(V * 16);
    float* xc[5] = {nullptr};
    xc[1] = A.take<float>(V * 16);
    NEED_ARENA();
    CK(conv("conv_input.0", feat, V, 8, 0, &subm[1], V, x0, 16, 0, nullptr, 0, 0, 0, 0, 1));
    CK(conv("conv1.0.0", x0, V, 16, 0, &subm[1], V, xc[1], 16, 0, nullptr, 0, 0, 0, 0, 1));
    {
        const int Cs[5] = {0, 16, 32, 64, 128};
        for (int l = 2; l <= 4; ++l) {
            const int Cc = Cs[l];
            float* a = A.take<float>(nv[l] * Cc);
            float* b = A.take<float>(nv[l] * Cc);
            xc[l] = A.take<float>(nv[l] * Cc);
            NEED_ARENA();
            const std::string p = "conv" + std::to_string(l);
            CK(conv((p + ".0.0").c_str(), xc[l - 1], nv[l - 1], Cc / 2, 0, &down[l], nv[l], a, Cc, 0, nullptr, 0, 0, 0, 0, 1));
            CK(conv((p + ".1.0").c_str(), a, nv[l], Cc, 0, &subm[l], nv[l], b, Cc, 0, nullptr, 0, 0, 0, 0, 1));
            CK(conv((p + ".2.0").c_str(), b, nv[l], Cc, 0, &subm[l], nv[l], xc[l], Cc, 0, nullptr, 0, 0, 0, 0, 1));
        }
    }
    float* enc = A.take<float>(std::max<int64_t>(nv[5], 1) * 128);
    NEED_ARENA();
    CK(conv("conv_out.0", xc[4], nv[4], 128, 0, &down5, nv[5], enc, 128, 0, nullptr, 0, 0, 0, 0, 1));
    host_mark("encoder enqueued");
    CK(link_streams(s, s2_inv));   // (inv_conv_out reads `enc` only: it runs on the second stream beside the BEV head, see below)

    // ---- BEV detection head in NHWC (height_compression.py:24-31, base_bev_backbone.py:84-115)
    const int64_t nsite = (int64_t)g.bevH * g.bevW * B;  // B images stacked along the row axis
    const InsmosConvW* wb0 = Lr("bev0");
    if (!wb0 || !g.nbr_bev) return INSMOS_EINVAL;
    const int nf = wb0->cout;
    float* bev = A.take<float>(nsite * g.nbev);
    float* fa = A.take<float>(nsite * nf);
    float* fb = A.take<float>(nsite * nf);
    const int upc = g.up_ch;
    float* upf = A.take<float>(nsite * 4 * upc);  // rows [y][x], columns [ky][kx][co] == (4*nsite, upc) sub-site rows
    const int64_t ncell = 4 * nsite;
    float* head = A.take<float>(ncell * g.head_ld);
    float* cb = A.take<float>((size_t)g.pre_max * 7 * B);   // candidate / prediction arrays: (B, max, .)
    float* cs = A.take<float>((size_t)g.pre_max * B);
    int32_t* cl = A.take<int32_t>((size_t)g.pre_max * B);
    int32_t* cc = A.take<int32_t>((size_t)g.pre_max * B);
    int32_t* cnt_c = A.take<int32_t>(4 * B);
    int32_t* keep = A.take<int32_t>((size_t)g.post_max * B);
    int32_t* cnt_k = A.take<int32_t>(4 * B);
    float* pb = A.take<float>((size_t)g.post_max * 7 * B);
    float* psc = A.take<float>((size_t)g.post_max * B);
    int64_t* pl = A.take<int64_t>((size_t)g.post_max * B);
    int32_t* nbr_bev_b = B > 1 ? A.take<int32_t>((size_t)9 * nsite) : nullptr;
    NEED_ARENA();
    CK(insmos_sparse_to_bev_b(enc, 128, 128, co[5], nv[5], g.bevD, g.bevH, g.bevW, B, bev, s));
    Table tb{B > 1 ? nbr_bev_b : const_cast<int32_t*>(g.nbr_bev), nullptr, 9, nsite};
    bool tb_ready = B == 1;  // (a batch builds its stacked 9-tap table only if a layer falls back to the generic kernel)
    // the dense 3x3 layers: LDS-tiled implicit GEMM (csrc/bev.hip) for the shapes it is built for, else the 9-tap table
    static const bool bev_kernel = [] { const char* e = getenv("INSMOS_BEV_KERNEL"); return !(e && e[0] == '0'); }();
    // constant-region skipping (csrc/bev.hip, SKIP): the empty part of the map stays a per-layer constant through the stack; row
    // groups of constant sites skip their MFMAs and store the constant.  Same bits (INSMOS_BEV_SKIP=0 computes every site).
    static const bool bev_skip = [] { const char* e = getenv("INSMOS_BEV_SKIP"); return !(e && e[0] == '0'); }();
    const uint8_t* bev_dist = nullptr;
    const float* bev_cv = nullptr;
    const float* bev_chead = nullptr;
    void* bev_list_ws = nullptr;
    size_t bev_list_wsb = 0;
    // INSMOS_BEV_SKIP_LIST: 1 = the skipping layers walk compacted row-group lists (k_bev_conv3x3_list), 0 = fixed 16 x 4 patches
    // (k_bev_conv3x3<SKIP>); same bits.  (Read per call: tools flip it inside one process.)
    const bool bev_list = [&] { const char* e = getenv("INSMOS_BEV_SKIP_LIST"); return e ? e[0] != '0' : B >= kBevSkipListMinB; }();
    // (under the split-bf16 experiment, mode 3, nothing is skipped: the cached constants are the exact-fp32 kernel's, and the
    //  list / deblock skipping entry points have no reduced-precision form)
    if (bev_kernel && bev_skip && insmos::conv_precision() != 3) {
        bool all_ok = true;
        for (int k = 0; k <= g.n_bev_layers; ++k) {
            const InsmosConvW* w = Lr("bev" + std::to_string(k));
            all_ok = all_ok && w && w->K == 9 && w->cin % 16 == 0 && (w->cout == 64 || w->cout == 128);
        }
        if (all_ok) {
            CK(ensure_bev_constants(C, s, &bev_cv, &bev_chead));
            uint8_t* d = A.take<uint8_t>((size_t)nsite);
            const size_t wsb = insmos_bev_distance_map_ws_bytes(B, g.bevH, g.bevW);
            void* ws = A.take<char>(wsb);
            NEED_ARENA();
            CK(insmos_bev_distance_map(co[5], nv[5], B, g.bevH, g.bevW, g.n_bev_layers + 1, d, ws, wsb, s));
            bev_dist = d;
            bev_list_wsb = insmos_bev_skip_ws_bytes(B, g.bevH, g.bevW);
            bev_list_ws = A.take<char>(bev_list_wsb);   // (one layer's row-group lists: the layers run one after the other on this stream)
            NEED_ARENA();
        }
    }
    int bev_layer = 0;
    auto bev_conv = [&](const std::string& name, const float* x, int ld_in, float* o) -> int {
        const InsmosConvW* w = Lr(name);
        if (!w) return INSMOS_EINVAL;
        const int layer = bev_layer++;
        // (the constant region shrinks by one ring per layer: past INSMOS_BEV_SKIP_LAYERS layers the skipped work no longer pays
        //  for the uneven workgroups; on the S0 windows it pays on all six layers, profiles/r04_bev_skip_layers.txt: default = all)
        static const int skip_layers = [] { const char* e = getenv("INSMOS_BEV_SKIP_LAYERS"); return e ? atoi(e) : 99; }();
        if (bev_kernel && bev_dist && layer < skip_layers && bev_list)
            return insmos_bev_conv3x3_skip_ws(x, B, g.bevH, g.bevW, ld_in, w->cin, w->w, w->b, o, nf, w->cout, 1, bev_dist, layer,
                                              bev_cv + (size_t)layer * 128, bev_list_ws, bev_list_wsb, s);
        if (bev_kernel && bev_dist && layer < skip_layers)
            return insmos_bev_conv3x3_skip(x, B, g.bevH, g.bevW, ld_in, w->cin, w->w, w->b, o, nf, w->cout, 1, bev_dist, layer,
                                           bev_cv + (size_t)layer * 128, s);
        if (bev_kernel && w->K == 9 && w->cin % 16 == 0 && (w->cout == 64 || w->cout == 128))
            return insmos_bev_conv3x3(x, B, g.bevH, g.bevW, ld_in, w->cin, w->w, w->b, o, nf, w->cout, 1, s);
        if (!tb_ready) {
            CK(insmos_dense_nbr2d_b(g.bevH, g.bevW, B, nbr_bev_b, s));
            tb_ready = true;
        }
        return conv(name.c_str(), x, nsite, ld_in, 0, &tb, nsite, o, nf, 0, nullptr, 0, 0, 0, 0, 1);
    };
    CK(bev_conv("bev0", bev, g.nbev, fa));
    for (int k = 0; k < g.n_bev_layers; ++k) {
        CK(bev_conv("bev" + std::to_string(k + 1), fa, nf, fb));
        std::swap(fa, fb);
    }
    {
        const InsmosConvW* wd = Lr("deconv");
        const InsmosConvW* wh = Lr("head");
        if (!wd || !wh) return INSMOS_EINVAL;
        if (upc == 256 && nf % 16 == 0 && g.head_ld <= 16) {
            // the 2x2 deconv output is read by the heads only: fused, it stays in the MFMA accumulators
            // (behind a stack that ran with constant-region skipping the constant sites' result is one vector per sub-site:
            //  INSMOS_DECONV_SKIP=0 computes every site; same bits)
            static const bool dskip = [] { const char* e = getenv("INSMOS_DECONV_SKIP"); return !(e && e[0] == '0'); }();
            static const int skip_layers = [] { const char* e = getenv("INSMOS_BEV_SKIP_LAYERS"); return e ? atoi(e) : 99; }();
            if (dskip && bev_dist && bev_chead && skip_layers > g.n_bev_layers)
                CK(insmos_deconv_head_skip(fa, nsite, nf, nf, wd->w, wd->b, upc, wh->w, wh->b, g.head_ld, head, g.head_ld, bev_dist, g.bevH,
                                           g.bevW, g.n_bev_layers + 1, bev_chead, s));
            else
                CK(insmos_deconv_head(fa, nsite, nf, nf, wd->w, wd->b, upc, wh->w, wh->b, g.head_ld, head, g.head_ld, s));
        } else {
            CK(conv("deconv", fa, nsite, nf, 0, nullptr, nsite, upf, 4 * upc, 0, nullptr, 0, 0, 0, 0, 1));
            CK(conv("head", upf, ncell, upc, 0, nullptr, ncell, head, g.head_ld, 0, nullptr, 0, 0, 0, 0, 0));  // upf as (4*nsite, upc)
        }
    }
    {
        const size_t wsb = insmos_center_decode_select_ws_bytes(ncell);
        const size_t mark = A.off;
        void* ws = A.take<char>(wsb);
        NEED_ARENA();
        CK(insmos_center_decode_select_b(head, g.head_ld, ncls, 2 * g.bevH, 2 * g.bevW, 2, B, g.out_factor, g.tvs[0], g.tvs[1],
                                         g.range[0], g.range[1], g.score_thresh, g.pre_max, cb, cs, cl, cc, cnt_c, ws, wsb, s));
        // (no rewind of the decode scratch: what is allocated below is written on the SECOND stream -- inv_conv_out into ci4 --
        //  while the decode kernels may still be running on this one; a region is only handed on inside one stream's order)
        (void)mark;
        const size_t wsn = insmos_nms_ws_bytes_b(g.pre_max, B);
        ws = A.take<char>(wsn);
        NEED_ARENA();
        CK(insmos_nms_rotated_bev_b(cb, cnt_c, g.pre_max, g.nms_thresh, g.post_max, B, keep, cnt_k, ws, wsn, s));
        // (the NMS workspace stays allocated: kernels below are stream-ordered after it, but keep it simple)
    }
    CK(insmos_gather_preds_b(cb, cs, cl, keep, cnt_k, g.pre_max, g.post_max, B, pb, psc, pl, s));
    host_mark("BEV head + NMS enqueued");

    // ---- upsample fusion (spconv_unet.py:319-402)
    int64_t nvmax = 0;
    for (int l = 1; l <= 5; ++l) nvmax = std::max(nvmax, nv[l]);
    int32_t* scratch = A.take<int32_t>(insmos_boxes_to_onehot_scratch_ints_b(g.post_max, B, nvmax));
    auto onehot = [&](int level, float mult, float* o, int ldo, int col) -> int {
        if (nv[level] == 0) return INSMOS_OK;
        // ("first voxel inside a box" is order-dependent in the reference: it is taken in the reference's row order)
        return insmos_boxes_to_onehot_rows(pb, pl, cnt_k, g.post_max, B, g.range, g.vs, 8.0f, mult, co[level], row_orig[level],
                                           row_new[level], nv[level], ncls, 16, g.quirk_exact, o + col, ldo, scratch, s);
    };
    // UR_block_forward up to (not including) conv_inv; catm[:, 0:C] already holds x_bottom
    auto ur_block = [&](int lvl, int Cc, const float* x_lat, int ld_lat, float* catm, float* m) -> int {
        float* t = A.take<float>(nv[lvl] * Cc);
        NEED_ARENA();
        const std::string tn = "conv_up_t" + std::to_string(lvl), mn = "conv_up_m" + std::to_string(lvl) + ".0";
        CK(conv((tn + ".conv1").c_str(), x_lat, nv[lvl], ld_lat, 0, &subm[lvl], nv[lvl], t, Cc, 0, nullptr, 0, 0, 0, 0, 1));
        CK(conv((tn + ".conv2").c_str(), t, nv[lvl], Cc, 0, &subm[lvl], nv[lvl], catm, 2 * Cc, Cc, x_lat, ld_lat, 0, 1, 0, 1));
        CK(conv(mn.c_str(), catm, nv[lvl], 2 * Cc, 0, &subm[lvl], nv[lvl], m, Cc, 0, catm, 2 * Cc, 0, 2, 1, 0));
        return INSMOS_OK;
    };
    float* ci4 = A.take<float>(nv[4] * 144);
    float* catm4 = A.take<float>(nv[4] * 256);
    float* m4 = A.take<float>(nv[4] * 128);
    float* ci3 = A.take<float>(nv[3] * 80);
    float* catm3 = A.take<float>(nv[3] * 128);
    float* m3 = A.take<float>(nv[3] * 64);
    float* ci2 = A.take<float>(nv[2] * 48);
    float* catm2 = A.take<float>(nv[2] * 64);
    float* m2 = A.take<float>(nv[2] * 32);
    float* ci1 = A.take<float>(V * 32);
    float* catm1 = A.take<float>(V * 32);
    float* m1 = A.take<float>(V * 16);
    float* ci0 = A.take<float>(V * 32);
    float* seg = A.take<float>(V * 16);
    float* vox_logits = A.take<float>(V * 4);
    float* logits = A.take<float>(ncur * 3);
    NEED_ARENA();
    CK(conv("inv_conv_out", enc, nv[5], 128, 0, &inv5, nv[4], ci4, 144, 0, nullptr, 0, 0, 0, 0, 0, 0, s2_inv));
    hipEvent_t ev_inv = nullptr;
    if (s2_inv != s) {
        CK(tl_aux.next(&ev_inv));
        HIP_TRY(hipEventRecord(ev_inv, s2_inv));
    }
    CK(onehot(4, 1.0f, ci4, 144, 128));
    // the finer levels' one-hot columns are not needed before their decoder level: second stream (behind onehot(4): the passes
    // share one scratch array), each consumer waits for its own
    hipEvent_t ev_oh[4] = {nullptr, nullptr, nullptr, nullptr};
    CK(link_streams(s, s2_oh));
    {
        s = s2_oh;
        const float mult[4] = {0.f, 8.0f, 4.0f, 2.0f};
        float* const dst[4] = {nullptr, ci1, ci2, ci3};
        const int ldo[4] = {0, 32, 48, 80}, col[4] = {0, 16, 32, 64};
        int rc = INSMOS_OK;
        for (int l = 3; l >= 1 && rc == INSMOS_OK; --l) {
            rc = onehot(l, mult[l], dst[l], ldo[l], col[l]);
            if (rc == INSMOS_OK && s2_oh != sm) {
                rc = tl_aux.next(&ev_oh[l]);
                if (rc == INSMOS_OK && hipEventRecord(ev_oh[l], s2_oh) != hipSuccess) rc = INSMOS_EHIP;
            }
        }
        s = sm;
        CK(rc);
    }
    if (ev_inv) HIP_TRY(hipStreamWaitEvent(s, ev_inv, 0));
    CK(conv("conv_up_instance_block.0", ci4, nv[4], 144, 0, &subm[4], nv[4], catm4, 256, 0, nullptr, 0, 0, 0, 0, 1));
    CK(ur_block(4, 128, catm4, 256, catm4, m4));
    CK(conv("inv_conv4.0", m4, nv[4], 128, 0, &inv[4], nv[3], ci3, 80, 0, nullptr, 0, 0, 0, 0, 1));
    if (ev_oh[3]) HIP_TRY(hipStreamWaitEvent(s, ev_oh[3], 0));
    CK(conv("conv_up_instance_block_up4.0", ci3, nv[3], 80, 0, &subm[3], nv[3], catm3, 128, 0, nullptr, 0, 0, 0, 0, 1));
    CK(ur_block(3, 64, xc[3], 64, catm3, m3));
    CK(conv("inv_conv3.0", m3, nv[3], 64, 0, &inv[3], nv[2], ci2, 48, 0, nullptr, 0, 0, 0, 0, 1));
    if (ev_oh[2]) HIP_TRY(hipStreamWaitEvent(s, ev_oh[2], 0));
    CK(conv("conv_up_instance_block_up3.0", ci2, nv[2], 48, 0, &subm[2], nv[2], catm2, 64, 0, nullptr, 0, 0, 0, 0, 1));
    CK(ur_block(2, 32, xc[2], 32, catm2, m2));
    CK(conv("inv_conv2.0", m2, nv[2], 32, 0, &inv[2], V, ci1, 32, 0, nullptr, 0, 0, 0, 0, 1));
    if (ev_oh[1]) HIP_TRY(hipStreamWaitEvent(s, ev_oh[1], 0));
    CK(conv("conv_up_instance_block_up2.0", ci1, V, 32, 0, &subm[1], V, catm1, 32, 0, nullptr, 0, 0, 0, 0, 1));
    CK(ur_block(1, 16, xc[1], 16, catm1, m1));
    CK(conv("conv_up_out.0.0", m1, V, 16, 0, &subm[1], V, ci0, 32, 0, nullptr, 0, 0, 0, 0, 1));
    // spconv_unet.py:401 re-uses the stride-1 instance features: same one-hots, copied instead of recomputed
    CK(insmos_copy_cols(ci1 + 16, 32, ci0 + 16, 32, V, 16, s));
    CK(conv("conv_up_instance_block_up1.0", ci0, V, 32, 0, &subm[1], V, seg, 16, 0, nullptr, 0, 0, 0, 0, 1));
    CK(conv("mos_seg", seg, V, 16, 0, nullptr, V, vox_logits, 4, 0, nullptr, 0, 0, 0, 0, 0));
    CK(insmos_gather_rows(vox_logits, 4, 3, pcid, ncur, logits, 3, s));
    // the one unavoidable read-back: the caller's output tensors are sized by the box counts
    {
        int32_t hk[4 * INSMOS_MAX_BATCH], hcand[4 * INSMOS_MAX_BATCH];
        int32_t* pin = tl_pinned.get();   // (2 x 4 B ints <= 4096 bytes: B <= 16)
        int32_t* hk_dst = pin ? pin : hk;
        int32_t* hc_dst = pin ? pin + 4 * INSMOS_MAX_BATCH : hcand;
        HIP_TRY(hipMemcpyAsync(hk_dst, cnt_k, (size_t)B * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(hc_dst, cnt_c, (size_t)B * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        host_mark("decoder enqueued");
        CK(wait_stream(s));
        host_mark("box counts back (end)");
        if (pin) { memcpy(hk, hk_dst, (size_t)B * 16); memcpy(hcand, hc_dst, (size_t)B * 16); }
        for (int b = 0; b < B; ++b) {
            outs[b].n_boxes = hk[4 * b];
            outs[b].n_candidates = hcand[4 * b];
        }
    }
    for (int b = 0; b < B; ++b) {
        InsmosForwardOut& o = outs[b];
        o.logits_off = (int64_t)((char*)(logits + cur_start[b] * 3) - (char*)arena);
        o.boxes_off = (int64_t)((char*)(pb + (size_t)b * g.post_max * 7) - (char*)arena);
        o.scores_off = (int64_t)((char*)(psc + (size_t)b * g.post_max) - (char*)arena);
        o.labels_off = (int64_t)((char*)(pl + (size_t)b * g.post_max) - (char*)arena);
        o.cur_points_off = (int64_t)((char*)(cur + cur_start[b] * 8) - (char*)arena);
        o.arena_needed = (int64_t)A.off;
    }
    return INSMOS_OK;
}

extern "C" int insmos_forward_windows(void* ctx, const float* const* points_host, const int64_t* n_points_host, int B,
                                      int ld_pts, void* arena, size_t arena_bytes, void* stream, InsmosForwardOut* outs) {
    return forward_windows_impl(ctx, points_host, n_points_host, B, ld_pts, arena, arena_bytes, stream, outs);
}

extern "C" int insmos_forward_window(void* ctx, const float* pts, int64_t N, int ld, void* arena, size_t arena_bytes,
                                     void* stream, InsmosForwardOut* out) {
    return forward_windows_impl(ctx, &pts, &N, 1, ld, arena, arena_bytes, stream, out);
}

// Which pieces of the calling host thread's next forwards go to its second stream (bits as INSMOS_TWO_STREAMS, which wins when
// set; -1 = default 15).  A caller that keeps several launch sets in flight turns it off: the sets overlap each other already.
extern "C" int insmos_forward_streams(int mask) {
    if (mask < -1 || mask > 15) return INSMOS_EINVAL;
    tl_stream_mask = mask;
    return INSMOS_OK;
}

// row regrouping of the 3D levels (coords.hip: insmos_regroup_rows3d[_global]), one decimal digit per level 4..1: 0 = off, 1 / 2 / 3 =
// blocks of 256 / 1024 / 4096 rows, 4 = whole windows, 5 = 4096-row blocks sorted by parity class first; -1 = back to the default
// (INSMOS_REGROUP_ROWS, else kRegroupDefault for launch sets of two windows or more and off for a single window).
// Process-wide; the outputs do not depend on it (tests/test_gpu_model.py).
// The calling host thread's second stream and its events are released (a worker thread calls this before it ends; a thread that
// moves to another device gets new ones on its own, see Aux::init).  Not done from a thread_local destructor: at process exit that
// would run after the HIP runtime's own teardown.
// The calling host thread's last forward as text, "stage:microseconds;..." (INSMOS_HOST_MARKS=1 in the environment, else empty).
extern "C" int insmos_forward_host_marks(char* buf, size_t cap) {
    if (!buf || cap == 0) return INSMOS_EINVAL;
    size_t off = 0;
    buf[0] = 0;
    for (int i = 0; i < tl_marks.n; ++i) {
        const int w = snprintf(buf + off, cap - off, "%s:%.1f;", tl_marks.name[i], tl_marks.us[i]);
        if (w < 0 || (size_t)w >= cap - off) return INSMOS_EWORKSPACE;
        off += (size_t)w;
    }
    return INSMOS_OK;
}

extern "C" int insmos_forward_thread_release(void) {
    if (tl_aux.s2) (void)hipStreamSynchronize(tl_aux.s2);
    tl_aux.release();
    return INSMOS_OK;
}

extern "C" int insmos_forward_regroup(int modes) {
    if (modes != -1 && !regroup_modes_ok(modes)) return INSMOS_EINVAL;
    g_regroup = modes;
    return INSMOS_OK;
}

extern "C" int insmos_debug_table_limit(int64_t bytes) {
    g_table_limit = (bytes > 0 && bytes < (1ll << 31)) ? bytes : (1ll << 31);
    return INSMOS_OK;
}
