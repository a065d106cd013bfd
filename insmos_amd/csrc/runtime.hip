// insmos_amd/csrc/runtime.hip -- version / error / profiler plumbing of libinsmos_hip.so.
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>
#include "common.h"

namespace insmos {
int g_last_hip_error = 0;

static bool g_prof = false;
struct Span { int kind; hipEvent_t e0, e1; int64_t meta[4]; };
static std::vector<Span> g_spans;
static std::mutex g_mu;
static const char* kNames[KK_COUNT] = {
    "quant_keys", "radix_sort", "scan", "quant_scatter", "level_down", "build_nbr", "vox_keys", "vox_segments",
    "vox_mean", "down_candidates", "down_unique", "sparse_conv_mfma", "dense_nbr2d", "sparse_to_bev", "center_decode",
    "select_topk", "nms_mask", "nms_reduce", "iou_bev", "gather_preds", "boxes_to_onehot", "gather_rows",
    "current_points", "fill_cols", "confusion", "memset", "batchnorm"};

bool prof_on() { return g_prof; }

ProfScope::ProfScope(int k, hipStream_t st) : kind(k), s(st) {
    if (!g_prof) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, s);
}
ProfScope::~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_spans.push_back({kind, e0, e1, {meta[0], meta[1], meta[2], meta[3]}});
}
}  // namespace insmos

using namespace insmos;

extern "C" int insmos_version(void) { return 100; }
extern "C" int insmos_last_hip_error(void) { return g_last_hip_error; }
extern "C" int insmos_prof_enable(int on) { g_prof = on != 0; return INSMOS_OK; }
extern "C" int insmos_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& sp : g_spans) { (void)hipEventDestroy(sp.e0); (void)hipEventDestroy(sp.e1); }
    g_spans.clear();
    return INSMOS_OK;
}
extern "C" const char* insmos_prof_name(int kind_id) {
    return (kind_id >= 0 && kind_id < KK_COUNT) ? kNames[kind_id] : "?";
}
extern "C" int insmos_prof_read(int max, int* kind_ids_host, double* total_ms_host, int64_t* launches_host) {
    HIP_TRY(hipDeviceSynchronize());
    double tot[KK_COUNT] = {0};
    int64_t cnt[KK_COUNT] = {0};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& sp : g_spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess) { tot[sp.kind] += ms; cnt[sp.kind] += 1; }
        }
    }
    int n = 0;
    for (int k = 0; k < KK_COUNT && n < max; ++k) {
        if (!cnt[k]) continue;
        kind_ids_host[n] = k; total_ms_host[n] = tot[k]; launches_host[n] = cnt[k]; ++n;
    }
    return n;
}

// Per-launch durations of one kernel kind in record order, with the four integers the launch site attached (the conv
// launcher records K, Cin, Cout and the rows computed): tools/batch_layers.py lines them up with the layer list.
extern "C" int insmos_prof_read_spans(int kind_id, int max, double* ms_host, int64_t* meta4_host) {
    if (!ms_host || !meta4_host || max <= 0) return INSMOS_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto& sp : g_spans) {
        if (sp.kind != kind_id || n >= max) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.e0, sp.e1) != hipSuccess) continue;
        ms_host[n] = ms;
        for (int i = 0; i < 4; ++i) meta4_host[4 * n + i] = sp.meta[i];
        ++n;
    }
    return n;
}

// Busy time of one kernel kind when launches on several streams overlap: the length of the UNION of the launches'
// [start, end] event intervals (same device clock), next to the plain sum of their durations.
extern "C" int insmos_prof_read_union(int kind_id, double* union_ms_host, double* sum_ms_host, int64_t* launches_host) {
    if (!union_ms_host || !sum_ms_host || !launches_host) return INSMOS_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<std::pair<double, double>> iv;
    double sum = 0.0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        hipEvent_t base = nullptr;
        for (auto& sp : g_spans) {
            if (!base) base = sp.e0;
            if (sp.kind != kind_id) continue;
            float a = 0.f, d = 0.f;
            if (hipEventElapsedTime(&a, base, sp.e0) != hipSuccess || hipEventElapsedTime(&d, sp.e0, sp.e1) != hipSuccess) continue;
            iv.emplace_back((double)a, (double)a + (double)d);
            sum += d;
        }
    }
    std::sort(iv.begin(), iv.end());
    double uni = 0.0, lo = 0.0, hi = -1.0;
    for (auto& p : iv) {
        if (hi < 0.0) { lo = p.first; hi = p.second; }
        else if (p.first <= hi) hi = std::max(hi, p.second);
        else { uni += hi - lo; lo = p.first; hi = p.second; }
    }
    if (hi >= 0.0) uni += hi - lo;
    *union_ms_host = uni;
    *sum_ms_host = sum;
    *launches_host = (int64_t)iv.size();
    return INSMOS_OK;
}
