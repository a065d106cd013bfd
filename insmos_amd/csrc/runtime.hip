// insmos_amd/csrc/runtime.hip -- version / error / profiler plumbing of libinsmos_hip.so.
#include <mutex>
#include <vector>
#include "common.h"

namespace insmos {
int g_last_hip_error = 0;

static bool g_prof = false;
struct Span { int kind; hipEvent_t e0, e1; };
static std::vector<Span> g_spans;
static std::mutex g_mu;
static const char* kNames[KK_COUNT] = {
    "quant_keys", "radix_sort", "scan", "quant_scatter", "level_down", "build_nbr", "vox_keys", "vox_segments",
    "vox_mean", "down_candidates", "down_unique", "sparse_conv_mfma", "dense_nbr2d", "sparse_to_bev", "center_decode",
    "select_topk", "nms_mask", "nms_reduce", "iou_bev", "gather_preds", "boxes_to_onehot", "gather_rows",
    "current_points", "fill_cols", "confusion", "memset"};

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
    g_spans.push_back({kind, e0, e1});
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
