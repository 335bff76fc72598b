// Live per-kernel timing for bench.py's roofline leg: a pair of HIP events recorded on the launch stream around
// every conv launch while profiling is enabled; wdm_prof_report() synchronises and aggregates per kernel name.
#include <string.h>

#include <map>
#include <string>

#include <atomic>
#include "common.h"

namespace wdm {
namespace {
struct Rec { hipEvent_t e0, e1; std::string name; double flops, bytes; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
bool prof_enabled() { return g_on; }
void prof_begin(hipStream_t s, const char* kernel, double flops, double bytes) {
    Rec r{get_event(), get_event(), kernel, flops, bytes};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().e1, s); }
static std::atomic<int> g_concurrent_streams{1};
int concurrent_streams() { return g_concurrent_streams.load(std::memory_order_relaxed); }
}  // namespace wdm

using namespace wdm;

extern "C" {

int wdm_prof_enable(int on) {
    g_on = on != 0;
    return WDM_OK;
}

// Re-reads the WDM_* experiment switches (common.h: EnvCfg); they are otherwise read once, at first use.
int wdm_env_refresh(void) {
    env_cfg();                // make sure the first-use initialisation has happened, then overwrite it
    env_cfg_refresh();
    return WDM_OK;
}

// Launches the caller keeps in flight side by side (include/wavedm.h); read by the workgroup-count rules of conv_dispatch.inc (concurrent_streams())
int wdm_set_concurrent_streams(int n) {
    if (n < 1 || n > 64) WDM_FAIL(WDM_EINVAL, "wdm_set_concurrent_streams: %d out of range (1 .. 64)", n);
    g_concurrent_streams.store(n, std::memory_order_relaxed);
    return WDM_OK;
}

// Aggregates and clears the recorded launches.  out[i] rows: launches, total_ms, total_flops, total_bytes.
int wdm_prof_report(wdm_prof_entry* out, int max_entries, int* n_entries) {
    if (!out || !n_entries) WDM_FAIL(WDM_EINVAL, "wdm_prof_report: null argument");
    std::map<std::string, wdm_prof_entry> agg;
    for (auto& r : g_recs) {
        WDM_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        WDM_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        wdm_prof_entry& e = agg[r.name];
        if (e.launches == 0) { memset(&e, 0, sizeof(e)); strncpy(e.kernel, r.name.c_str(), sizeof(e.kernel) - 1); }
        e.launches += 1; e.total_ms += ms; e.total_flops += r.flops; e.total_bytes += r.bytes;
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
    int n = 0;
    for (auto& kv : agg) { if (n < max_entries) out[n++] = kv.second; }
    *n_entries = n;
    return WDM_OK;
}

}  // extern "C"
