// Per-launch timing of instrumented kernels for bench.py's roofline lines: HIP events recorded on the launch stream
// around each launch while enabled (devqa_profile).  Measurement hook, not part of the data path; thread-safe (the MEND
// training prefetch thread and the main thread may both launch instrumented kernels).
#include <mutex>
#include <vector>
#include "common.h"

namespace {
constexpr size_t PROF_MAX_PAIRS = 98304;
std::mutex g_mu;
bool g_on = false;
std::vector<hipEvent_t> g_ev;     // pairs: start, stop
std::vector<int> g_slot;
std::vector<double> g_work;
size_t g_used = 0;
int64_t g_dropped = 0;        // launches that found the event pool full since devqa_profile(1): reported, never silent
}  // namespace

int devqa_prof_begin(int slot, hipStream_t st) {
    if (!g_on) return -1;           // unsynchronised fast path: a launch racing with devqa_profile(1) is simply not recorded
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_on || slot < 0 || slot >= DEVQA_PROF_SLOTS) return -1;
    if (g_used >= PROF_MAX_PAIRS) { ++g_dropped; return -1; }
    const int idx = (int)g_used++;
    g_slot[idx] = slot;
    g_work[idx] = -1.0;             // closed by devqa_prof_end
    (void)hipEventRecord(g_ev[2 * idx], st);
    return idx;
}

void devqa_prof_end(int idx, double work, hipStream_t st) {
    if (idx < 0) return;
    (void)hipEventRecord(g_ev[2 * idx + 1], st);
    std::lock_guard<std::mutex> lock(g_mu);
    g_work[idx] = work;
}

extern "C" int devqa_profile(int enable) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (enable) {
        if (g_ev.empty()) {
            g_ev.resize(2 * PROF_MAX_PAIRS);
            for (auto& e : g_ev)
                if (hipEventCreate(&e) != hipSuccess) { g_ev.clear(); return devqa_fail(DEVQA_E_HIP, "profile: hipEventCreate failed"); }
            g_slot.resize(PROF_MAX_PAIRS);
            g_work.resize(PROF_MAX_PAIRS);
        }
        if (enable != 2) {      // 2 = resume after a pause: the records taken so far stay (bench.py instruments a sample of its timed steps)
            g_used = 0;
            g_dropped = 0;
        }
    }
    g_on = enable != 0;
    return DEVQA_OK;
}

extern "C" int devqa_profile_read(int slot, double* ms, double* work, int64_t* launches) {
    DEVQA_CHECK_ARG(ms && work && launches && slot >= 0 && slot < DEVQA_PROF_SLOTS, "profile_read: bad argument");
    std::lock_guard<std::mutex> lock(g_mu);
    *ms = 0.0; *work = 0.0; *launches = 0;
    for (size_t i = 0; i < g_used; ++i) {
        if (g_slot[i] != slot || g_work[i] < 0.0) continue;
        float t = 0.f;
        if (hipEventSynchronize(g_ev[2 * i + 1]) != hipSuccess) return devqa_fail(DEVQA_E_HIP, "profile: event sync");
        if (hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess) return devqa_fail(DEVQA_E_HIP, "profile: elapsed");
        *ms += t; *work += g_work[i]; *launches += 1;
    }
    return DEVQA_OK;
}

extern "C" int devqa_profile_dropped(int64_t* dropped) {
    DEVQA_CHECK_ARG(dropped, "profile_dropped: bad argument");
    std::lock_guard<std::mutex> lock(g_mu);
    *dropped = g_dropped;
    return DEVQA_OK;
}

// r01 interface kept: slots 0..3 are the GEMM tile variants
extern "C" int devqa_profile_gemm(int enable) { return devqa_profile(enable); }
extern "C" int devqa_profile_gemm_read(double* ms, double* flops, int64_t* launches) {
    for (int v = 0; v < 4; ++v) {
        const int rc = devqa_profile_read(v, &ms[v], &flops[v], &launches[v]);
        if (rc != DEVQA_OK) return rc;
    }
    return DEVQA_OK;
}
