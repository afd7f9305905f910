// capi.hip — error channel and device queries of the C-ABI (include/mit_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "../../include/mit_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";

int mit_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char *mit_last_error(void) { return g_err; }
extern "C" int mit_abi_version(void) { return MIT_ABI_VERSION; }

namespace {
std::atomic<int> g_cotenant_safe{-1};  // -1: not read yet
}
bool mit_cotenant_safe() {
    int v = g_cotenant_safe.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("MIT_COTENANT_SAFE");
        v = (e && *e && *e != '0') ? 1 : 0;
        g_cotenant_safe.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int mit_cotenant_safe_set(int on) {
    const int prev = mit_cotenant_safe() ? 1 : 0;
    if (on >= 0) g_cotenant_safe.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

#ifndef MIT_SOURCE_DIGEST
#define MIT_SOURCE_DIGEST "unknown"
#endif
// the marker makes the digest readable from the file without dlopen()ing a possibly stale binary (lib.py:_stale)
static const char kDigestMarker[] = "MIT_SOURCE_DIGEST=" MIT_SOURCE_DIGEST;
extern "C" const char *mit_source_digest(void) { return kDigestMarker + sizeof("MIT_SOURCE_DIGEST=") - 1; }

// ---- generic kernel-time probe -------------------------------------------------------------------------------------
namespace {
struct KRec {
    const char *name;
    hipEvent_t start, stop;
    double bytes, flops;
    bool closed;
};
std::mutex g_kmu;
bool g_kon = false;
std::vector<KRec> g_krecs;
}  // namespace

bool mit_probe_on() { return g_kon; }

void mit_probe_reset(bool on) {
    std::lock_guard<std::mutex> lk(g_kmu);
    for (auto &r : g_krecs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_krecs.clear();
    g_kon = on;
}

MitProbeScope::MitProbeScope(const char *name, hipStream_t s, double alg_bytes, double alg_flops) : idx_(-1), s_(s) {
    if (!g_kon) return;
    std::lock_guard<std::mutex> lk(g_kmu);
    KRec r{name, nullptr, nullptr, alg_bytes, alg_flops, false};
    if (hipEventCreate(&r.start) != hipSuccess) return;
    if (hipEventCreate(&r.stop) != hipSuccess) {
        (void)hipEventDestroy(r.start);
        return;
    }
    (void)hipEventRecord(r.start, s);
    idx_ = (int)g_krecs.size();
    g_krecs.push_back(r);
}

MitProbeScope::~MitProbeScope() {
    if (idx_ < 0) return;
    std::lock_guard<std::mutex> lk(g_kmu);
    if (idx_ < (int)g_krecs.size()) {
        (void)hipEventRecord(g_krecs[idx_].stop, s_);
        g_krecs[idx_].closed = true;
    }
}

extern "C" int mit_prof_kernels_read(MitProfKernelStat *stats, int max_stats, int *n_stats) {
    if (!stats || !n_stats || max_stats <= 0) return mit_set_error("mit_prof_kernels_read: bad arguments");
    std::lock_guard<std::mutex> lk(g_kmu);
    int n = 0;
    for (auto &r : g_krecs) {
        if (!r.closed) continue;
        MIT_CHECK_HIP(hipEventSynchronize(r.stop));
        float ms = 0.f;
        MIT_CHECK_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        int k = 0;
        for (; k < n; ++k)
            if (!strcmp(stats[k].name, r.name)) break;
        if (k == n) {
            if (n == max_stats) continue;  // more distinct kernels than the caller has room for: the rest is dropped
            memset(&stats[k], 0, sizeof(stats[k]));
            strncpy(stats[k].name, r.name, sizeof(stats[k].name) - 1);
            ++n;
        }
        stats[k].launches += 1;
        stats[k].ms += ms;
        stats[k].alg_bytes += r.bytes;
        stats[k].alg_flops += r.flops;
    }
    *n_stats = n;
    return 0;
}

extern "C" int mit_device_count(int *count) {
    if (!count) return mit_set_error("mit_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return mit_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

extern "C" int mit_device_name(int device, char *buf, int buflen) {
    if (!buf || buflen <= 0) return mit_set_error("mit_device_name: bad buffer");
    hipDeviceProp_t prop;
    MIT_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}
