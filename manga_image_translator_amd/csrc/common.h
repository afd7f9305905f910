// common.h — internal helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>

// Records a thread-local error string (returned by mit_last_error) and returns 1.
int mit_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define MIT_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return mit_set_error("%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define MIT_CHECK_LAUNCH(what)                                                                 \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess) return mit_set_error("%s: launch failed: %s", what, hipGetErrorString(_e)); \
    } while (0)

static inline int mit_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Dynamic LDS beyond 64 KB needs an opt-in per kernel and per device (hipFuncAttributeMaxDynamicSharedMemorySize).  One static
// instance per kernel; ensure() raises the grant whenever a launch needs more than what was granted before (a layer with more taps
// needs a larger gather table than the first launch of that kernel did).  Thread-safe and monotonic: the engines launch from several
// host threads, and a smaller grant must never land between another thread's larger grant and its launch.
struct DynSmemOptIn {
    std::atomic<size_t> granted[16];
    std::mutex mu;
    DynSmemOptIn() {
        for (auto &g : granted) g.store(0, std::memory_order_relaxed);
    }
    void ensure(const void *kern, size_t smem) {
        if (smem <= 64 * 1024) return;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t> &g = granted[dev & 15];
        if (g.load(std::memory_order_acquire) >= smem) return;
        std::lock_guard<std::mutex> lk(mu);
        if (g.load(std::memory_order_relaxed) >= smem) return;
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        g.store(smem, std::memory_order_release);
    }
};

// MIT_COTENANT_SAFE (mit_cotenant_safe_set): the pre-fix mitigation of the co-tenancy failure — the FFT rows kernels take a
// whole CU's LDS so that no other queue's kernel can be co-resident.  Not needed since the library is built without SLP-packed fp32
// instructions; off by default, kept as a switch.
bool mit_cotenant_safe();

// ---- generic kernel-time probe (mit_prof_kernels_read): while mit_prof_enable(1) is in force, a MitProbeScope around a
// launch brackets it with HIP events on its stream and files it under `name` with the caller's algorithmic bytes / FLOPs.
bool mit_probe_on();
void mit_probe_reset(bool on);
struct MitProbeScope {
    MitProbeScope(const char *name, hipStream_t s, double alg_bytes, double alg_flops = 0.0);
    ~MitProbeScope();
    MitProbeScope(const MitProbeScope &) = delete;
    MitProbeScope &operator=(const MitProbeScope &) = delete;

   private:
    int idx_;
    hipStream_t s_;
};
