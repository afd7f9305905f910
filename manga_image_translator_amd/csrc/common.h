// common.h — internal helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
