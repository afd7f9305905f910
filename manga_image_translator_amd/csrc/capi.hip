// capi.hip — error channel and device queries of the C-ABI (include/mit_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
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
