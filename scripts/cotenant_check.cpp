// cotenant_check.cpp — which property of a co-resident kernel disturbs other queues' kernels on MI355X?  (profiles/README r03v-x)
//
// Round 3 found that the 128 x 128 split-bf16 GEMM tile (168 VGPRs, 49.6 KB LDS, bf16 MFMA), looping in ANOTHER PROCESS, makes
// mit_rfft_rows launches of this process produce wrong workgroups, while the same arithmetic on the 128 x 64 tile (120 VGPRs, 37 KB)
// does not.  This program separates the candidates with synthetic co-tenants:
//     child process : loops ONE aggressor kernel  — registers only (NV live VGPRs), LDS only (BYTES of dynamic LDS), MFMA only, or mixes
//     parent process: a quiet reference of the victim first, then the victim repeated while the child runs; launches that differ are counted
// victims: mit_rfft_rows of libmit_hip.so (the kernel seen failing) and a synthetic LDS-exchange kernel (barrier-separated transposes).
// Torch-free; build with scripts/build_cotenant_check.sh, run as `scripts/cotenant_check [launches per aggressor, default 200]`.
#include <hip/hip_runtime.h>
#include <math.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
#include "mit_hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- aggressors -------------------------------------------------------------------------------------------------------------
// NV values per lane stay live in VGPRs across a long dependent loop (the asm statements pin them); optional dynamic LDS traffic with
// barriers; optional bf16 MFMAs.  MINW sets the register budget through the launch bounds (256 threads: 1 -> 512, 2 -> 256, 3 -> 168).
template <int NV, int MINW, bool USE_LDS, bool USE_MFMA>
__global__ __launch_bounds__(256, MINW) void hog_kernel(float *out, int iters, int lds_floats) {
    extern __shared__ float lds[];
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = (float)(threadIdx.x + i);
    f32x16 acc = {0};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)1.0f, b[i] = (__bf16)0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i] = r[i] * 1.0001f + 0.5f;
            asm volatile("" : "+v"(r[i]));
        }
        if (USE_LDS) {
            for (int j = threadIdx.x; j < lds_floats; j += 256) lds[j] = r[0] + (float)(it + j);
            __syncthreads();
            float s = 0.f;
            for (int j = threadIdx.x; j < lds_floats; j += 256) s += lds[lds_floats - 1 - j];
            r[0] += s * 1e-9f;
            __syncthreads();
        }
        if (USE_MFMA) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    float s = acc[0];
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

struct Aggressor {
    const char *name;
    void (*launch)(float *, hipStream_t);
};

template <int NV, int MINW, bool L, bool M>
void launch_hog(float *out, hipStream_t st, int lds_bytes, int wgs) {
    auto k = hog_kernel<NV, MINW, L, M>;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        set = true;
    }
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), (size_t)lds_bytes, st, out, 400, lds_bytes / 4);
}

static const Aggressor kAggressors[] = {
    {"idle (no co-tenant)", nullptr},
    {"registers only: 40 live VGPRs, no LDS", [](float *o, hipStream_t s) { launch_hog<32, 4, false, false>(o, s, 0, 2048); }},
    {"registers only: ~120 VGPRs (launch bounds 256 x 4)", [](float *o, hipStream_t s) { launch_hog<116, 4, false, false>(o, s, 0, 2048); }},
    {"registers only: ~168 VGPRs (launch bounds 256 x 3)", [](float *o, hipStream_t s) { launch_hog<164, 3, false, false>(o, s, 0, 2048); }},
    {"registers only: ~250 VGPRs (launch bounds 256 x 2)", [](float *o, hipStream_t s) { launch_hog<244, 2, false, false>(o, s, 0, 2048); }},
    {"LDS only: 37 KB per workgroup, 40 VGPRs", [](float *o, hipStream_t s) { launch_hog<32, 4, true, false>(o, s, 37 * 1024, 2048); }},
    {"LDS only: 49.6 KB per workgroup, 40 VGPRs", [](float *o, hipStream_t s) { launch_hog<32, 4, true, false>(o, s, 49664, 2048); }},
    {"LDS only: 76 KB per workgroup, 40 VGPRs", [](float *o, hipStream_t s) { launch_hog<32, 4, true, false>(o, s, 76 * 1024, 2048); }},
    {"MFMA only: bf16 32x32x16 chains, 40 VGPRs", [](float *o, hipStream_t s) { launch_hog<16, 4, false, true>(o, s, 0, 2048); }},
    {"~168 VGPRs + 49.6 KB LDS", [](float *o, hipStream_t s) { launch_hog<146, 3, true, false>(o, s, 49664, 2048); }},
    {"~168 VGPRs + MFMA", [](float *o, hipStream_t s) { launch_hog<140, 3, false, true>(o, s, 0, 2048); }},
    {"~168 VGPRs + 49.6 KB LDS + MFMA (the split tile's footprint)", [](float *o, hipStream_t s) { launch_hog<126, 3, true, true>(o, s, 49664, 2048); }},
    {"~120 VGPRs + 37 KB LDS + MFMA (the 128 x 64 tile's footprint)", [](float *o, hipStream_t s) { launch_hog<88, 4, true, true>(o, s, 37 * 1024, 2048); }},
    {"40 VGPRs + 37 KB LDS + MFMA", [](float *o, hipStream_t s) { launch_hog<16, 4, true, true>(o, s, 37 * 1024, 2048); }},
    {"40 VGPRs + 4 KB LDS + MFMA", [](float *o, hipStream_t s) { launch_hog<16, 4, true, true>(o, s, 4 * 1024, 2048); }},
    {"~120 VGPRs + 4 KB LDS + MFMA", [](float *o, hipStream_t s) { launch_hog<88, 4, true, true>(o, s, 4 * 1024, 2048); }},
    {"~120 VGPRs + 37 KB LDS, no MFMA", [](float *o, hipStream_t s) { launch_hog<100, 4, true, false>(o, s, 37 * 1024, 2048); }},
};
constexpr int kNumAggressors = sizeof(kAggressors) / sizeof(kAggressors[0]);

static void child_loop(int which, int ready_fd) {
    float *out;
    CK(hipMalloc(&out, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    char c = 'r';
    bool told = false;
    for (;;) {
        for (int i = 0; i < 8; ++i) kAggressors[which].launch(out, st);
        CK(hipStreamSynchronize(st));
        if (!told) {
            if (write(ready_fd, &c, 1) != 1) exit(3);
            told = true;
        }
    }
}

// ---- victims ----------------------------------------------------------------------------------------------------------------
// synthetic: 256 threads pass a 64 x 64 tile through LDS transposes, four barrier-separated rounds; every output word is a
// function of the input only
__global__ __launch_bounds__(256) void exchange_kernel(const float *in, float *out) {
    __shared__ float t[64 * 65];
    const int tid = threadIdx.x;
    const float *src = in + (size_t)blockIdx.x * 4096;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = src[tid + 256 * i];
    for (int round = 0; round < 4; ++round) {
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            t[(e >> 6) * 65 + (e & 63)] = v[i];
        }
        __syncthreads();
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            v[i] = t[(e & 63) * 65 + (e >> 6)] * 1.0009765625f + (float)round;
        }
        __syncthreads();
    }
    for (int i = 0; i < 16; ++i) out[(size_t)blockIdx.x * 4096 + tid + 256 * i] = v[i];
}

// the same exchange through DYNAMIC LDS with 16-byte accesses (ds_write_b128 / ds_read_b128), as the FFT rows kernels use
__global__ __launch_bounds__(256) void exchange128_kernel(const float *in, float *out) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float4 *t = reinterpret_cast<float4 *>(dyn);  // 1024 float4
    const int tid = threadIdx.x;
    const float4 *src = reinterpret_cast<const float4 *>(in + (size_t)blockIdx.x * 4096);
    float4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = src[tid + 256 * i];
    for (int round = 0; round < 4; ++round) {
        for (int i = 0; i < 4; ++i) t[tid + 256 * i] = v[i];
        __syncthreads();
        for (int i = 0; i < 4; ++i) {
            const float4 u = t[(tid * 37 + 256 * i + 101 * round) & 1023];
            v[i] = make_float4(u.x * 1.0009765625f + (float)round, u.y + 1.f, u.z * 0.5f, u.w - (float)i);
        }
        __syncthreads();
    }
    float4 *dst = reinterpret_cast<float4 *>(out + (size_t)blockIdx.x * 4096);
    for (int i = 0; i < 4; ++i) dst[tid + 256 * i] = v[i];
}

// static LDS, 4-byte accesses, plus cross-lane traffic (v_readlane / v_writelane, what SGPR spills compile to)
__global__ __launch_bounds__(256) void lanes_kernel(const float *in, float *out) {
    __shared__ float t[64 * 65];
    const int tid = threadIdx.x;
    const float *src = in + (size_t)blockIdx.x * 4096;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = src[tid + 256 * i];
    for (int round = 0; round < 4; ++round) {
        for (int i = 0; i < 16; ++i) {
            const float other = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), (i * 7 + round) & 63));
            v[i] = v[i] * 0.75f + other * 0.25f;
            const int e = tid + 256 * i;
            t[(e >> 6) * 65 + (e & 63)] = v[i];
        }
        __syncthreads();
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i;
            v[i] = t[(e & 63) * 65 + (e >> 6)] + (float)round;
        }
        __syncthreads();
    }
    for (int i = 0; i < 16; ++i) out[(size_t)blockIdx.x * 4096 + tid + 256 * i] = v[i];
}

// UNBALANCED phases: in every round ONE wave works for a long time and publishes 64 values through LDS, the other three go straight to the
// barrier and then read them.  A barrier that lets a wave through early shows up as stale values (the FFT rows kernels are unbalanced in
// the same way: a radix-4 stage of a 12-point row has work for three of the eight half-wave slots).
__global__ __launch_bounds__(256) void unbalanced_kernel(const float *in, float *out, int spin) {
    __shared__ float t[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc = in[(size_t)blockIdx.x * 256 + tid];
    if (tid < 64) t[tid] = 0.f;
    __syncthreads();
    for (int round = 0; round < 16; ++round) {
        if (wave == (round & 3)) {
            float x = acc + (float)round;
            for (int i = 0; i < spin; ++i) x = x * 0.999f + 0.001f;
            t[lane] = x;
        }
        __syncthreads();
        acc += t[(lane + round) & 63];
        __syncthreads();
    }
    out[(size_t)blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int first = argc > 2 ? atoi(argv[2]) : 1;  // skip the aggressors before this index (0 = idle is always run)
    // spawn the children BEFORE this process touches HIP; each waits for its turn on a pipe
    int go[kNumAggressors][2], ready[kNumAggressors][2];
    pid_t pids[kNumAggressors];
    for (int a = first; a < kNumAggressors; ++a) {
        if (pipe(go[a]) || pipe(ready[a])) return 2;
        pids[a] = fork();
        if (pids[a] == 0) {
            prctl(PR_SET_PDEATHSIG, SIGKILL);  // a killed parent must not leave a child looping kernels on the GPU
            char c;
            if (read(go[a][0], &c, 1) != 1) exit(0);
            child_loop(a, ready[a][1]);
            exit(0);
        }
    }
    // victim inputs and quiet references
    const int B = 4, h = 32, w = 24, C = 192, N = w / 2, wk = N + 1;
    const size_t n_in = (size_t)B * h * w * C, plane = (size_t)h * wk * C, n_out = (size_t)B * 2 * plane;
    std::vector<float> hin(n_in), htab(2 * (2 * N + 1));
    uint32_t seed = 12345;
    for (auto &v : hin) seed = seed * 1664525u + 1013904223u, v = (float)((seed >> 8) & 0xffff) / 32768.f - 1.f;
    for (int j = 0; j < N; ++j) htab[2 * j] = (float)cos(2.0 * M_PI * j / N), htab[2 * j + 1] = (float)sin(2.0 * M_PI * j / N);
    for (int k = 0; k <= N; ++k) htab[2 * (N + k)] = (float)cos(2.0 * M_PI * k / w), htab[2 * (N + k) + 1] = (float)sin(2.0 * M_PI * k / w);
    float *din, *dtab, *dout, *xin, *xout;
    CK(hipMalloc(&din, n_in * 4));
    CK(hipMalloc(&dtab, htab.size() * 4));
    CK(hipMalloc(&dout, n_out * 4));
    const int XWG = 512;
    CK(hipMalloc(&xin, (size_t)XWG * 4096 * 4));
    CK(hipMalloc(&xout, (size_t)XWG * 4096 * 4));
    CK(hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtab, htab.data(), htab.size() * 4, hipMemcpyHostToDevice));
    {
        std::vector<float> hx((size_t)XWG * 4096);
        for (auto &v : hx) seed = seed * 1664525u + 1013904223u, v = (float)((seed >> 8) & 0xffff) / 65536.f;
        CK(hipMemcpy(xin, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    auto run_rfft = [&](std::vector<float> &host) {
        CK(hipMemset(dout, 0xff, n_out * 4));
        if (mit_rfft_rows(din, (int64_t)h * w * C, (int64_t)w * C, C, dout, 2 * (int64_t)plane, (int64_t)plane, (int64_t)wk * C, C, dtab, B, h, w, C,
                          1.0f / sqrtf((float)w), nullptr)) {
            fprintf(stderr, "mit_rfft_rows: %s\n", mit_last_error());
            exit(2);
        }
        CK(hipDeviceSynchronize());
        host.resize(n_out);
        CK(hipMemcpy(host.data(), dout, n_out * 4, hipMemcpyDeviceToHost));
    };
    auto run_x = [&](std::vector<float> &host) {
        hipLaunchKernelGGL(exchange_kernel, dim3(XWG), dim3(256), 0, 0, xin, xout);
        CK(hipDeviceSynchronize());
        host.resize((size_t)XWG * 4096);
        CK(hipMemcpy(host.data(), xout, host.size() * 4, hipMemcpyDeviceToHost));
    };
    auto run_k = [&](int which, std::vector<float> &host) {
        if (which == 0) hipLaunchKernelGGL(exchange128_kernel, dim3(XWG), dim3(256), 16384, 0, xin, xout);
        else if (which == 2) hipLaunchKernelGGL(unbalanced_kernel, dim3(4096), dim3(256), 0, 0, xin, xout, 400);
        else hipLaunchKernelGGL(lanes_kernel, dim3(XWG), dim3(256), 0, 0, xin, xout);
        CK(hipDeviceSynchronize());
        host.resize((size_t)XWG * 4096);
        CK(hipMemcpy(host.data(), xout, host.size() * 4, hipMemcpyDeviceToHost));
    };
    std::vector<float> ref_r, ref_x, ref_b, ref_l, ref_u, got;
    run_rfft(ref_r);
    run_x(ref_x);
    run_k(0, ref_b);
    run_k(1, ref_l);
    run_k(2, ref_u);
    printf("%-66s %-22s %s\n", "co-tenant (another process)", "rfft_rows launches bad", "LDS-exchange b32 / dynamic b128 / b32 + readlane / UNBALANCED waves: launches bad");
    for (int a = 0; a < kNumAggressors; ++a) {
        if (a && a < first) continue;
        if (a) {
            char c = 'g';
            if (write(go[a][1], &c, 1) != 1 || read(ready[a][0], &c, 1) != 1) return 2;
            usleep(300 * 1000);
        }
        int bad_r = 0, bad_x = 0, bad_b = 0, bad_l = 0, bad_u = 0;
        for (int i = 0; i < launches; ++i) {
            run_rfft(got);
            bad_r += memcmp(got.data(), ref_r.data(), n_out * 4) != 0;
            run_x(got);
            bad_x += memcmp(got.data(), ref_x.data(), got.size() * 4) != 0;
            run_k(0, got);
            bad_b += memcmp(got.data(), ref_b.data(), got.size() * 4) != 0;
            run_k(1, got);
            bad_l += memcmp(got.data(), ref_l.data(), got.size() * 4) != 0;
            run_k(2, got);
            bad_u += memcmp(got.data(), ref_u.data(), (size_t)4096 * 256 * 4) != 0;
        }
        printf("%-66s %4d of %-14d %4d / %d / %d / %d of %d\n", kAggressors[a].name, bad_r, launches, bad_x, bad_b, bad_l, bad_u, launches);
        if (a) {
            kill(pids[a], SIGKILL);
            waitpid(pids[a], nullptr, 0);
        }
    }
    return 0;
}
