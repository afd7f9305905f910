// grid_barrier_check.cpp — what does a hand-written grid barrier cost on MI355X, and are writes of one XCD visible on the others after it?
// Decides whether the B = 1 decoder step (≈ 60 dependent launches of 4.5-16 us, ocr_decoder.hip) is worth a persistent kernel.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 scripts/dev/grid_barrier_check.cpp -o scripts/dev/grid_barrier_check
//   scripts/dev/grid_barrier_check
//
// Each phase: every workgroup writes `iter` into its slot of a buffer (plain stores), barrier, reads the slots of G/2+1 other workgroups
// (plain loads; they sit on other XCDs: workgroups are dealt round-robin) and counts mismatches, barrier.  Also timed: the same number
// of empty dependent kernel launches on one stream (the launch floor the barrier competes with).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

// MODE 0: agent-scope fences both sides (the portable form); MODE 1: release fence + acquire fence only in thread 0, spin with s_sleep
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned *ctr, const unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (MODE == 1) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void phases_kernel(unsigned *ctr, int *slots, int *bad, const int iters, const int words) {
    const int G = gridDim.x, g = blockIdx.x;
    unsigned target = 0;
    int mism = 0;
    for (int it = 1; it <= iters; ++it) {
        for (int i = threadIdx.x; i < words; i += 256) slots[(size_t)g * words + i] = it;
        target += G;
        grid_barrier<MODE>(ctr, target);
        for (int j = 1; j <= 9; ++j) {
            const int o = (g + j * (G / 9 + 1)) % G;
            for (int i = threadIdx.x; i < words; i += 256) mism += slots[(size_t)o * words + i] != it;
        }
        target += G;
        grid_barrier<MODE>(ctr, target);
    }
    if (mism) atomicAdd(bad, mism);
}

__global__ void empty_kernel(int *p) {
    if (p && threadIdx.x == 1234567) *p = 1;
}
__global__ __launch_bounds__(256) void small_kernel(int *slots, const int it, const int words) {
    for (int i = threadIdx.x; i < words; i += 256) slots[(size_t)blockIdx.x * words + i] = it;
}

template <int MODE>
static void run(int G, int iters, int words) {
    unsigned *ctr;
    int *slots, *bad;
    CK(hipMalloc(&ctr, 256));
    CK(hipMalloc(&slots, (size_t)G * words * 4));
    CK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    int hb = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(ctr, 0, 256));
        CK(hipMemset(bad, 0, 4));
        CK(hipMemset(slots, 0, (size_t)G * words * 4));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(phases_kernel<MODE>, dim3(G), dim3(256), 0, 0, ctr, slots, bad, iters, words);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        int b;
        CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
        hb += b;
    }
    printf("mode %d  G=%4d  words/WG=%5d : %7.2f us per barrier (incl. its phase), stale reads %d\n", MODE, G, words, best * 1e3f / (2.f * iters), hb);
    CK(hipFree(ctr));
    CK(hipFree(slots));
    CK(hipFree(bad));
}

int main() {
    const int iters = 500;
    for (int words : {256, 8192})
        for (int G : {32, 64, 128, 256, 512}) {
            run<0>(G, iters, words);
            run<1>(G, iters, words);
        }
    // the launch floor: dependent kernels on one stream
    int *slots;
    CK(hipMalloc(&slots, (size_t)512 * 8192 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int G : {64, 256}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(small_kernel, dim3(G), dim3(256), 0, 0, slots, i, 256);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("1000 dependent launches of a %d-workgroup store kernel: %.2f us each\n", G, ms);
        }
    }
    return 0;
}
