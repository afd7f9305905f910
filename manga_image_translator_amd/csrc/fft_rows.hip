// fft_rows.hip — real FFT / inverse real FFT along the W axis of LaMa's FourierUnit, as mixed-radix LDS butterflies.
//
// FourierUnit (manga_translator/inpainting/inpainting_lama_mpe.py:228,252) runs rfftn / irfftn over (H, W) with
// norm='ortho'.  W/8 is 182 = 2*7*13 for the BASELINE page, so the W axis is not a radix-2 problem; it used to be a dense
// DFT on the MFMA GEMM that executed 20x the transform's algorithmic FLOPs.  Here a real row of even length w is packed
// as w/2 complex points z[n] = x[2n] + i x[2n+1], transformed by a Stockham autosort FFT over the radices
// {2,3,4,5,7,11,13} (odd primes through their cos/sin symmetry: (P-1)^2 multiplies instead of 4 P^2), and untangled into
// the w/2+1 Hermitian bins (the inverse runs the same steps backwards and ignores the imaginary parts of the DC / Nyquist
// bins exactly like pocketfft's c2r).  Activations are NHWC, so a workgroup takes one (b, h) row and 32 adjacent channels:
// every global access is a 128-byte channel segment, every LDS access has the 32 lanes of a half-wave on consecutive
// (re, im) pairs — one full 256-byte bank row, conflict-free.  HBM-bound: the row is read once and its spectrum written
// once (forward), or the spectrum read once and the row + residual read/written once (inverse).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CC = 32;     // channels per workgroup
constexpr int NSLOT = 8;   // butterflies in flight per workgroup (256 threads / CC)
constexpr int MAXRAD = 8;

struct RowPlan {
    int nrad;
    int radix[MAXRAD];
};

// ---- compile-time cos / sin of 2 pi m / P (double Taylor series on [-pi, pi], rounded once to fp32) ----
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cx_cos(double x) {
    double term = 1.0, sum = 1.0;
    for (int i = 1; i <= 20; ++i) {
        term *= -x * x / ((2.0 * i - 1.0) * (2.0 * i));
        sum += term;
    }
    return sum;
}
constexpr double cx_sin(double x) {
    double term = x, sum = x;
    for (int i = 1; i <= 20; ++i) {
        term *= -x * x / ((2.0 * i) * (2.0 * i + 1.0));
        sum += term;
    }
    return sum;
}
template <int P>
struct TrigTab {
    float c[P], s[P];
    constexpr TrigTab() : c{}, s{} {
        for (int m = 0; m < P; ++m) {
            double a = 2.0 * kPi * m / P;
            if (a > kPi) a -= 2.0 * kPi;
            c[m] = (float)cx_cos(a);
            s[m] = (float)cx_sin(a);
        }
    }
};

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x));
}

// A (re, im) pair out of LDS.  SAFE = false: one 8-byte load — the butterfly inputs of a stage sit a constant stride apart and hipcc
// merges them into two-address reads (ds_read2_b64 / ds_read2st64_b64).  SAFE = true (the kernels of MIT_COTENANT_SAFE launches): TWO
// 4-byte loads, kept apart by the volatile — the mitigation of round 4 for the co-tenancy failure (wrong workgroups while another
// queue's MFMA kernel shares the CU), together with a whole CU's LDS per launch; 30-45 % slower (385 vs 265 us, 437 vs 335 us per
// 16-page launch).  The failure turned out to sit in the butterflies' ARITHMETIC, not in the reads: the SLP vectoriser had packed it
// into v_pk_*_f32 with op_sel / neg modifiers (1317 of them in this file), which go wrong beside MFMA co-tenants on gfx950; the narrow
// reads only changed the instruction mix enough to hide it for the radix-4 / 3 plan.  Built with -fno-slp-vectorize the default form is
// exact (tests/test_cotenant_gpu.py); the SAFE form stays as a switch.
template <bool SAFE>
__device__ __forceinline__ float2 lds_pair(const float2 *p) {
    if constexpr (SAFE) {
        const volatile float *f = reinterpret_cast<const volatile float *>(p);
        return make_float2(f[0], f[1]);
    } else {
        return *p;
    }
}

// y_k = sum_i x_i exp(-/+ 2 pi i ik / P), in place.
template <int P, bool INV>
__device__ __forceinline__ void butterfly(float2 (&x)[P]) {
    if constexpr (P == 2) {
        const float2 a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
    } else if constexpr (P == 4) {
        const float2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]), s13 = cadd(x[1], x[3]), d13 = csub(x[1], x[3]);
        // forward: -i * d13 = (d13.y, -d13.x); inverse: +i * d13 = (-d13.y, d13.x)
        const float2 r = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
        x[0] = cadd(s02, s13);
        x[2] = csub(s02, s13);
        x[1] = cadd(d02, r);
        x[3] = csub(d02, r);
    } else {
        constexpr int H = (P - 1) / 2;
        constexpr TrigTab<P> T{};
        float2 a[H], b[H];
#pragma unroll
        for (int j = 0; j < H; ++j) {
            a[j] = cadd(x[j + 1], x[P - 1 - j]);
            b[j] = csub(x[j + 1], x[P - 1 - j]);
        }
        const float2 x0 = x[0];
        float2 y0 = x0;
#pragma unroll
        for (int j = 0; j < H; ++j) y0 = cadd(y0, a[j]);
        x[0] = y0;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            float2 ck = x0, sk = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 1; j <= H; ++j) {
                const float c = T.c[(j * k) % P], s = T.s[(j * k) % P];
                ck.x = __builtin_fmaf(a[j - 1].x, c, ck.x);
                ck.y = __builtin_fmaf(a[j - 1].y, c, ck.y);
                sk.x = __builtin_fmaf(b[j - 1].x, s, sk.x);
                sk.y = __builtin_fmaf(b[j - 1].y, s, sk.y);
            }
            // forward: y_k = C - i S, y_{P-k} = C + i S  (i S = (-S.y, S.x)); inverse: swapped
            const float2 lo = make_float2(ck.x + sk.y, ck.y - sk.x), hi = make_float2(ck.x - sk.y, ck.y + sk.x);
            x[k] = INV ? hi : lo;
            x[P - k] = INV ? lo : hi;
        }
    }
}

// One Stockham stage of radix P over the N-point sequences of CC channels: current sub-length n, stride s.
//   dst[q + s (P p + k)] = (sum_i src[q + s (p + m i)] w_P^{ik}) w_n^{pk},   m = n / P, q < s, p < m
template <int P, bool INV, bool SAFE>
__device__ __forceinline__ void stage(const float2 *__restrict__ src, float2 *__restrict__ dst, const float2 *__restrict__ tw, int N,
                                      int n, int s, int slot, int c) {
    const int m = n / P, nb = N / P, tws = N / n;
    for (int bf = slot; bf < nb; bf += NSLOT) {
        const int q = bf % s, p = bf / s;
        float2 x[P];
#pragma unroll
        for (int i = 0; i < P; ++i) x[i] = lds_pair<SAFE>(src + (q + s * (p + m * i)) * CC + c);
        butterfly<P, INV>(x);
        if (m > 1) {
#pragma unroll
            for (int k = 1; k < P; ++k) {
                float2 w = tw[p * k * tws];  // (cos, sin)(2 pi p k / n); p k < n
                w.y = INV ? w.y : -w.y;
                x[k] = cmul(x[k], w);
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) dst[(q + s * (P * p + k)) * CC + c] = x[k];
    }
}

// Runs every stage of the plan, ping-ponging between the two LDS buffers; returns the buffer holding the result.
template <bool INV, bool SAFE>
__device__ __forceinline__ float2 *run_stages(float2 *a, float2 *b, const float2 *tw, const RowPlan &plan, int N, int slot, int c) {
    int n = N, s = 1;
    for (int st = 0; st < plan.nrad; ++st) {
        const int P = plan.radix[st];
        switch (P) {
            case 2: stage<2, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
            case 3: stage<3, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
            case 4: stage<4, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
            case 5: stage<5, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
            case 7: stage<7, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
            case 11: stage<11, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
            default: stage<13, INV, SAFE>(a, b, tw, N, n, s, slot, c); break;
        }
        __syncthreads();
        float2 *t = a;
        a = b;
        b = t;
        n /= P;
        s *= P;
    }
    return a;
}

constexpr int LD_UNROLL = 3;  // (row pair) loads in flight per thread: 3 x 32 pairs cover N <= 96 in one round

// Forward: x[b, h, w, C] (real) -> planar spectrum out[b, t, h, k, C], k <= w/2, scaled by `scale` (1/sqrt(w) for 'ortho').
template <bool SAFE>
__global__ __launch_bounds__(256) void rfft_rows_kernel(const float *__restrict__ in, int64_t in_bs, int64_t in_hs, int64_t in_ws,
                                                         float *__restrict__ out, int64_t out_bs, int64_t out_ts, int64_t out_hs,
                                                         int64_t out_ks, const float2 *__restrict__ tables, RowPlan plan, int N,
                                                         int Cn, float scale, int dbg_zero) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2 *bufA = lds2;                    // [N + 1][CC]
    float2 *bufB = lds2 + (N + 1) * CC;     // [N + 1][CC]
    if (dbg_zero) {  // diagnostics (scripts/diag_rfft_load.py): start from a cleared LDS image
        for (int j = threadIdx.x; j < 2 * (N + 1) * CC + 2 * N + 1; j += 256) lds2[j] = make_float2(0.f, 0.f);
        __syncthreads();
    }
    float2 *tw = lds2 + 2 * (N + 1) * CC;   // [N]      (cos, sin)(2 pi j / N)
    float2 *tw2 = tw + N;                   // [N + 1]  (cos, sin)(2 pi k / w), w = 2 N
    const int c0 = blockIdx.x * CC;
    const float *ib = in + (int64_t)blockIdx.z * in_bs + (int64_t)blockIdx.y * in_hs + c0;
    float *ob = out + (int64_t)blockIdx.z * out_bs + (int64_t)blockIdx.y * out_hs + c0;
    const int l8 = threadIdx.x & 7, nrow = threadIdx.x >> 3;
    const bool cok = c0 + l8 * 4 < Cn;  // Cn % 4 == 0
    for (int base = 0; base < N; base += 32 * LD_UNROLL) {
        f32x4 ev[LD_UNROLL], od[LD_UNROLL];
#pragma unroll
        for (int u = 0; u < LD_UNROLL; ++u) {
            const int n0 = base + nrow + 32 * u;
            ev[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            od[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n0 < N && cok) {
                const float *p = ib + (int64_t)(2 * n0) * in_ws + l8 * 4;
                ev[u] = *reinterpret_cast<const f32x4 *>(p);
                od[u] = *reinterpret_cast<const f32x4 *>(p + in_ws);
            }
        }
#pragma unroll
        for (int u = 0; u < LD_UNROLL; ++u) {
            const int n0 = base + nrow + 32 * u;
            if (n0 < N) {
                f32x4 *d = reinterpret_cast<f32x4 *>(bufA + n0 * CC + l8 * 4);
                d[0] = f32x4{ev[u][0], od[u][0], ev[u][1], od[u][1]};
                d[1] = f32x4{ev[u][2], od[u][2], ev[u][3], od[u][3]};
            }
        }
    }
    for (int j = threadIdx.x; j < 2 * N + 1; j += 256) tw[j] = tables[j];
    __syncthreads();
    const int slot = threadIdx.x >> 5, c = threadIdx.x & 31;
    const float2 *Z = run_stages<false, SAFE>(bufA, bufB, tw, plan, N, slot, c);
    // untangle: X[k] = (Z[k] + conj Z[N-k]) / 2 - i e^{-2 pi i k / w} (Z[k] - conj Z[N-k]) / 2,  k = 0 .. N (Z[N] = Z[0])
    const bool sok = c0 + c < Cn;
    const float hs = 0.5f * scale;
    for (int k = slot; k <= N; k += NSLOT) {
        const float2 zk = lds_pair<SAFE>(Z + (k == N ? 0 : k) * CC + c);
        const float2 zr = lds_pair<SAFE>(Z + (k == 0 ? 0 : N - k) * CC + c);
        const float2 zn = make_float2(zr.x, -zr.y);
        const float2 e = cadd(zk, zn), d = csub(zk, zn);
        const float2 w = tw2[k];                                  // (cos, sin); e^{-i th} = (cos, -sin)
        const float2 o = cmul(make_float2(d.y, -d.x), make_float2(w.x, -w.y));  // -i d e^{-i th}
        if (sok) {
            float *po = ob + (int64_t)k * out_ks + c;
            po[0] = (e.x + o.x) * hs;
            po[out_ts] = (e.y + o.y) * hs;
        }
    }
}

// Inverse: planar Hermitian half spectrum in[b, t, h, k, C], k <= w/2 -> real rows out[b, h, w, C] = scale * irfft (+ res).
template <bool SAFE>
__global__ __launch_bounds__(256) void irfft_rows_kernel(const float *__restrict__ in, int64_t in_bs, int64_t in_ts, int64_t in_hs,
                                                          int64_t in_ks, float *__restrict__ out, int64_t out_bs, int64_t out_hs,
                                                          int64_t out_ws, const float *__restrict__ res, int64_t res_bs,
                                                          int64_t res_hs, int64_t res_ws, const float2 *__restrict__ tables,
                                                          RowPlan plan, int N, int Cn, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2 *bufA = lds2;
    float2 *bufB = lds2 + (N + 1) * CC;
    float2 *tw = lds2 + 2 * (N + 1) * CC;
    float2 *tw2 = tw + N;
    const int c0 = blockIdx.x * CC;
    const float *ib = in + (int64_t)blockIdx.z * in_bs + (int64_t)blockIdx.y * in_hs + c0;
    float *ob = out + (int64_t)blockIdx.z * out_bs + (int64_t)blockIdx.y * out_hs + c0;
    const float *rb = res ? res + (int64_t)blockIdx.z * res_bs + (int64_t)blockIdx.y * res_hs + c0 : nullptr;
    const int l8 = threadIdx.x & 7, nrow = threadIdx.x >> 3;
    const bool cok = c0 + l8 * 4 < Cn;
    for (int base = 0; base <= N; base += 32 * LD_UNROLL) {
        f32x4 re[LD_UNROLL], im[LD_UNROLL];
#pragma unroll
        for (int u = 0; u < LD_UNROLL; ++u) {
            const int k = base + nrow + 32 * u;
            re[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            im[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k <= N && cok) {
                const float *p = ib + (int64_t)k * in_ks + l8 * 4;
                re[u] = *reinterpret_cast<const f32x4 *>(p);
                if (k != 0 && k != N) im[u] = *reinterpret_cast<const f32x4 *>(p + in_ts);  // c2r ignores Im of DC / Nyquist
            }
        }
#pragma unroll
        for (int u = 0; u < LD_UNROLL; ++u) {
            const int k = base + nrow + 32 * u;
            if (k <= N) {
                f32x4 *d = reinterpret_cast<f32x4 *>(bufA + k * CC + l8 * 4);
                d[0] = f32x4{re[u][0], im[u][0], re[u][1], im[u][1]};
                d[1] = f32x4{re[u][2], im[u][2], re[u][3], im[u][3]};
            }
        }
    }
    for (int j = threadIdx.x; j < 2 * N + 1; j += 256) tw[j] = tables[j];
    __syncthreads();
    const int slot = threadIdx.x >> 5, c = threadIdx.x & 31;
    // tangle: Z[k] = (X[k] + conj X[N-k]) + i e^{+2 pi i k / w} (X[k] - conj X[N-k]),  k < N
    for (int k = slot; k < N; k += NSLOT) {
        const float2 xk = lds_pair<SAFE>(bufA + k * CC + c);
        const float2 xr = lds_pair<SAFE>(bufA + (N - k) * CC + c);
        const float2 xn = make_float2(xr.x, -xr.y);
        const float2 e = cadd(xk, xn), d = csub(xk, xn);
        const float2 o = cmul(make_float2(-d.y, d.x), tw2[k]);  // i d e^{+i th}
        bufB[k * CC + c] = cadd(e, o);
    }
    __syncthreads();
    const float2 *Z = run_stages<true, SAFE>(bufB, bufA, tw, plan, N, slot, c);
    // z[n] = x[2n] + i x[2n+1]; same thread <-> (row pair, 4 channels) mapping as the forward load
    for (int base = 0; base < N; base += 32 * LD_UNROLL) {
        f32x4 r0[LD_UNROLL], r1[LD_UNROLL];
#pragma unroll
        for (int u = 0; u < LD_UNROLL; ++u) {
            const int n0 = base + nrow + 32 * u;
            r0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            r1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rb && n0 < N && cok) {
                const float *p = rb + (int64_t)(2 * n0) * res_ws + l8 * 4;
                r0[u] = *reinterpret_cast<const f32x4 *>(p);
                r1[u] = *reinterpret_cast<const f32x4 *>(p + res_ws);
            }
        }
#pragma unroll
        for (int u = 0; u < LD_UNROLL; ++u) {
            const int n0 = base + nrow + 32 * u;
            if (n0 < N && cok) {
                const f32x4 *sv = reinterpret_cast<const f32x4 *>(Z + n0 * CC + l8 * 4);
                const f32x4 a = sv[0], b = sv[1];
                f32x4 ev = f32x4{a[0], a[2], b[0], b[2]}, od = f32x4{a[1], a[3], b[1], b[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ev[e] = __builtin_fmaf(ev[e], scale, r0[u][e]);
                    od[e] = __builtin_fmaf(od[e], scale, r1[u][e]);
                }
                float *p = ob + (int64_t)(2 * n0) * out_ws + l8 * 4;
                *reinterpret_cast<f32x4 *>(p) = ev;
                *reinterpret_cast<f32x4 *>(p + out_ws) = od;
            }
        }
    }
}

int make_plan(int N, RowPlan *plan) {
    static const int kRadices[] = {4, 2, 3, 5, 7, 11, 13};
    plan->nrad = 0;
    int n = N;
    for (int r : kRadices)
        while (n % r == 0) {
            if (plan->nrad == MAXRAD) return 1;
            plan->radix[plan->nrad++] = r;
            n /= r;
        }
    return n != 1;
}

bool aligned4(int64_t v) { return (v & 3) == 0; }

// with MIT_COTENANT_SAFE: 140 of a CU's 160 KB — no workgroup of the split-bf16 GEMM tiles (>= 37 KB) fits beside it (round 3: "giving the
// victim a whole CU removes it"); costs these two kernels their own co-residency (three workgroups per CU otherwise)
constexpr size_t kCotenantSafeLds = 140 * 1024;

}  // namespace

extern "C" int mit_rfft_rows_supported(int w) {
    RowPlan plan;
    return w >= 4 && w <= 512 && (w & 1) == 0 && make_plan(w / 2, &plan) == 0;
}

extern "C" int mit_rfft_rows(const float *in_dev, int64_t in_bs, int64_t in_hs, int64_t in_ws, float *out_dev, int64_t out_bs,
                             int64_t out_ts, int64_t out_hs, int64_t out_ks, const float *tables_dev, int B, int h, int w, int C,
                             float scale, void *stream) {
    if (!in_dev || !out_dev || !tables_dev) return mit_set_error("mit_rfft_rows: null pointer");
    RowPlan plan;
    if (!mit_rfft_rows_supported(w) || make_plan(w / 2, &plan))
        return mit_set_error("mit_rfft_rows: w = %d is not an even product of {2,3,5,7,11,13} <= 512", w);
    if (B <= 0 || B > 65535 || h <= 0 || h > 65535 || C <= 0 || (C & 3)) return mit_set_error("mit_rfft_rows: bad size");
    if (!aligned4(in_bs) || !aligned4(in_hs) || !aligned4(in_ws) || !aligned4(out_bs) || !aligned4(out_ts) || !aligned4(out_hs) ||
        !aligned4(out_ks) || (reinterpret_cast<uintptr_t>(in_dev) & 15) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return mit_set_error("mit_rfft_rows: strides and bases must be multiples of 4 floats");
    const int N = w / 2;
    size_t smem = ((size_t)2 * (N + 1) * CC + 2 * N + 1) * sizeof(float2);
    int dbg_zero = 0;
    if (const char *e = getenv("MIT_FFT_ROWS_DEBUG")) {  // diagnostics: "<extra LDS bytes>,<clear LDS first 0|1>"
        int pad = 0;
        if (sscanf(e, "%d,%d", &pad, &dbg_zero) >= 1 && pad > 0) smem += (size_t)pad;
    }
    const bool safe = mit_cotenant_safe();
    if (safe && smem < kCotenantSafeLds) smem = kCotenantSafeLds;
    static DynSmemOptIn optin_fast, optin_safe;
    auto kern = safe ? rfft_rows_kernel<true> : rfft_rows_kernel<false>;
    (safe ? optin_safe : optin_fast).ensure(reinterpret_cast<const void *>(kern), smem);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const double rows = (double)B * h * C;
    // algorithmic bytes: the real rows read once, the half spectrum written once; FLOPs: 2.5 w log2 w per real row
    MitProbeScope probe("rfft_rows_kernel", st, 4.0 * rows * (w + 2.0 * (N + 1)), 2.5 * w * log2((double)w) * rows);
    dim3 grid(mit_div_up(C, CC), h, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, st, in_dev, in_bs, in_hs, in_ws, out_dev, out_bs, out_ts, out_hs, out_ks,
                       reinterpret_cast<const float2 *>(tables_dev), plan, N, C, scale, dbg_zero);
    MIT_CHECK_LAUNCH("mit_rfft_rows");
    return 0;
}

extern "C" int mit_irfft_rows(const float *in_dev, int64_t in_bs, int64_t in_ts, int64_t in_hs, int64_t in_ks, float *out_dev,
                              int64_t out_bs, int64_t out_hs, int64_t out_ws, const float *res_dev, int64_t res_bs, int64_t res_hs,
                              int64_t res_ws, const float *tables_dev, int B, int h, int w, int C, float scale, void *stream) {
    if (!in_dev || !out_dev || !tables_dev) return mit_set_error("mit_irfft_rows: null pointer");
    RowPlan plan;
    if (!mit_rfft_rows_supported(w) || make_plan(w / 2, &plan))
        return mit_set_error("mit_irfft_rows: w = %d is not an even product of {2,3,5,7,11,13} <= 512", w);
    if (B <= 0 || B > 65535 || h <= 0 || h > 65535 || C <= 0 || (C & 3)) return mit_set_error("mit_irfft_rows: bad size");
    if (!aligned4(in_bs) || !aligned4(in_ts) || !aligned4(in_hs) || !aligned4(in_ks) || !aligned4(out_bs) || !aligned4(out_hs) ||
        !aligned4(out_ws) || (reinterpret_cast<uintptr_t>(in_dev) & 15) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return mit_set_error("mit_irfft_rows: strides and bases must be multiples of 4 floats");
    if (res_dev && (!aligned4(res_bs) || !aligned4(res_hs) || !aligned4(res_ws) || (reinterpret_cast<uintptr_t>(res_dev) & 15)))
        return mit_set_error("mit_irfft_rows: residual strides and base must be multiples of 4 floats");
    const int N = w / 2;
    size_t smem = ((size_t)2 * (N + 1) * CC + 2 * N + 1) * sizeof(float2);
    const bool safe = mit_cotenant_safe();
    if (safe && smem < kCotenantSafeLds) smem = kCotenantSafeLds;
    static DynSmemOptIn optin_fast, optin_safe;
    auto kern = safe ? irfft_rows_kernel<true> : irfft_rows_kernel<false>;
    (safe ? optin_safe : optin_fast).ensure(reinterpret_cast<const void *>(kern), smem);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const double rows = (double)B * h * C;
    MitProbeScope probe("irfft_rows_kernel", st, 4.0 * rows * ((res_dev ? 2.0 : 1.0) * w + 2.0 * (N + 1)), 2.5 * w * log2((double)w) * rows);
    dim3 grid(mit_div_up(C, CC), h, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, st, in_dev, in_bs, in_ts, in_hs, in_ks, out_dev, out_bs, out_hs, out_ws,
                       res_dev, res_bs, res_hs, res_ws, reinterpret_cast<const float2 *>(tables_dev), plan, N, C, scale);
    MIT_CHECK_LAUNCH("mit_irfft_rows");
    return 0;
}
