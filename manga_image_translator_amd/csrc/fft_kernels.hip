// fft_kernels.hip — complex FFT along the H axis of LaMa's FourierUnit, as LDS butterflies.
//
// FourierUnit (manga_translator/inpainting/inpainting_lama_mpe.py:228,252) runs rfftn / irfftn over (H, W) with
// norm='ortho'.  The W axis (W/8 = 182 = 2*7*13 for the BASELINE page, not a power of two) stays a small dense DFT on
// the MFMA GEMM; the H axis (H/8 = 256) is this kernel: a radix-2 decimation-in-time FFT over columns of planar
// re/im data [t][h][col], 32 adjacent columns per workgroup so that every global access is a full 128-byte row
// segment and every LDS access has the 32 lanes of a half-wave on 32 consecutive banks.  HBM-bound: one read and one
// write of the tensor (72 MB per call at 192 ch x 256 x 92 bins), against 9.3 GFLOP for the same transform as a dense
// [2h x 2h] GEMM.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mit_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CT = 32;     // columns per workgroup

template <int S>
__device__ __forceinline__ void fft_pass(float *re, float *im, const float2 *tws, const int h, const int s0, const int inverse) {
    constexpr int P = 1 << S;
    const int nb = h >> 1;
    const int ntask = (h >> S) * CT;
    for (int task = threadIdx.x; task < ntask; task += blockDim.x) {
        const int col = task & (CT - 1), grp = task / CT;
        const int lo = grp & ((1 << s0) - 1), hi = grp >> s0;
        const int base = (hi << (s0 + S)) | lo;  // row of member m: base | (m << s0)
        float xr[P], xi[P];
#pragma unroll
        for (int m = 0; m < P; ++m) {
            xr[m] = re[(base | (m << s0)) * CT + col];
            xi[m] = im[(base | (m << s0)) * CT + col];
        }
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const int s = s0 + t;
            const int tstep = nb >> s;  // twiddle index stride: k * h / (2 * half), half = 1 << s
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (m & (1 << t)) continue;
                const int m1 = m | (1 << t);
                const int k = ((m & ((1 << t) - 1)) << s0) | lo;  // row(m) & (half - 1)
                const float2 w = tws[k * tstep];  // (cos, sin) of 2 pi k tstep / h
                const float wr = w.x, wi = inverse ? w.y : -w.y;
                const float tr = xr[m1] * wr - xi[m1] * wi, ti = xr[m1] * wi + xi[m1] * wr;
                const float ur = xr[m], ui = xi[m];
                xr[m] = ur + tr;
                xi[m] = ui + ti;
                xr[m1] = ur - tr;
                xi[m1] = ui - ti;
            }
        }
#pragma unroll
        for (int m = 0; m < P; ++m) {
            re[(base | (m << s0)) * CT + col] = xr[m];
            im[(base | (m << s0)) * CT + col] = xi[m];
        }
    }
}

template <int ROWS, int NT = 256>  // rows per thread in the load / store passes = h / (NT / 8)
__global__ __launch_bounds__(NT) void fft_cols_kernel(const float *__restrict__ in, int64_t in_bs, int64_t in_ts, int64_t in_hs,
                                                        float *__restrict__ out, int64_t out_bs, int64_t out_ts, int64_t out_hs,
                                                        const float2 *__restrict__ tw, int h, int logh, int64_t ncols,
                                                        int inverse, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *re = lds;               // [h][CT]
    float *im = lds + h * CT;
    float2 *tws = reinterpret_cast<float2 *>(lds + 2 * h * CT);  // [h/2]
    const int64_t cb = (int64_t)blockIdx.x * CT;
    const float *ib = in + (int64_t)blockIdx.y * in_bs + cb;
    float *ob = out + (int64_t)blockIdx.y * out_bs + cb;
    // ---- load: thread (cq, rg) fetches 4 adjacent columns of rows rg, rg+32, ...: a wave covers 8 rows x 128 B per
    // instruction and all 2*ROWS loads are in flight together; rows go to their bit-reversed LDS slot (decimation in time)
    const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const bool vec_ok = cb + cq * 4 + 3 < ncols;
    f32x4 va[ROWS], vb[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int r = rg + (NT / 8) * i;
        va[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        vb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < h) {
            const float *pa = ib + (int64_t)r * in_hs + cq * 4;
            if (vec_ok) {
                va[i] = *reinterpret_cast<const f32x4 *>(pa);
                vb[i] = *reinterpret_cast<const f32x4 *>(pa + in_ts);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (cb + cq * 4 + e < ncols) {
                        va[i][e] = pa[e];
                        vb[i][e] = pa[in_ts + e];
                    }
            }
        }
    }
    for (int k = threadIdx.x; k < (h >> 1); k += NT) tws[k] = tw[k];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int r = rg + (NT / 8) * i;
        if (r < h) {
            const int rr = __brev((unsigned)r) >> (32 - logh);
            *reinterpret_cast<f32x4 *>(re + rr * CT + cq * 4) = va[i];
            *reinterpret_cast<f32x4 *>(im + rr * CT + cq * 4) = vb[i];
        }
    }
    __syncthreads();
    // ---- butterflies: the radix-2 decimation-in-time graph, S stages at a time in registers.  Stages s0 .. s0+S-1 only mix
    // the 2^S rows that differ in bits [s0, s0+S) of the row index, so one thread owns such a set for one column and the LDS
    // round trips drop from logh to ceil(logh / 4).  Every butterfly evaluates the same expression with the same twiddle as
    // the stage-at-a-time form, so the results are bit-identical to it.
    {
        const int npass = (logh + 3) >> 2;
        int s0 = 0;
        for (int ps = 0; ps < npass; ++ps) {
            const int S = (logh - s0 + (npass - ps) - 1) / (npass - ps);  // even split, larger passes first
            switch (S) {
                case 1: fft_pass<1>(re, im, tws, h, s0, inverse); break;
                case 2: fft_pass<2>(re, im, tws, h, s0, inverse); break;
                case 3: fft_pass<3>(re, im, tws, h, s0, inverse); break;
                default: fft_pass<4>(re, im, tws, h, s0, inverse); break;
            }
            s0 += S;
            __syncthreads();
        }
    }
    // ---- store (same mapping as the load) ----
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int r = rg + (NT / 8) * i;
        if (r >= h) continue;
        f32x4 a = *reinterpret_cast<const f32x4 *>(re + r * CT + cq * 4);
        f32x4 b = *reinterpret_cast<const f32x4 *>(im + r * CT + cq * 4);
        a *= scale;
        b *= scale;
        float *po = ob + (int64_t)r * out_hs + cq * 4;
        if (vec_ok) {
            *reinterpret_cast<f32x4 *>(po) = a;
            *reinterpret_cast<f32x4 *>(po + out_ts) = b;
        } else {
            for (int e = 0; e < 4; ++e)
                if (cb + cq * 4 + e < ncols) {
                    po[e] = a[e];
                    po[out_ts + e] = b[e];
                }
        }
    }
}

}  // namespace

extern "C" int mit_fft_cols(const float *in_dev, int64_t in_bs, int64_t in_ts, int64_t in_hs, float *out_dev, int64_t out_bs,
                            int64_t out_ts, int64_t out_hs, const float *twiddle_dev, int B, int h, int64_t ncols, int inverse,
                            float scale, void *stream) {
    if (!in_dev || !out_dev || !twiddle_dev) return mit_set_error("mit_fft_cols: null pointer");
    if (h < 2 || h > 512 || (h & (h - 1))) return mit_set_error("mit_fft_cols: h must be a power of two in [2, 512] (got %d)", h);
    if (B <= 0 || B > 65535 || ncols <= 0) return mit_set_error("mit_fft_cols: bad size");
    int logh = 0;
    while ((1 << logh) < h) ++logh;
    if ((ncols & 3) || (in_ts & 3) || (in_hs & 3) || (in_bs & 3) || (out_ts & 3) || (out_hs & 3) || (out_bs & 3) ||
        (reinterpret_cast<uintptr_t>(in_dev) & 15) || (reinterpret_cast<uintptr_t>(out_dev) & 15))
        return mit_set_error("mit_fft_cols: column count, strides and bases must be multiples of 4 floats");
    const size_t smem = (size_t)2 * h * CT * sizeof(float) + (size_t)(h / 2) * sizeof(float2);
    dim3 grid(mit_div_up(ncols, CT), B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float2 *tw2 = reinterpret_cast<const float2 *>(twiddle_dev);
    // algorithmic bytes: the planar complex spectrum read once and written once; FLOPs: 5 h log2 h per complex column
    MitProbeScope probe("fft_cols_kernel", st, 2.0 * 8.0 * (double)B * h * (double)ncols, 5.0 * (double)h * logh * (double)ncols * B);
#define MIT_FFT_LAUNCH(ROWS, NT)                                                                                               \
    do {                                                                                                                       \
        auto kern = fft_cols_kernel<ROWS, NT>;                                                                                 \
        static DynSmemOptIn optin;                                                                                             \
        optin.ensure(reinterpret_cast<const void *>(kern), smem);                                                              \
        hipLaunchKernelGGL(kern, grid, dim3(NT), smem, st, in_dev, in_bs, in_ts, in_hs, out_dev, out_bs, out_ts, out_hs, tw2, h, logh, \
                           ncols, inverse, scale);                                                                            \
    } while (0)
    // 64 KB of LDS per workgroup at h = 256 allows two workgroups per CU; 512 threads each (instead of 256) double the waves that
    // keep loads and stores in flight while the other workgroup is in its butterfly passes
    static const bool nt256 = getenv("MIT_FFT_256") != nullptr;  // A/B knob for scripts/
    if (h <= 32) MIT_FFT_LAUNCH(1, 256);
    else if (h <= 64) MIT_FFT_LAUNCH(2, 256);
    else if (h <= 128) MIT_FFT_LAUNCH(4, 256);
    else if (h <= 256) { if (nt256) MIT_FFT_LAUNCH(8, 256); else MIT_FFT_LAUNCH(4, 512); }
    else { if (nt256) MIT_FFT_LAUNCH(16, 256); else MIT_FFT_LAUNCH(8, 512); }
#undef MIT_FFT_LAUNCH
    MIT_CHECK_LAUNCH("mit_fft_cols");
    return 0;
}
