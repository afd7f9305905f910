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
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr int CT = 32;     // columns per workgroup
constexpr int TG = 8;      // thread groups (rows in flight) per column: 256 threads = 32 x 8

__global__ __launch_bounds__(256) void fft_cols_kernel(const float *__restrict__ in, int64_t in_bs, int64_t in_ts, int64_t in_hs,
                                                        float *__restrict__ out, int64_t out_bs, int64_t out_ts, int64_t out_hs,
                                                        const float2 *__restrict__ tw, int h, int logh, int64_t ncols,
                                                        int inverse, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *re = lds;               // [h][CT]
    float *im = lds + h * CT;
    const int col = threadIdx.x & (CT - 1), g = threadIdx.x / CT;
    const int64_t c0 = (int64_t)blockIdx.x * CT + col;
    const bool live = c0 < ncols;
    const float *ib = in + (int64_t)blockIdx.y * in_bs + c0;
    float *ob = out + (int64_t)blockIdx.y * out_bs + c0;
    // load with bit-reversed row index (decimation in time)
    for (int r = g; r < h; r += TG) {
        const int rr = __brev((unsigned)r) >> (32 - logh);
        float a = 0.f, b = 0.f;
        if (live) {
            a = ib[(int64_t)r * in_hs];
            b = ib[in_ts + (int64_t)r * in_hs];
        }
        re[rr * CT + col] = a;
        im[rr * CT + col] = b;
    }
    __syncthreads();
    const int nb = h >> 1;  // butterflies per column
    for (int s = 0; s < logh; ++s) {
        const int half = 1 << s;
        const int tstep = nb >> s;  // twiddle index stride: k * h / (2 * half)
        for (int j = g; j < nb; j += TG) {
            const int k = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + k, i1 = i0 + half;
            const float2 w = tw[k * tstep];  // (cos, sin) of 2 pi k tstep / h
            const float wr = w.x, wi = inverse ? w.y : -w.y;
            const float xr = re[i1 * CT + col], xi = im[i1 * CT + col];
            const float tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
            const float ur = re[i0 * CT + col], ui = im[i0 * CT + col];
            re[i0 * CT + col] = ur + tr;
            im[i0 * CT + col] = ui + ti;
            re[i1 * CT + col] = ur - tr;
            im[i1 * CT + col] = ui - ti;
        }
        __syncthreads();
    }
    if (live) {
        for (int r = g; r < h; r += TG) {
            ob[(int64_t)r * out_hs] = re[r * CT + col] * scale;
            ob[out_ts + (int64_t)r * out_hs] = im[r * CT + col] * scale;
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
    const size_t smem = (size_t)2 * h * CT * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && smem > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(mit_div_up(ncols, CT), B), block(256);
    hipLaunchKernelGGL(fft_cols_kernel, grid, block, smem, reinterpret_cast<hipStream_t>(stream), in_dev, in_bs, in_ts, in_hs, out_dev,
                       out_bs, out_ts, out_hs, reinterpret_cast<const float2 *>(twiddle_dev), h, logh, ncols, inverse, scale);
    MIT_CHECK_LAUNCH("mit_fft_cols");
    return 0;
}
