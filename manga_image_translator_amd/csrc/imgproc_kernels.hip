// imgproc_kernels.hip — 8-bit page resizes and the mask-select composite that bracket the inpainter for pages that are not
// already <= inpainting_size and a multiple of 8 (LamaMPEInpainter._infer, inpainting_lama_mpe.py:63-79,112-117), on the GPU so
// the page crosses PCIe once in each direction as bytes.  HBM-bound: every source byte is read about once (the four taps of
// neighbouring outputs hit the same cache lines), every destination byte written once.
//
// The arithmetic is OpenCV's, restated from its documented fixed-point rules (parity against the real library is unpinned —
// OpenCV is installed nowhere this runs; the oracle holds an independent restatement):
//   mode 0  cv2.INTER_LINEAR (8-bit path): separable bilinear, 11-bit coefficients, horizontal sums kept in 32 bits, vertical pass
//           (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
//   mode 1  exact 2x shrink, which OpenCV routes to the INTER_AREA box mean for INTER_LINEAR and INTER_LINEAR_EXACT alike:
//           (a + b + c + d + 2) >> 2
//   mode 2  cv2.INTER_LINEAR_EXACT (resize_bitExact, ufixedpoint16): 8.8 coefficients, horizontal c0*a + c1*b in 16 bits,
//           vertical (r0*h0 + r1*h1 + 32768) >> 16
// Coefficient tables come from the host (imgproc.py): per destination index the first source index and two 16-bit weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t *__restrict__ src, int H, int W, int C, uint8_t *__restrict__ dst, int dh,
                                                        int dw, int mode, const int *__restrict__ yidx, const uint16_t *__restrict__ ycoef,
                                                        const int *__restrict__ xidx, const uint16_t *__restrict__ xcoef, int B) {
    const int64_t total = (int64_t)B * dh * dw;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % dw);
        const int64_t r = i / dw;
        const int y = (int)(r % dh);
        const int b = (int)(r / dh);
        const uint8_t *p = src + (int64_t)b * H * W * C;
        uint8_t *o = dst + i * C;
        if (mode == 1) {
            const uint8_t *q = p + ((int64_t)(2 * y) * W + 2 * x) * C;
            for (int k = 0; k < C; ++k) o[k] = (uint8_t)((q[k] + q[C + k] + q[(int64_t)W * C + k] + q[(int64_t)W * C + C + k] + 2) >> 2);
            continue;
        }
        const int y0 = yidx[y], y1 = min(y0 + 1, H - 1), x0 = xidx[x], x1 = min(x0 + 1, W - 1);
        const uint32_t a0 = xcoef[2 * x], a1 = xcoef[2 * x + 1], b0 = ycoef[2 * y], b1 = ycoef[2 * y + 1];
        const uint8_t *p00 = p + ((int64_t)y0 * W + x0) * C, *p01 = p + ((int64_t)y0 * W + x1) * C;
        const uint8_t *p10 = p + ((int64_t)y1 * W + x0) * C, *p11 = p + ((int64_t)y1 * W + x1) * C;
        for (int k = 0; k < C; ++k) {
            const uint32_t r0 = p00[k] * a0 + p01[k] * a1;
            const uint32_t r1 = p10[k] * a0 + p11[k] * a1;
            uint32_t v;
            if (mode == 0) v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            else v = (b0 * r0 + b1 * r1 + 32768u) >> 16;
            o[k] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

// out = mask >= thr ? a : b, per pixel over C channels (the final composite of _infer :114-117 with the ORIGINAL mask)
__global__ __launch_bounds__(256) void select_u8_kernel(const uint8_t *__restrict__ mask, int thr, const uint8_t *__restrict__ a,
                                                        const uint8_t *__restrict__ b, uint8_t *__restrict__ out, int64_t npix, int C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const uint8_t *s = mask[i] >= thr ? a : b;
        for (int k = 0; k < C; ++k) out[i * C + k] = s[i * C + k];
    }
}

unsigned grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g > 1048576 ? 1048576 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mit_resize_u8(const uint8_t *src_dev, int B, int H, int W, int C, uint8_t *dst_dev, int dh, int dw, int mode,
                             const int *yidx_dev, const uint16_t *ycoef_dev, const int *xidx_dev, const uint16_t *xcoef_dev, void *stream) {
    if (!src_dev || !dst_dev) return mit_set_error("mit_resize_u8: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || dh <= 0 || dw <= 0 || C < 1 || C > 4) return mit_set_error("mit_resize_u8: bad shape (1 <= C <= 4)");
    if (mode < 0 || mode > 2) return mit_set_error("mit_resize_u8: mode must be 0 (INTER_LINEAR), 1 (2x box mean) or 2 (INTER_LINEAR_EXACT)");
    if (mode == 1 && (H != 2 * dh || W != 2 * dw)) return mit_set_error("mit_resize_u8: mode 1 needs an exact 2x shrink");
    if (mode != 1 && (!yidx_dev || !ycoef_dev || !xidx_dev || !xcoef_dev)) return mit_set_error("mit_resize_u8: missing tap tables");
    const int64_t total = (int64_t)B * dh * dw;
    MitProbeScope probe("resize_u8_kernel", (hipStream_t)stream, (double)B * C * ((double)H * W + (double)dh * dw));
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src_dev, H, W, C, dst_dev, dh, dw, mode,
                       yidx_dev, ycoef_dev, xidx_dev, xcoef_dev, B);
    MIT_CHECK_LAUNCH("mit_resize_u8");
    return 0;
}

extern "C" int mit_select_u8(const uint8_t *mask_dev, int thr, const uint8_t *a_dev, const uint8_t *b_dev, uint8_t *out_dev, int64_t npix,
                             int C, void *stream) {
    if (!mask_dev || !a_dev || !b_dev || !out_dev) return mit_set_error("mit_select_u8: null pointer");
    if (npix <= 0 || C < 1 || C > 4) return mit_set_error("mit_select_u8: bad size");
    hipLaunchKernelGGL(select_u8_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, mask_dev, thr, a_dev, b_dev, out_dev,
                       npix, C);
    MIT_CHECK_LAUNCH("mit_select_u8");
    return 0;
}
