// ocr_warp.hip — text-line rectification for the OCR stage: every quadrilateral of a chunk is
// warped out of its page straight into the zero-padded uint8 chunk tensor [N,48,Wp,3] that the
// recogniser consumes (HBM-bound gather: ~4 source bytes read per output byte, one write).
//
// Reference: Quadrilateral.get_transformed_region (manga_translator/utils/generic.py:445-481):
//   cv2.warpPerspective(img_croped, M, (w, h))  [+ cv2.rotate(ROTATE_90_COUNTERCLOCKWISE) for 'v']
// and the chunk packing of Model48pxOCR._infer (manga_translator/ocr/model_48px.py:83-91).
//
// The arithmetic follows OpenCV's WarpPerspectiveInvoker + remapBilinear for 8-bit images
// (imgproc/src/imgwarp.cpp): coordinates in double, scaled by INTER_TAB_SIZE = 32 and rounded to
// nearest-even; 5 fractional bits select the bilinear weights, which for the 1/32 grid are the
// exact integers (32-fx)*(32-fy)*32 ... (sum 2^15); result = (sum + 2^14) >> 15; BORDER_CONSTANT 0.

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr int INTER_BITS = 5;
constexpr int INTER_TAB_SIZE = 1 << INTER_BITS;

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__global__ __launch_bounds__(256) void ocr_warp_kernel(const uint8_t *__restrict__ pages, int64_t page_stride, int W,
                                                        const MitWarpLine *__restrict__ lines, uint8_t *__restrict__ out,
                                                        int Hout, int Wp) {
    const MitWarpLine ln = lines[blockIdx.y];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= Hout * Wp) return;
    const int oy = pix / Wp, ox = pix - oy * Wp;
    uint8_t *dst = out + (((int64_t)ln.out_row * Hout + oy) * Wp + ox) * 3;
    // destination pixel of the un-rotated warp
    int x, y;
    bool inside;
    if (ln.vertical) {  // rotate 90 CCW: out(y', x') = region(y = x', x = dw - 1 - y')
        x = ln.dw - 1 - oy;
        y = ox;
        inside = oy < ln.dw && ox < ln.dh;
    } else {
        x = ox;
        y = oy;
        inside = oy < ln.dh && ox < ln.dw;
    }
    if (!inside) {  // chunk padding (np.zeros, model_48px.py:87)
        dst[0] = dst[1] = dst[2] = 0;
        return;
    }
    // WarpPerspectiveInvoker: blocks of bw0 columns; X0 is evaluated at the block's first column
    const int bh0 = ln.dh < 32 ? ln.dh : 32;
    int bw0 = 4096 / bh0;
    if (bw0 > ln.dw) bw0 = ln.dw;
    const int xb = (x / bw0) * bw0, x1 = x - xb;
    const double *M = ln.minv;
    const double X0 = M[0] * xb + M[1] * y + M[2];
    const double Y0 = M[3] * xb + M[4] * y + M[5];
    const double W0 = M[6] * xb + M[7] * y + M[8];
    double Wd = W0 + M[6] * x1;
    Wd = Wd != 0.0 ? (double)INTER_TAB_SIZE / Wd : 0.0;
    const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + M[0] * x1) * Wd));
    const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + M[3] * x1) * Wd));
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);  // cvRound
    const int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
    const int ax = X & (INTER_TAB_SIZE - 1), ay = Y & (INTER_TAB_SIZE - 1);
    const int w00 = (INTER_TAB_SIZE - ax) * (INTER_TAB_SIZE - ay) * 32, w01 = ax * (INTER_TAB_SIZE - ay) * 32;
    const int w10 = (INTER_TAB_SIZE - ax) * ay * 32, w11 = ax * ay * 32;
    const uint8_t *src = pages + (int64_t)ln.page * page_stride + ((int64_t)ln.y1 * W + ln.x1) * 3;
    const int cw = ln.cw, ch = ln.ch;
    if (sx >= cw || sx + 1 < 0 || sy >= ch || sy + 1 < 0) {
        dst[0] = dst[1] = dst[2] = 0;
        return;
    }
    const bool x0ok = sx >= 0, x1ok = sx + 1 < cw, y0ok = sy >= 0, y1ok = sy + 1 < ch;
    const uint8_t *p00 = src + ((int64_t)sy * W + sx) * 3;
    const uint8_t *p10 = p00 + (int64_t)W * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v00 = (x0ok && y0ok) ? p00[c] : 0, v01 = (x1ok && y0ok) ? p00[3 + c] : 0;
        const int v10 = (x0ok && y1ok) ? p10[c] : 0, v11 = (x1ok && y1ok) ? p10[3 + c] : 0;
        const int s = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
        dst[c] = (uint8_t)((s + (1 << 14)) >> 15);
    }
}

}  // namespace

extern "C" int mit_ocr_warp_lines(const uint8_t *pages_dev, int H, int W, const MitWarpLine *lines_dev, int n_lines,
                                  uint8_t *out_dev, int Hout, int Wp, void *stream) {
    if (!pages_dev || !lines_dev || !out_dev) return mit_set_error("mit_ocr_warp_lines: null pointer");
    if (H <= 0 || W <= 0 || Hout <= 0 || Wp <= 0 || n_lines < 0) return mit_set_error("mit_ocr_warp_lines: bad size");
    if (n_lines == 0) return 0;
    if (n_lines > 65535) return mit_set_error("mit_ocr_warp_lines: more than 65535 lines in one call");
    dim3 grid(mit_div_up((int64_t)Hout * Wp, 256), n_lines);
    hipLaunchKernelGGL(ocr_warp_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pages_dev,
                       (int64_t)H * W * 3, W, lines_dev, out_dev, Hout, Wp);
    MIT_CHECK_LAUNCH("mit_ocr_warp_lines");
    return 0;
}
