// ctd_kernels.hip — memory-bound pieces of the text-detection stage (and generic NHWC helpers).
//
// Reference: manga_translator/detection/ctd.py (preprocess_img :17-28, postprocess_mask :30-44),
// ctd_utils/utils/imgproc_utils.py letterbox :69-100, yolov5/common.py SPPF :181-197,
// ctd_utils/basemodel.py double_conv_c3 (AvgPool2d) :28-39, utils/db_utils.py binarize :75.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

// ---- letterbox: u8 [B,H,W,3] --(cv2.resize INTER_LINEAR)--> [nh,nw] --pad--> fp32 NHWC [B,S,S,4] / 255 ----
// mode 0: copy (no resize); mode 1: exact 2x shrink = 2x2 box mean (a+b+c+d+2)>>2; mode 2: OpenCV's 11-bit
// fixed-point bilinear, taps from host tables (index + two short coefficients per destination index).
__global__ void ctd_prep_kernel(const uint8_t *__restrict__ img, int H, int W, int nh, int nw, int S, int mode,
                                const int *__restrict__ yidx, const short *__restrict__ ycoef,
                                const int *__restrict__ xidx, const short *__restrict__ xcoef, float4 *__restrict__ out,
                                int B) {
    const int64_t total = (int64_t)B * S * S;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int x = (int)(i % S);
        const int64_t r = i / S;
        const int y = (int)(r % S);
        const int b = (int)(r / S);
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (y < nh && x < nw) {
            const uint8_t *p = img + (int64_t)b * H * W * 3;
            int c[3];
            if (mode == 0) {
                for (int k = 0; k < 3; ++k) c[k] = p[((int64_t)y * W + x) * 3 + k];
            } else if (mode == 1) {
                const uint8_t *q = p + ((int64_t)(2 * y) * W + 2 * x) * 3;
                for (int k = 0; k < 3; ++k) c[k] = (q[k] + q[3 + k] + q[(int64_t)W * 3 + k] + q[(int64_t)W * 3 + 3 + k] + 2) >> 2;
            } else {
                const int y0 = yidx[y], y1 = min(y0 + 1, H - 1), x0 = xidx[x], x1 = min(x0 + 1, W - 1);
                const int a0 = xcoef[2 * x], a1 = xcoef[2 * x + 1], b0 = ycoef[2 * y], b1 = ycoef[2 * y + 1];
                for (int k = 0; k < 3; ++k) {
                    const int r0 = p[((int64_t)y0 * W + x0) * 3 + k] * a0 + p[((int64_t)y0 * W + x1) * 3 + k] * a1;
                    const int r1 = p[((int64_t)y1 * W + x0) * 3 + k] * a0 + p[((int64_t)y1 * W + x1) * 3 + k] * a1;
                    int o = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                    c[k] = o < 0 ? 0 : (o > 255 ? 255 : o);
                }
            }
            v.x = (float)c[0] / 255.0f;  // astype(np.float32) / 255  ctd.py:23
            v.y = (float)c[1] / 255.0f;
            v.z = (float)c[2] / 255.0f;
        }
        out[i] = v;
    }
}

// ---- NHWC max-pool k x k, stride 1, pad k/2 (-inf padding), 4 channels per thread ----
__global__ void maxpool_kernel(const float *__restrict__ in, int64_t in_ps, float *__restrict__ out, int64_t out_ps, int B,
                               int H, int W, int C4, int k) {
    const int64_t total = (int64_t)B * H * W * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int r = k / 2;
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        float4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const float4 v = *reinterpret_cast<const float4 *>(in + (((int64_t)b * H + yy) * W + xx) * in_ps + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4 *>(out + (((int64_t)b * H + y) * W + x) * out_ps + c4 * 4) = m;
    }
}

// ---- NHWC 2x2 average pool, stride 2 ----
__global__ void avgpool2_kernel(const float *__restrict__ in, int64_t in_ps, float *__restrict__ out, int64_t out_ps, int B,
                                int Ho, int Wo, int C4) {
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int Wi = Wo * 2, Hi = Ho * 2;
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float *base = in + (((int64_t)b * Hi + 2 * y) * Wi + 2 * x) * in_ps + c4 * 4;
        const float4 a = *reinterpret_cast<const float4 *>(base);
        const float4 bb = *reinterpret_cast<const float4 *>(base + in_ps);
        const float4 c = *reinterpret_cast<const float4 *>(base + (int64_t)Wi * in_ps);
        const float4 d = *reinterpret_cast<const float4 *>(base + (int64_t)Wi * in_ps + in_ps);
        float4 o;
        o.x = (a.x + bb.x + c.x + d.x) * 0.25f; o.y = (a.y + bb.y + c.y + d.y) * 0.25f;
        o.z = (a.z + bb.z + c.z + d.z) * 0.25f; o.w = (a.w + bb.w + c.w + d.w) * 0.25f;
        *reinterpret_cast<float4 *>(out + (((int64_t)b * Ho + y) * Wo + x) * out_ps + c4 * 4) = o;
    }
}

// ---- channel-slice copy (the one concat a producer cannot write in place) ----
__global__ void copy_channels_kernel(const float *__restrict__ in, int64_t in_ps, float *__restrict__ out, int64_t out_ps,
                                     int64_t npix, int C4) {
    const int64_t total = npix * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        const int64_t p = i / C4;
        *reinterpret_cast<float4 *>(out + p * out_ps + c4 * 4) = *reinterpret_cast<const float4 *>(in + p * in_ps + c4 * 4);
    }
}

// ---- fp32 map -> u8: mode 0 = (uint8)(v*255) truncation (postprocess_mask ctd.py:41-44), mode 1 = v > thr,
//      mode 2 = (uint8)(clip(v, 0, 1) * 255) (ESRGANUpscalerPytorch._infer, upscaling/esrgan_pytorch.py:545) ----
__global__ void map_to_u8_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, int64_t n, int mode, float thr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = in[i];
        if (mode == 0) out[i] = (uint8_t)(int)(v * 255.0f);
        else if (mode == 1) out[i] = (uint8_t)(v > thr ? 1 : 0);
        else out[i] = (uint8_t)(int)(fminf(fmaxf(v, 0.f), 1.f) * 255.0f);
    }
}

// ---- NHWC max-pool k x k, stride s, pad p (-inf padding) ----
__global__ void maxpool2d_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int H, int W, int C4, int Ho, int Wo,
                                 int k, int s, int p) {
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int C = C4 * 4;
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t q = i / C4;
        const int x = (int)(q % Wo);
        q /= Wo;
        const int y = (int)(q % Ho);
        const int b = (int)(q / Ho);
        float4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int ky = 0; ky < k; ++ky) {
            const int yy = y * s + ky - p;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = x * s + kx - p;
                if (xx < 0 || xx >= W) continue;
                const float4 v = *reinterpret_cast<const float4 *>(in + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4 *>(out + (((int64_t)b * Ho + y) * Wo + x) * C + c4 * 4) = m;
    }
}

// ---- out = a * x + y, elementwise float4 (RRDB's residual scaling, upscaling/esrgan_pytorch.py:112) ----
__global__ void axpy_kernel(float4 *__restrict__ out, float a, const float4 *__restrict__ x, const float4 *__restrict__ y, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const float4 xv = x[i], yv = y[i];
        float4 o;
        o.x = xv.x * a + yv.x; o.y = xv.y * a + yv.y; o.z = xv.z * a + yv.z; o.w = xv.w * a + yv.w;
        out[i] = o;
    }
}

}  // namespace

extern "C" int mit_maxpool2d_nhwc(const float *in_dev, float *out_dev, int B, int H, int W, int C, int k, int s, int p, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_maxpool2d_nhwc: null pointer");
    if ((C & 3) || k <= 0 || s <= 0 || p < 0 || 2 * p > k) return mit_set_error("mit_maxpool2d_nhwc: bad geometry");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    if (B <= 0 || Ho <= 0 || Wo <= 0) return mit_set_error("mit_maxpool2d_nhwc: empty output");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, B, H, W, C / 4,
                       Ho, Wo, k, s, p);
    MIT_CHECK_LAUNCH("mit_maxpool2d_nhwc");
    return 0;
}

extern "C" int mit_axpy(float *out_dev, float a, const float *x_dev, const float *y_dev, int64_t n, void *stream) {
    if (!out_dev || !x_dev || !y_dev) return mit_set_error("mit_axpy: null pointer");
    if ((n & 3) || ((reinterpret_cast<uintptr_t>(out_dev) | reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(y_dev)) & 15))
        return mit_set_error("mit_axpy: n %% 4 == 0 and 16-byte aligned pointers required");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4 *>(out_dev), a,
                       reinterpret_cast<const float4 *>(x_dev), reinterpret_cast<const float4 *>(y_dev), n / 4);
    MIT_CHECK_LAUNCH("mit_axpy");
    return 0;
}

extern "C" int mit_ctd_prep(const uint8_t *img_dev, int B, int H, int W, int nh, int nw, int S, int mode,
                            const int *yidx_dev, const short *ycoef_dev, const int *xidx_dev, const short *xcoef_dev,
                            float *out_dev, void *stream) {
    if (!img_dev || !out_dev) return mit_set_error("mit_ctd_prep: null pointer");
    if (mode < 0 || mode > 2) return mit_set_error("mit_ctd_prep: bad mode %d", mode);
    if (mode == 2 && (!yidx_dev || !ycoef_dev || !xidx_dev || !xcoef_dev)) return mit_set_error("mit_ctd_prep: missing resize tables");
    if (nh > S || nw > S || nh <= 0 || nw <= 0) return mit_set_error("mit_ctd_prep: bad letterbox size");
    if (mode == 1 && (H != 2 * nh || W != 2 * nw)) return mit_set_error("mit_ctd_prep: mode 1 needs an exact 2x shrink");
    if (mode == 0 && (H != nh || W != nw)) return mit_set_error("mit_ctd_prep: mode 0 needs equal sizes");
    const int64_t total = (int64_t)B * S * S;
    hipLaunchKernelGGL(ctd_prep_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img_dev, H, W, nh, nw,
                       S, mode, yidx_dev, ycoef_dev, xidx_dev, xcoef_dev, reinterpret_cast<float4 *>(out_dev), B);
    MIT_CHECK_LAUNCH("mit_ctd_prep");
    return 0;
}

extern "C" int mit_maxpool_nhwc(const float *in_dev, int64_t in_pixstride, float *out_dev, int64_t out_pixstride, int B,
                                int H, int W, int C, int k, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_maxpool_nhwc: null pointer");
    if ((C & 3) || (in_pixstride & 3) || (out_pixstride & 3) || !(k & 1)) return mit_set_error("mit_maxpool_nhwc: C/strides %% 4, odd k");
    const int64_t total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, in_pixstride,
                       out_dev, out_pixstride, B, H, W, C / 4, k);
    MIT_CHECK_LAUNCH("mit_maxpool_nhwc");
    return 0;
}

extern "C" int mit_avgpool2_nhwc(const float *in_dev, int64_t in_pixstride, float *out_dev, int64_t out_pixstride, int B,
                                 int Ho, int Wo, int C, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_avgpool2_nhwc: null pointer");
    if ((C & 3) || (in_pixstride & 3) || (out_pixstride & 3)) return mit_set_error("mit_avgpool2_nhwc: C/strides %% 4");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, in_pixstride,
                       out_dev, out_pixstride, B, Ho, Wo, C / 4);
    MIT_CHECK_LAUNCH("mit_avgpool2_nhwc");
    return 0;
}

extern "C" int mit_copy_channels(const float *in_dev, int64_t in_pixstride, float *out_dev, int64_t out_pixstride,
                                 int64_t npix, int C, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_copy_channels: null pointer");
    if ((C & 3) || (in_pixstride & 3) || (out_pixstride & 3)) return mit_set_error("mit_copy_channels: C/strides %% 4");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(npix * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, in_dev,
                       in_pixstride, out_dev, out_pixstride, npix, C / 4);
    MIT_CHECK_LAUNCH("mit_copy_channels");
    return 0;
}

extern "C" int mit_map_to_u8(const float *in_dev, uint8_t *out_dev, int64_t n, int mode, float thr, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_map_to_u8: null pointer");
    hipLaunchKernelGGL(map_to_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, n, mode, thr);
    MIT_CHECK_LAUNCH("mit_map_to_u8");
    return 0;
}
