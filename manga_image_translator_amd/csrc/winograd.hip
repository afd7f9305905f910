// Winograd F(4x4, 3x3) transforms for stride-1 3x3 convolutions on NHWC fp32 (Lavin & Gray, "Fast Algorithms for
// Convolutional Neural Networks", 2015; interpolation points 0, +-1, +-2, inf).
//
//   V_z = (B^T d B)_z        6x6 input patch d of every 4x4 output tile          -> mit_wino43_input
//   M_z = V_z @ U_z          36 independent [T x Cin] @ [Cin x Cout] products     -> mit_conv_gemm, Z = 36
//   y   = A^T m A            4x4 outputs per tile + scale / bias / act / residual -> mit_wino43_output
//
// 2.25 multiplications per output instead of 9.  Both transforms are pure HBM streaming kernels (one thread per
// (tile, channel pair), channels fastest so every access is a contiguous run of the NHWC / [T][C] rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

// consecutive workgroup ids are dealt round-robin to the 8 XCDs (one L2 each): give every XCD one contiguous run of tiles,
// so that the 6x6 patches of neighbouring tiles (2 shared rows / columns) meet in the same L2 instead of being fetched twice
__device__ __forceinline__ int64_t xcd_contiguous_block(int64_t bid, int64_t nwg) {
    const int64_t xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ int wino_src_index(int i, int n, int pad_mode) {
    // padded coordinate -> source row/column; -1 = contributes zero.  Coordinates past the far border only feed outputs
    // that are never stored (partial tiles); they are clamped so that the transform sees values of ordinary magnitude.
    if (pad_mode == MIT_PAD_REFLECT) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
        return i < 0 ? 0 : i;
    }
    return (i < 0 || i >= n) ? -1 : i;
}

// B^T (rows): [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(float2 &d0, float2 &d1, float2 &d2, float2 &d3, float2 &d4, float2 &d5) {
#define MIT_BT(c)                                              \
    {                                                          \
        const float a0 = d0.c, a1 = d1.c, a2 = d2.c, a3 = d3.c, a4 = d4.c, a5 = d5.c; \
        d0.c = 4.f * a0 - 5.f * a2 + a4;                       \
        d1.c = (a3 + a4) - 4.f * (a1 + a2);                    \
        d2.c = 4.f * (a1 - a2) + (a4 - a3);                    \
        d3.c = 2.f * (a3 - a1) + (a4 - a2);                    \
        d4.c = 2.f * (a1 - a3) + (a4 - a2);                    \
        d5.c = 4.f * a1 - 5.f * a3 + a5;                       \
    }
    MIT_BT(x)
    MIT_BT(y)
#undef MIT_BT
}

__global__ __launch_bounds__(256) void wino43_input_kernel(const float *__restrict__ x, int64_t x_bs, int64_t x_ys, int64_t x_xs,
                                                           float *__restrict__ v, int B, int H, int W, int C2, int th, int tw,
                                                           int pad_mode, int64_t total, int64_t zstride) {
    const int64_t idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c2 = (int)(idx % C2);
    const int64_t tile = idx / C2;
    const int tx = (int)(tile % tw);
    const int ty = (int)((tile / tw) % th);
    const int b = (int)(tile / ((int64_t)tw * th));
    const float *xb = x + (int64_t)b * x_bs + c2 * 2;
    int64_t coff[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ix = wino_src_index(tx * 4 - 1 + s, W, pad_mode);
        coff[s] = ix < 0 ? -1 : (int64_t)ix * x_xs;
    }
    float2 d[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int iy = wino_src_index(ty * 4 - 1 + r, H, pad_mode);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float2 t = {0.f, 0.f};
            if (iy >= 0 && coff[s] >= 0) t = *reinterpret_cast<const float2 *>(xb + (int64_t)iy * x_ys + coff[s]);
            d[r][s] = t;
        }
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) bt6(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);  // B^T d
#pragma unroll
    for (int r = 0; r < 6; ++r) bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]);  // (B^T d) B
    float *vo = v + tile * (int64_t)(C2 * 2) + c2 * 2;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int s = 0; s < 6; ++s) *reinterpret_cast<float2 *>(vo + (int64_t)(r * 6 + s) * zstride) = d[r][s];
}

// A^T (rows): [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                    float &y0, float &y1, float &y2, float &y3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = (m0 + s12) + s34;
    y1 = d12 + 2.f * d34;
    y2 = s12 + 4.f * s34;
    y3 = (d12 + 8.f * d34) + m5;
}

__device__ __forceinline__ float wino_act(float v, int act, float alpha) {
    if (act == MIT_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MIT_ACT_LEAKY) return v > 0.f ? v : v * alpha;
    return v;
}

__global__ __launch_bounds__(256) void wino43_output_kernel(const float *__restrict__ m, float *__restrict__ y, int64_t y_bs,
                                                            int64_t y_ys, int64_t y_xs, const float *__restrict__ post, int64_t p_bs,
                                                            int64_t p_ys, int64_t p_xs, const float *__restrict__ scale,
                                                            const float *__restrict__ bias, int B, int H, int W, int N2, int th,
                                                            int tw, int act, float alpha, int64_t total, int64_t zstride) {
    const int64_t idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n2 = (int)(idx % N2);
    const int64_t tile = idx / N2;
    const int tx = (int)(tile % tw);
    const int ty = (int)((tile / tw) % th);
    const int b = (int)(tile / ((int64_t)tw * th));
    const float *mi = m + tile * (int64_t)(N2 * 2) + n2 * 2;
    float2 t[4][6];  // A^T m, column by column
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        float2 c[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) c[r] = *reinterpret_cast<const float2 *>(mi + (int64_t)(r * 6 + s) * zstride);
        at6(c[0].x, c[1].x, c[2].x, c[3].x, c[4].x, c[5].x, t[0][s].x, t[1][s].x, t[2][s].x, t[3][s].x);
        at6(c[0].y, c[1].y, c[2].y, c[3].y, c[4].y, c[5].y, t[0][s].y, t[1][s].y, t[2][s].y, t[3][s].y);
    }
    const float2 sc = scale ? *reinterpret_cast<const float2 *>(scale + n2 * 2) : float2{1.f, 1.f};
    const float2 bi = bias ? *reinterpret_cast<const float2 *>(bias + n2 * 2) : float2{0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int oy = ty * 4 + a;
        float2 o[4];
        at6(t[a][0].x, t[a][1].x, t[a][2].x, t[a][3].x, t[a][4].x, t[a][5].x, o[0].x, o[1].x, o[2].x, o[3].x);
        at6(t[a][0].y, t[a][1].y, t[a][2].y, t[a][3].y, t[a][4].y, t[a][5].y, o[0].y, o[1].y, o[2].y, o[3].y);
        if (oy >= H) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = tx * 4 + e;
            if (ox >= W) continue;
            float2 r;
            r.x = wino_act(o[e].x * sc.x + bi.x, act, alpha);
            r.y = wino_act(o[e].y * sc.y + bi.y, act, alpha);
            if (post) {
                const float2 pv = *reinterpret_cast<const float2 *>(post + (int64_t)b * p_bs + (int64_t)oy * p_ys + (int64_t)ox * p_xs + n2 * 2);
                r.x += pv.x;
                r.y += pv.y;
            }
            *reinterpret_cast<float2 *>(y + (int64_t)b * y_bs + (int64_t)oy * y_ys + (int64_t)ox * y_xs + n2 * 2) = r;
        }
    }
}

}  // namespace

extern "C" int mit_wino43_input(const float *x_dev, int64_t x_bs, int64_t x_ys, int64_t x_xs, float *v_dev, int B, int H, int W, int C,
                                int pad_mode, void *stream) {
    if (!x_dev || !v_dev) return mit_set_error("mit_wino43_input: null pointer");
    if (B <= 0 || H < 2 || W < 2 || C <= 0 || (C & 1)) return mit_set_error("mit_wino43_input: bad shape (H, W >= 2, even C)");
    if ((x_bs & 1) || (x_ys & 1) || (x_xs & 1) || (reinterpret_cast<uintptr_t>(x_dev) & 7) || (reinterpret_cast<uintptr_t>(v_dev) & 7))
        return mit_set_error("mit_wino43_input: strides / bases must be multiples of 2 floats");
    if (pad_mode != MIT_PAD_ZERO && pad_mode != MIT_PAD_REFLECT) return mit_set_error("mit_wino43_input: bad pad mode");
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int64_t T = (int64_t)B * th * tw, total = T * (C / 2);
    const int64_t nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffLL) return mit_set_error("mit_wino43_input: problem too large");
    // algorithmic bytes: the input read once + the 36/16-times larger transformed tensor written once
    MitProbeScope probe("wino43_input_kernel", (hipStream_t)stream, 4.0 * ((double)B * H * W * C + 36.0 * (double)T * C));
    hipLaunchKernelGGL(wino43_input_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x_dev, x_bs, x_ys,
                       x_xs, v_dev, B, H, W, C / 2, th, tw, pad_mode, total, T * C);
    MIT_CHECK_LAUNCH("mit_wino43_input");
    return 0;
}

extern "C" int mit_wino43_output(const float *m_dev, float *y_dev, int64_t y_bs, int64_t y_ys, int64_t y_xs, const float *post_dev,
                                 int64_t p_bs, int64_t p_ys, int64_t p_xs, const float *scale_dev, const float *bias_dev, int B, int H,
                                 int W, int N, int act, float alpha, void *stream) {
    if (!m_dev || !y_dev) return mit_set_error("mit_wino43_output: null pointer");
    if (B <= 0 || H < 2 || W < 2 || N <= 0 || (N & 1)) return mit_set_error("mit_wino43_output: bad shape (even N)");
    if ((y_bs & 1) || (y_ys & 1) || (y_xs & 1) || (p_bs & 1) || (p_ys & 1) || (p_xs & 1) || (reinterpret_cast<uintptr_t>(y_dev) & 7) ||
        (reinterpret_cast<uintptr_t>(m_dev) & 7) || (reinterpret_cast<uintptr_t>(post_dev) & 7))
        return mit_set_error("mit_wino43_output: strides / bases must be multiples of 2 floats");
    if (act != MIT_ACT_NONE && act != MIT_ACT_RELU && act != MIT_ACT_LEAKY) return mit_set_error("mit_wino43_output: act must be none / relu / leaky");
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int64_t T = (int64_t)B * th * tw, total = T * (N / 2);
    const int64_t nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffLL) return mit_set_error("mit_wino43_output: problem too large");
    // algorithmic bytes: the 36 products read once + the output written once (+ the residual read once)
    MitProbeScope probe("wino43_output_kernel", (hipStream_t)stream,
                        4.0 * (36.0 * (double)T * N + (double)B * H * W * N * (post_dev ? 2.0 : 1.0)));
    hipLaunchKernelGGL(wino43_output_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, m_dev, y_dev, y_bs,
                       y_ys, y_xs, post_dev, p_bs, p_ys, p_xs, scale_dev, bias_dev, B, H, W, N / 2, th, tw, act, alpha, total, T * N);
    MIT_CHECK_LAUNCH("mit_wino43_output");
    return 0;
}
