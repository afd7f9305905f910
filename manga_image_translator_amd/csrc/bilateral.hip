// bilateral.hip — cv2.bilateralFilter on 8-bit RGB pages, on the GPU.
//
// The reference smooths the page with cv2.bilateralFilter(img, 17, 80, 80) before the per-line DenseCRF of the mask
// refinement (manga_translator/mask_refinement/text_mask_utils.py:159) and before the DBNet detector
// (manga_translator/detection/default.py:64).  OpenCV's 8-bit path (imgproc/src/bilateral_filter.dispatch.cpp,
// bilateralFilter_8u + bilateralFilterInvoker_8u): radius = d / 2, the circular support {(i, j): sqrt(i^2 + j^2) <= radius}
// in row-major order, BORDER_REFLECT_101, weights space_weight[k] * color_weight[|db| + |dg| + |dr|] from two float tables
// ((float)exp(double)), fp32 sums in tap order, result cvRound(sum * (1 / wsum)).  The tables come from the host
// (imgproc.bilateral_tables) so the kernel and the numpy oracle consume identical floats; sums are unfused multiply-adds
// (OpenCV's own result depends on whether its build dispatches to FMA — parity with the real library is unpinned, it is
// installed nowhere this runs).
//
// One workgroup filters a 32 x 32 tile: the (32 + 2r)^2 source window is staged once in LDS as packed 0x00BBGGRR words
// (consecutive lanes read consecutive words: conflict-free), the 768-entry colour table sits beside it.  VALU/LDS-bound:
// 197 taps x ~20 instructions per pixel against 6 bytes of HBM traffic per pixel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr int TILE = 32;
constexpr int MAX_RADIUS = 16;
constexpr int MAX_TAPS = (2 * MAX_RADIUS + 1) * (2 * MAX_RADIUS + 1);

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// taps: per tap (dy << 16 | (dx & 0xffff)) and its spatial weight
__global__ __launch_bounds__(256) void bilateral_u8c3_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int H, int W,
                                                             int radius, int ntaps, const int *__restrict__ tap_ofs,
                                                             const float *__restrict__ tap_w, const float *__restrict__ color_w) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];
    const int win = TILE + 2 * radius;
    uint32_t *tile = lds_u32;                                        // [win][win]
    float *cw = reinterpret_cast<float *>(lds_u32 + win * win);       // [768]
    float *sw = cw + 768;                                             // [ntaps]
    int *so = reinterpret_cast<int *>(sw + ntaps);                    // [ntaps]  LDS word offset of the tap relative to the centre
    const int b = blockIdx.z;
    const uint8_t *sp = src + (int64_t)b * H * W * 3;
    uint8_t *dp = dst + (int64_t)b * H * W * 3;
    const int x0 = blockIdx.x * TILE - radius, y0 = blockIdx.y * TILE - radius;
    for (int i = threadIdx.x; i < win * win; i += 256) {
        const int ty = i / win, tx = i - ty * win;
        const int sy = reflect101(y0 + ty, H), sx = reflect101(x0 + tx, W);
        const uint8_t *p = sp + ((int64_t)sy * W + sx) * 3;
        tile[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
    for (int i = threadIdx.x; i < 768; i += 256) cw[i] = color_w[i];
    for (int i = threadIdx.x; i < ntaps; i += 256) {
        sw[i] = tap_w[i];
        const int o = tap_ofs[i];
        so[i] = (o >> 16) * win + (int)(int16_t)(o & 0xffff);
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;  // 8 rows of 32 per pass, 4 passes
#pragma unroll 1
    for (int pass = 0; pass < TILE / 8; ++pass) {
        const int ly = ly0 + 8 * pass;
        const int gx = blockIdx.x * TILE + lx, gy = blockIdx.y * TILE + ly;
        const int centre = (ly + radius) * win + lx + radius;
        const uint32_t c0 = tile[centre];
        const int r0 = c0 & 255, g0 = (c0 >> 8) & 255, b0 = (c0 >> 16) & 255;
        float sr = 0.f, sg = 0.f, sb = 0.f, ws = 0.f;
        for (int k = 0; k < ntaps; ++k) {
            const uint32_t v = tile[centre + so[k]];
            const int r = v & 255, g = (v >> 8) & 255, bb = (v >> 16) & 255;
            const float w = sw[k] * cw[abs(r - r0) + abs(g - g0) + abs(bb - b0)];
            sr += (float)r * w;
            sg += (float)g * w;
            sb += (float)bb * w;
            ws += w;
        }
        if (gx < W && gy < H) {
            const float inv = 1.f / ws;
            uint8_t *o = dp + ((int64_t)gy * W + gx) * 3;
            o[0] = (uint8_t)__float2int_rn(sr * inv);
            o[1] = (uint8_t)__float2int_rn(sg * inv);
            o[2] = (uint8_t)__float2int_rn(sb * inv);
        }
    }
}

}  // namespace

extern "C" int mit_bilateral_u8c3(const uint8_t *src_dev, uint8_t *dst_dev, int B, int H, int W, int radius, int ntaps,
                                  const int *tap_ofs_dev, const float *tap_w_dev, const float *color_w_dev, void *stream) {
    if (!src_dev || !dst_dev || !tap_ofs_dev || !tap_w_dev || !color_w_dev) return mit_set_error("mit_bilateral_u8c3: null pointer");
    if (src_dev == dst_dev) return mit_set_error("mit_bilateral_u8c3: in-place filtering is not supported");
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0) return mit_set_error("mit_bilateral_u8c3: bad shape");
    if (radius < 1 || radius > MAX_RADIUS || ntaps < 1 || ntaps > MAX_TAPS)
        return mit_set_error("mit_bilateral_u8c3: radius must be in [1, %d] (got %d, %d taps)", MAX_RADIUS, radius, ntaps);
    const int win = TILE + 2 * radius;
    const size_t smem = ((size_t)win * win + 768 + 2 * (size_t)ntaps) * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MitProbeScope probe("bilateral_u8c3_kernel", st, 6.0 * B * H * W, 8.0 * ntaps * (double)B * H * W);
    dim3 grid(mit_div_up(W, TILE), mit_div_up(H, TILE), B);
    hipLaunchKernelGGL(bilateral_u8c3_kernel, grid, dim3(256), smem, st, src_dev, dst_dev, H, W, radius, ntaps, tap_ofs_dev, tap_w_dev,
                       color_w_dev);
    MIT_CHECK_LAUNCH("mit_bilateral_u8c3");
    return 0;
}
