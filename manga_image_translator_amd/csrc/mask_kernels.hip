// mask_kernels.hip — the tail of the mask refinement between OCR and inpainting on the device:
// per-line elliptical dilation of the CRF-refined component masks + their union, the final small dilation, binarisation
// (mask_refinement/text_mask_utils.py:178-195 cv2.dilate(..., getStructuringElement(MORPH_ELLIPSE, (k, k))) per line, :194 the closing
// dilate; mask_refinement/__init__.py:28-29 the > 0 -> 255 after the resize back to page size).  On the host this was 50 ms of a
// 77 ms page (scipy maximum_filter with a 21 x 21 footprint over 32 line windows); here it is one launch per step.
//
// cv2.dilate / scipy.ndimage.maximum_filter(footprint, mode='constant', cval=0) with an odd k x k ellipse:
//   dst(y, x) = max over rows i of the footprint, columns |j - c| <= hw(i), of src(y + i - c, x + j - c),  c = k / 2,
//   hw(i) = rint(c * sqrt((c^2 - (i - c)^2) / c^2))  evaluated in double (cv2.getStructuringElement's MORPH_ELLIPSE rows),
// source pixels outside the job's source rectangle (and, as the host path dilates a sub-array, outside its window) count as 0.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr int MAX_K = 255;

// one block = 64 x 4 pixels of one job's window
__global__ __launch_bounds__(256) void mask_dilate_jobs_kernel(const uint8_t *__restrict__ src, const MitDilateJob *__restrict__ jobs, uint8_t *dst,
                                                               const int W, const int merge) {
    const MitDilateJob jb = jobs[blockIdx.z];
    __shared__ int hw[MAX_K];
    const int k = jb.k, c = k >> 1;
    for (int i = threadIdx.x; i < k; i += 256) {
        const int dy = i - c;
        hw[i] = c ? (int)rint((double)c * sqrt((double)(c * c - dy * dy) / (double)(c * c))) : 0;
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= jb.dw || y >= jb.dh) return;
    const int px = jb.dx + x, py = jb.dy + y;  // page coordinates of this output pixel
    // source pixels that count: inside the source rectangle and inside the window
    const int x_lo = max(jb.sx, jb.dx), x_hi = min(jb.sx + jb.sw, jb.dx + jb.dw) - 1;
    const int y_lo = max(jb.sy, jb.dy), y_hi = min(jb.sy + jb.sh, jb.dy + jb.dh) - 1;
    int best = 0;
    for (int i = 0; i < k && best < 255; ++i) {
        const int sy = py + i - c;
        if (sy < y_lo || sy > y_hi) continue;
        const int h = hw[i];
        const int xa = max(px - h, x_lo), xb = min(px + h, x_hi);
        const uint8_t *row = src + jb.src_off + (int64_t)(sy - jb.sy) * jb.spitch - jb.sx;
        for (int sx = xa; sx <= xb; ++sx) {
            const int v = row[sx];
            if (v > best) {
                best = v;
                if (best == 255) break;
            }
        }
    }
    uint8_t *o = dst + (int64_t)py * W + px;
    if (!merge) *o = (uint8_t)best;
    else if (best) *o = (uint8_t)best;  // union of {0, 255} masks: concurrent writers store the same byte
}

__global__ __launch_bounds__(256) void binarize_u8_kernel(uint8_t *p, const int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (p[i]) p[i] = 255;
}

}  // namespace

extern "C" int mit_mask_dilate_jobs(const uint8_t *src_dev, const MitDilateJob *jobs_host, int n_jobs, uint8_t *dst_dev, int H, int W, int merge,
                                    MitDilateJob *jobs_dev, void *stream) {
    if (n_jobs == 0) return 0;
    if (!src_dev || !jobs_host || !dst_dev || !jobs_dev) return mit_set_error("mit_mask_dilate_jobs: null pointer");
    if (n_jobs < 0 || n_jobs > 65535 || H <= 0 || W <= 0) return mit_set_error("mit_mask_dilate_jobs: bad size");
    int mw = 0, mh = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const MitDilateJob &b = jobs_host[j];
        if (b.k < 1 || b.k > MAX_K || !(b.k & 1)) return mit_set_error("mit_mask_dilate_jobs: job %d: k = %d must be odd and <= %d", j, b.k, MAX_K);
        if (b.dw <= 0 || b.dh <= 0 || b.dx < 0 || b.dy < 0 || b.dx + b.dw > W || b.dy + b.dh > H)
            return mit_set_error("mit_mask_dilate_jobs: job %d: window outside the page", j);
        if (b.sw <= 0 || b.sh <= 0 || b.spitch < b.sw || b.src_off < 0) return mit_set_error("mit_mask_dilate_jobs: job %d: bad source rectangle", j);
        mw = b.dw > mw ? b.dw : mw;
        mh = b.dh > mh ? b.dh : mh;
    }
    hipStream_t st = (hipStream_t)stream;
    MIT_CHECK_HIP(hipMemcpyAsync(jobs_dev, jobs_host, (size_t)n_jobs * sizeof(MitDilateJob), hipMemcpyHostToDevice, st));
    MitProbeScope probe("mask_dilate_jobs_kernel", st, 0.0);
    hipLaunchKernelGGL(mask_dilate_jobs_kernel, dim3(mit_div_up(mw, 64), mit_div_up(mh, 4), n_jobs), dim3(256), 0, st, src_dev, jobs_dev, dst_dev, W,
                       merge);
    MIT_CHECK_LAUNCH("mit_mask_dilate_jobs");
    return 0;
}

extern "C" int mit_binarize_u8(uint8_t *buf_dev, int64_t n, void *stream) {
    if (!buf_dev || n <= 0) return mit_set_error("mit_binarize_u8: bad argument");
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(binarize_u8_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, (hipStream_t)stream, buf_dev, n);
    MIT_CHECK_LAUNCH("mit_binarize_u8");
    return 0;
}
