// ocr_dwconv.hip — the ConvNeXt blocks' depthwise convolutions (ConvNeXtBlock.dwconv + folded BatchNorm, manga_translator/ocr/model_48px.py:195-206)
// and their C entries.  A translation unit of its own because it is built WITH packed fp32 math (build.py PACKED_FP32_BY_DESIGN): the
// row-blocked kernel's inner loop is 3136 FMAs per thread on float4 values whose (x, y) / (z, w) halves and weight halves are natural
// register pairs — as v_pk_fma_f32 (no modifier, tests/test_build_flags.py) that is half the VALU instructions; every lane of a pair
// computes exactly the scalar fmaf it replaces, so results are bit-identical to the scalar form.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

// ---- depthwise k x k conv (stride 1, pad k/2) + per-channel scale/bias (conv bias + folded BN) ----
// w layout [k*k][C]; 4 channels per thread.
__global__ void dwconv_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ scale,
                              const float *__restrict__ bias, float *__restrict__ out, int B, int H, int W, int C4, int k) {
    const int64_t total = (int64_t)B * H * W * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int r = k / 2;
    const int C = C4 * 4;
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < k; ++ky) {
            const int yy = y + ky - r;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = x + kx - r;
                if (xx < 0 || xx >= W) continue;
                const float4 v = *reinterpret_cast<const float4 *>(in + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
                const float4 ww = *reinterpret_cast<const float4 *>(w + (int64_t)(ky * k + kx) * C + c4 * 4);
                acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y);
                acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
            }
        }
        const float4 s = *reinterpret_cast<const float4 *>(scale + c4 * 4);
        const float4 bb = *reinterpret_cast<const float4 *>(bias + c4 * 4);
        float4 o;
        o.x = acc.x * s.x + bb.x; o.y = acc.y * s.y + bb.y; o.z = acc.z * s.z + bb.z; o.w = acc.w * s.w + bb.w;
        *reinterpret_cast<float4 *>(out + (((int64_t)b * H + y) * W + x) * C + c4 * 4) = o;
    }
}


// ---- ragged depthwise conv: several [B_s, H_s, W_s, C] images concatenated along the pixel axis ----
// (the OCR chunks of a page group have different widths; their activations live back to back so that the
// pointwise convs run as ONE GEMM over all rows).  Each thread produces XT = 4 consecutive output columns of
// one row for 4 channels: per kernel row it loads K + 3 input float4 and K weight float4 for 4K float4-FMAs
// (0.6 loads per FMA instead of 2), channel-contiguous so every load instruction covers whole pixels.
// Accumulation order per output = (ky, kx) ascending with fmaf, identical to dwconv_kernel.
template <int K>
__global__ __launch_bounds__(256) void dwconv_ragged_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                             const float *__restrict__ scale, const float *__restrict__ bias,
                                                             float *__restrict__ out, const MitRaggedSeg *__restrict__ segs,
                                                             int nsegs, int C4, int64_t total_items) {
    constexpr int XT = 4;
    constexpr int R = K / 2;
    const int C = C4 * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (an XCD-contiguous block -> item mapping was measured here and rejected: the PMC pass shows 3.3 GB fetched per launch for 0.58 GB
    // of input, but those re-reads are served by the Infinity Cache and the kernel is bound by its L1 load count, not by HBM; giving
    // each XCD its own window of the tensor made it 12 % slower)
    const int vb = blockIdx.x;
    for (int64_t it = (int64_t)vb * blockDim.x + threadIdx.x; it < total_items; it += stride) {
        const int c4 = (int)(it % C4);
        const int64_t g = it / C4;  // (segment, image row, x group)
        int lo = 0, hi = nsegs - 1;
        while (lo < hi) {  // last segment whose first group index is <= g
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].group_start <= g) lo = mid; else hi = mid - 1;
        }
        const MitRaggedSeg sg = segs[lo];
        const int xgroups = (sg.W + XT - 1) / XT;
        const int64_t lg = g - sg.group_start;
        const int xg = (int)(lg % xgroups);
        const int64_t row = lg / xgroups;  // b * H + y
        const int y = (int)(row % sg.H);
        const int x0 = xg * XT;
        const float *ib = in + (sg.pixel_start + (row - y) * sg.W) * C + c4 * 4;  // image b, row 0
        f32x4_t acc[XT];
#pragma unroll
        for (int j = 0; j < XT; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yy = y + ky - R;
            if (yy < 0 || yy >= sg.H) continue;
            const float *rowp = ib + (int64_t)yy * sg.W * C;
            f32x4_t v[K + XT - 1];
#pragma unroll
            for (int j = 0; j < K + XT - 1; ++j) {
                const int xx = x0 + j - R;
                v[j] = (xx >= 0 && xx < sg.W) ? *reinterpret_cast<const f32x4_t *>(rowp + (int64_t)xx * C)
                                              : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4_t ww = *reinterpret_cast<const f32x4_t *>(w + (int64_t)(ky * K + kx) * C + c4 * 4);
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    // out-of-image taps must not enter the fmaf chain at all (dwconv_kernel skips them); adding
                    // ww * 0 is exact for finite ww, so the value is unchanged
                    acc[j].x = fmaf(v[j + kx].x, ww.x, acc[j].x);
                    acc[j].y = fmaf(v[j + kx].y, ww.y, acc[j].y);
                    acc[j].z = fmaf(v[j + kx].z, ww.z, acc[j].z);
                    acc[j].w = fmaf(v[j + kx].w, ww.w, acc[j].w);
                }
            }
        }
        const f32x4_t sc = *reinterpret_cast<const f32x4_t *>(scale + c4 * 4);
        const f32x4_t bb = *reinterpret_cast<const f32x4_t *>(bias + c4 * 4);
        float *ob = out + (sg.pixel_start + row * sg.W) * C + c4 * 4;
#pragma unroll
        for (int j = 0; j < XT; ++j) {
            if (x0 + j >= sg.W) break;
            f32x4_t o;
            o.x = acc[j].x * sc.x + bb.x; o.y = acc[j].y * sc.y + bb.y; o.z = acc[j].z * sc.z + bb.z; o.w = acc[j].w * sc.w + bb.w;
            *reinterpret_cast<f32x4_t *>(ob + (int64_t)(x0 + j) * C) = o;
        }
    }
}

// Row-blocked form for segments that all have the same height H with H % YT == 0 (the OCR stages: 24 / 12 / 6 rows): a thread
// produces YT output rows x XT = 4 columns x 4 channels, walking the YT + K - 1 input rows once — (K + 3) float4 loads per input row
// feed up to YT * K * 4 float4-FMAs instead of K * 4 (the per-row kernel above is bound by its L1 load count: 0.6 loads per FMA) —
// and the K x K x C weights sit in LDS (conflict-free: consecutive lanes = consecutive channel quads).  Per output the fmaf chain
// is still (ky, kx) ascending over the in-image rows, so the result is bit-identical to dwconv_ragged_kernel.
template <int K, int YT>
__global__ __launch_bounds__(256) void dwconv_ragged_rows_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                                  const float *__restrict__ scale, const float *__restrict__ bias,
                                                                  float *__restrict__ out, const MitRaggedSeg *__restrict__ segs,
                                                                  int nsegs, int C4, int H, int64_t total_items) {
    constexpr int XT = 4;
    constexpr int R = K / 2;
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [K * K][C]
    const int C = C4 * 4;
    for (int i = threadIdx.x; i < K * K * C4; i += blockDim.x)
        reinterpret_cast<f32x4_t *>(wl)[i] = reinterpret_cast<const f32x4_t *>(w)[i];
    __syncthreads();
    const int YG = H / YT;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // The accumulators start from a zero the compiler cannot see through: folded into the first packed FMA as an inline constant it
    // would need op_sel_hi on that operand — a modifier form, which this code base keeps out of its ISA.
    float zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total_items; it += stride) {
        const int c4 = (int)(it % C4);
        const int64_t g = it / C4;  // (segment, image, row group, x group); a segment's first index = group_start / YT (H % YT == 0)
        int lo = 0, hi = nsegs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].group_start / YT <= g) lo = mid; else hi = mid - 1;
        }
        const MitRaggedSeg sg = segs[lo];
        const int xgroups = (sg.W + XT - 1) / XT;
        const int64_t lg = g - sg.group_start / YT;
        const int xg = (int)(lg % xgroups);
        const int64_t rg = lg / xgroups;  // b * YG + yg
        const int yg = (int)(rg % YG);
        const int64_t b = rg / YG;
        const int x0 = xg * XT, y0 = yg * YT;
        const float *ib = in + (sg.pixel_start + b * (int64_t)H * sg.W) * C + c4 * 4;  // image b, row 0
        f32x4_t acc[YT][XT];
#pragma unroll
        for (int oy = 0; oy < YT; ++oy)
#pragma unroll
            for (int j = 0; j < XT; ++j) acc[oy][j] = f32x4_t{zero, zero, zero, zero};
#pragma unroll
        for (int dr = 0; dr < YT + K - 1; ++dr) {
            const int yy = y0 + dr - R;
            if (yy < 0 || yy >= H) continue;
            const float *rowp = ib + (int64_t)yy * sg.W * C;
            f32x4_t v[K + XT - 1];
#pragma unroll
            for (int j = 0; j < K + XT - 1; ++j) {
                const int xx = x0 + j - R;
                v[j] = (xx >= 0 && xx < sg.W) ? *reinterpret_cast<const f32x4_t *>(rowp + (int64_t)xx * C)
                                              : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int oy = 0; oy < YT; ++oy) {
                const int ky = dr - oy;  // input row y0 + dr - R is tap ky of output row y0 + oy
                if (ky < 0 || ky >= K) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4_t ww = *reinterpret_cast<const f32x4_t *>(wl + (ky * K + kx) * C + c4 * 4);
#pragma unroll
                    for (int j = 0; j < XT; ++j) {  // two packed FMAs: lanes (x, y) and (z, w), each lane the scalar fmaf(v, w, acc)
                        const f32x2_t lo = __builtin_elementwise_fma(f32x2_t{v[j + kx].x, v[j + kx].y}, f32x2_t{ww.x, ww.y}, f32x2_t{acc[oy][j].x, acc[oy][j].y});
                        const f32x2_t hi = __builtin_elementwise_fma(f32x2_t{v[j + kx].z, v[j + kx].w}, f32x2_t{ww.z, ww.w}, f32x2_t{acc[oy][j].z, acc[oy][j].w});
                        acc[oy][j] = f32x4_t{lo.x, lo.y, hi.x, hi.y};
                    }
                }
            }
        }
        const f32x4_t sc = *reinterpret_cast<const f32x4_t *>(scale + c4 * 4);
        const f32x4_t bb = *reinterpret_cast<const f32x4_t *>(bias + c4 * 4);
#pragma unroll
        for (int oy = 0; oy < YT; ++oy) {
            float *ob = out + (sg.pixel_start + (b * H + y0 + oy) * (int64_t)sg.W) * C + c4 * 4;
#pragma unroll
            for (int j = 0; j < XT; ++j) {
                if (x0 + j >= sg.W) break;
                f32x4_t o;
                o.x = acc[oy][j].x * sc.x + bb.x; o.y = acc[oy][j].y * sc.y + bb.y; o.z = acc[oy][j].z * sc.z + bb.z; o.w = acc[oy][j].w * sc.w + bb.w;
                *reinterpret_cast<f32x4_t *>(ob + (int64_t)(x0 + j) * C) = o;
            }
        }
    }
}

}  // namespace

extern "C" int mit_dwconv_nhwc(const float *in_dev, const float *w_dev, const float *scale_dev, const float *bias_dev,
                               float *out_dev, int B, int H, int W, int C, int k, void *stream) {
    if (!in_dev || !w_dev || !scale_dev || !bias_dev || !out_dev) return mit_set_error("mit_dwconv_nhwc: null pointer");
    if ((C & 3) || !(k & 1)) return mit_set_error("mit_dwconv_nhwc: C %% 4 == 0 and odd k required");
    const int64_t total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, w_dev, scale_dev,
                       bias_dev, out_dev, B, H, W, C / 4, k);
    MIT_CHECK_LAUNCH("mit_dwconv_nhwc");
    return 0;
}


extern "C" int mit_dwconv_nhwc_ragged(const float *in_dev, const float *w_dev, const float *scale_dev, const float *bias_dev,
                                      float *out_dev, const MitRaggedSeg *segs_dev, int nsegs, int64_t total_groups, int C, int k,
                                      void *stream) {
    if (!in_dev || !w_dev || !scale_dev || !bias_dev || !out_dev || !segs_dev) return mit_set_error("mit_dwconv_nhwc_ragged: null pointer");
    if ((C & 3) || nsegs <= 0 || total_groups < 0) return mit_set_error("mit_dwconv_nhwc_ragged: bad arguments");
    if (total_groups == 0) return 0;
    const int C4 = C / 4;
    const int64_t items = total_groups * C4;
    const dim3 grid(grid_for(items, 256) * 4 > 65535 ? 65535 : grid_for(items, 256) * 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: the activation read once and written once (a work item = one row group of up to 4 pixels, so this counts
    // the ragged right edge of each image as full groups: an upper bound within W % 4 of exact); FLOPs 2 k^2 per element
    const double elems = 4.0 * (double)total_groups * C;
    MitProbeScope probe(k == 7 ? "dwconv_ragged_kernel<7>" : k == 5 ? "dwconv_ragged_kernel<5>" : "dwconv_ragged_kernel<3>", s, 8.0 * elems, 2.0 * k * k * elems);
    switch (k) {
        case 3: hipLaunchKernelGGL(dwconv_ragged_kernel<3>, grid, block, 0, s, in_dev, w_dev, scale_dev, bias_dev, out_dev, segs_dev, nsegs, C4, items); break;
        case 5: hipLaunchKernelGGL(dwconv_ragged_kernel<5>, grid, block, 0, s, in_dev, w_dev, scale_dev, bias_dev, out_dev, segs_dev, nsegs, C4, items); break;
        case 7: hipLaunchKernelGGL(dwconv_ragged_kernel<7>, grid, block, 0, s, in_dev, w_dev, scale_dev, bias_dev, out_dev, segs_dev, nsegs, C4, items); break;
        default: return mit_set_error("mit_dwconv_nhwc_ragged: k must be 3, 5 or 7 (got %d)", k);
    }
    MIT_CHECK_LAUNCH("mit_dwconv_nhwc_ragged");
    return 0;
}

extern "C" int mit_dwconv_nhwc_ragged_rows(const float *in_dev, const float *w_dev, const float *scale_dev, const float *bias_dev,
                                           float *out_dev, const MitRaggedSeg *segs_dev, int nsegs, int64_t total_groups, int C, int k,
                                           int common_H, void *stream) {
    static const bool off = getenv("MIT_DWCONV_NO_ROWS") != nullptr;  // A/B knob for scripts/
    const int yt = (common_H > 0 && common_H % 4 == 0) ? 4 : (common_H > 0 && common_H % 2 == 0) ? 2 : 0;
    const size_t smem = (size_t)k * k * C * sizeof(float);
    if (off || yt == 0 || smem > 64 * 1024 || (k != 3 && k != 5 && k != 7))
        return mit_dwconv_nhwc_ragged(in_dev, w_dev, scale_dev, bias_dev, out_dev, segs_dev, nsegs, total_groups, C, k, stream);
    if (!in_dev || !w_dev || !scale_dev || !bias_dev || !out_dev || !segs_dev) return mit_set_error("mit_dwconv_nhwc_ragged_rows: null pointer");
    if ((C & 3) || nsegs <= 0 || total_groups < 0 || total_groups % yt) return mit_set_error("mit_dwconv_nhwc_ragged_rows: bad arguments");
    if (total_groups == 0) return 0;
    const int C4 = C / 4;
    const int64_t items = total_groups / yt * C4;
    const dim3 grid(grid_for(items, 256) * 4 > 65535 ? 65535 : grid_for(items, 256) * 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    const double elems = 4.0 * (double)total_groups * C;
    MitProbeScope probe(k == 7 ? "dwconv_ragged_kernel<7>" : k == 5 ? "dwconv_ragged_kernel<5>" : "dwconv_ragged_kernel<3>", s, 8.0 * elems, 2.0 * k * k * elems);
#define MIT_DWR(KK, YY) hipLaunchKernelGGL((dwconv_ragged_rows_kernel<KK, YY>), grid, block, smem, s, in_dev, w_dev, scale_dev, bias_dev, out_dev, segs_dev, nsegs, C4, common_H, items)
    switch (k * 10 + yt) {
        case 34: MIT_DWR(3, 4); break;
        case 32: MIT_DWR(3, 2); break;
        case 54: MIT_DWR(5, 4); break;
        case 52: MIT_DWR(5, 2); break;
        case 74: MIT_DWR(7, 4); break;
        default: MIT_DWR(7, 2); break;
    }
#undef MIT_DWR
    MIT_CHECK_LAUNCH("mit_dwconv_nhwc_ragged_rows");
    return 0;
}

