// ocr_kernels.hip — non-GEMM kernels of the 48px OCR stage (ConvNeXt backbone glue, XPOS
// transformer pieces, beam-search bookkeeping).  All fp32; the dense projections run on
// mit_conv_gemm.
//
// Reference: manga_translator/ocr/model_48px.py (ConvNeXtBlock :203-214, XposMultiheadAttention
// :327-394, decoder_forward :548-572, infer_beam_batch_tensor :678-801) and
// ocr/xpos_relative_position.py (:9-71).

#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mit_hip.h"
#include "common.h"
#include "ocr_kernels.h"
#include "bf16_split.h"
#include "ln_rows8.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef MIT_CONV_EXPERIMENTS   // phase stamps of workgroup (0, 0) of the cross-attention kernel (100 MHz clock): scripts/dev only
__device__ unsigned long long g_att_stamps[32];
#define MIT_ATT_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_att_stamps[i] = wall_clock64(); g_att_stamps[8 + (i)] = clock64(); } } while (0)
#define MIT_ATT_STAMP2(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_att_stamps[16 + (i)] = wall_clock64(); } while (0)
extern "C" int mit_dev_att_stamps(unsigned long long *out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_att_stamps), sizeof(g_att_stamps)) == hipSuccess ? 0 : 1; }
#else
#define MIT_ATT_STAMP(i) do { } while (0)
#define MIT_ATT_STAMP2(i) do { } while (0)
#endif

namespace {

// eight consecutive fp32 values (16-byte aligned, LDS) -> the three cells of (k-cell k8, row) of a planar output
__device__ __forceinline__ void store_cells(const OcrPlanes &o, const int k8, const int64_t row, const float *v8) {
    mitcg::u32x4 h, m, l;
    mitcg::split8(*reinterpret_cast<const f32x4 *>(v8), *reinterpret_cast<const f32x4 *>(v8 + 4), h, m, l);
    mitcg::u32x4 *dst = reinterpret_cast<mitcg::u32x4 *>(o.p) + (int64_t)k8 * o.ld + row;
    const int64_t plane = (int64_t)o.K8 * o.ld;
    dst[0] = h;
    dst[plane] = m;
    dst[2 * plane] = l;
}
__device__ __forceinline__ void wave_lds_fence() {  // a wave's own LDS accesses complete in order: wait for them, keep the compiler from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

// ---- u8 line crops [N,48,Wp,3] -> fp32 NHWC [N,48,Wp,4] = ((x - 127.5) / 127.5, 0) (:115) ----
__global__ void ocr_prep_kernel(const uint8_t *__restrict__ in, float4 *__restrict__ out, int64_t npix) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < npix; i += stride) {
        float4 v;
        v.x = ((float)in[3 * i + 0] - 127.5f) / 127.5f;
        v.y = ((float)in[3 * i + 1] - 127.5f) / 127.5f;
        v.z = ((float)in[3 * i + 2] - 127.5f) / 127.5f;
        v.w = 0.f;
        out[i] = v;
    }
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
            for (int j = 0; j < XT; ++j) acc[oy][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
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
                    for (int j = 0; j < XT; ++j) {
                        acc[oy][j].x = fmaf(v[j + kx].x, ww.x, acc[oy][j].x);
                        acc[oy][j].y = fmaf(v[j + kx].y, ww.y, acc[oy][j].y);
                        acc[oy][j].z = fmaf(v[j + kx].z, ww.z, acc[oy][j].z);
                        acc[oy][j].w = fmaf(v[j + kx].w, ww.w, acc[oy][j].w);
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

// ---- general NHWC average pool (count_include_pad = True, the nn.AvgPool2d default): the FAN backbone's
// AvgPool2d(2, stride=(2, 1), padding=(0, 1)) (ocr/model_48px_ctc.py:303) ----
__global__ void avgpool_general_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int H, int W, int C4, int Ho,
                                       int Wo, int kh, int kw, int sh, int sw, int ph, int pw) {
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int C = C4 * 4;
    const float inv = 1.0f / (float)(kh * kw);
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < kh; ++ky) {
            const int yy = y * sh + ky - ph;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int xx = x * sw + kx - pw;
                if (xx < 0 || xx >= W) continue;
                acc += *reinterpret_cast<const f32x4_t *>(in + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
            }
        }
        *reinterpret_cast<f32x4_t *>(out + (((int64_t)b * Ho + y) * Wo + x) * C + c4 * 4) = acc * inv;
    }
}

// ---- y = act(x * scale[c] + bias[c]) over NHWC pixels: the pre-activation BatchNorm + ReLU of the FAN BasicBlock
// (model_48px_ctc.py:392-394), whose input also feeds the residual and therefore cannot be folded into a conv ----
__global__ void affine_act_kernel(const float *__restrict__ in, int64_t in_pix, const float *__restrict__ scale,
                                  const float *__restrict__ bias, float *__restrict__ out, int64_t out_pix, int64_t npix, int C4,
                                  int relu) {
    const int64_t total = npix * C4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        const int64_t p = i / C4;
        const f32x4_t v = *reinterpret_cast<const f32x4_t *>(in + p * in_pix + c4 * 4);
        const f32x4_t s = *reinterpret_cast<const f32x4_t *>(scale + c4 * 4);
        const f32x4_t b = *reinterpret_cast<const f32x4_t *>(bias + c4 * 4);
        f32x4_t o;
        o.x = v.x * s.x + b.x; o.y = v.y * s.y + b.y; o.z = v.z * s.z + b.z; o.w = v.w * s.w + b.w;
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<f32x4_t *>(out + p * out_pix + c4 * 4) = o;
    }
}

// ---- u8 RGB pixels -> fp32 NHWC4 (4th channel 0).  mode 0: (x - 127.5) / 127.5 (model_48px.py:115); mode 1: x / 127.5 - 1
// (det_batch_forward_default, detection/default.py:19); mode 2: x / 255.  The three forms differ in their fp32 rounding. ----
__global__ void u8_to_f32_kernel(const uint8_t *__restrict__ in, float4 *__restrict__ out, int64_t npix, int mode) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < npix; i += stride) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = (float)in[3 * i + c];
            v[c] = mode == 0 ? (x - 127.5f) / 127.5f : (mode == 1 ? x / 127.5f - 1.0f : x / 255.0f);
        }
        out[i] = float4{v[0], v[1], v[2], 0.f};
    }
}

// ---- x <- sigmoid(x) ----
__global__ void sigmoid_kernel(float *__restrict__ x, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = 1.f / (1.f + expf(-x[i]));
}

// ---- x <- gelu(x), erf form ----
__global__ void gelu_kernel(float *__restrict__ x, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = x[i];
        x[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}

// ---- LayerNorm over the last dim (D <= 64*8), one wave per row ----
// PLANAR: the output goes out as bf16 planes (cells of 8 values): a row passes through 512 floats of LDS per wave.  Only that
// instantiation declares the LDS buffer — the row-major form (encoder, mit_layernorm) keeps its occupancy.
template <bool PLANAR>
__global__ void layernorm_kernel(const float *__restrict__ in, int64_t in_rs, const float *__restrict__ w,
                                 const float *__restrict__ b, float *__restrict__ out, int64_t out_rs, int rows, int D,
                                 float eps, OcrPlanes pl) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *x = in + (int64_t)row * in_rs;
    float v[8];
    float sum = 0.f;
    int n = 0;
    for (int d = lane; d < D; d += 64) {
        v[n] = x[d];
        sum += v[n];
        ++n;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)D;
    float var = 0.f;
    for (int j = 0; j < n; ++j) {
        const float t = v[j] - mean;
        var += t * t;
    }
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)D + eps);
    n = 0;
    if constexpr (PLANAR) {
        __shared__ __attribute__((aligned(16))) float ybuf[4][512];
        float *yb = ybuf[threadIdx.x >> 6];
        for (int d = lane; d < D; d += 64) {
            yb[d] = (v[n] - mean) * rstd * w[d] + b[d];
            ++n;
        }
        wave_lds_fence();
        for (int c = lane; c < (D >> 3); c += 64) store_cells(pl, c, row, yb + c * 8);
        return;
    }
    float *y = out + (int64_t)row * out_rs;
    for (int d = lane; d < D; d += 64) {
        y[d] = (v[n] - mean) * rstd * w[d] + b[d];
        ++n;
    }
}

// ---- XPOS rotation (xpos_relative_position.py:36-39,54-71) on [R, T, heads*80] ----
// position index i = i0 + t selects the sin/cos row, p = p0 + t the scale row (centred positions).
// dstep (optional, the decoder's device-resident step counter — see ocr_decoder.hip): mode 1 = "the query of this step": the input row
// starts step * dyn_in floats further, i0 = step, p0 = step + minpos; mode 2 = "the key history": rows t <= step only, p0 = minpos;
// minpos = -((step + 2) / 2), the centred origin of the positions 0 .. step (python -(step + 1) // 2).
__global__ void xpos_rotate_kernel(const float *__restrict__ in, int64_t in_rs, int64_t in_ts, float *__restrict__ out,
                                   int64_t out_rs, int64_t out_ts, int R, int T, int i0, int p0, int downscale,
                                   const float *__restrict__ cosT, const float *__restrict__ sinT,
                                   const float *__restrict__ scaleT, const float *__restrict__ iscaleT, int pmax,
                                   const int *__restrict__ dstep, int dyn_mode, int64_t dyn_in) {
    int t_end = T;
    if (dstep) {
        const int step = *dstep;
        const int minpos = -((step + 2) / 2);
        if (dyn_mode == 1) {
            in += (int64_t)step * dyn_in;
            i0 = step;
            p0 = step + minpos;
        } else {
            t_end = step + 1;
            p0 = minpos;
        }
    }
    // one thread per (r, t, pair j of 160 pairs = 4 heads x 40)
    const int64_t total = (int64_t)R * T * 160;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int pair = (int)(i % 160);
        const int64_t rt = i / 160;
        const int t = (int)(rt % T);
        const int r = (int)(rt / T);
        if (t >= t_end) continue;
        const int j = pair % 40;
        const int ii = i0 + t, pp = p0 + t + pmax;
        const float sc = downscale ? iscaleT[pp * 40 + j] : scaleT[pp * 40 + j];
        const float c = cosT[ii * 40 + j] * sc;
        const float s = sinT[ii * 40 + j] * sc;
        const float2 x = *reinterpret_cast<const float2 *>(in + (int64_t)r * in_rs + (int64_t)t * in_ts + pair * 2);
        float2 o;
        o.x = x.x * c + (-x.y) * s;  // (x * cos) + (rotate_every_two(x) * sin)
        o.y = x.y * c + x.x * s;
        *reinterpret_cast<float2 *>(out + (int64_t)r * out_rs + (int64_t)t * out_ts + pair * 2) = o;
    }
}

// ---- softmax(q k^T + key mask) v, gridDim.y heads x HD (4 x 80 for the 48px model, 8 x 40 for 48px_ctc), one wave per
// (query, head, row) ----
__global__ void attention_kernel(const float *__restrict__ Q, int64_t q_rs, int64_t q_ts, const float *__restrict__ K,
                                 int64_t k_rs, int64_t k_ts, const float *__restrict__ V, int64_t v_rs, int64_t v_ts,
                                 float *__restrict__ O, int64_t o_rs, int64_t o_ts, const int *__restrict__ klen, int Tk,
                                 int kv_div, int HD, const int *__restrict__ dstep, OcrAttXpos xp, OcrPlanes opl) {
    // dstep without folded rotation / with rotated keys: the decoder's self-attention over the tokens 0 .. step (LDS is sized for the longest history)
    if (dstep && (!xp.cos_t || xp.rot_k)) Tk = *dstep + 1;
    const int step = dstep ? *dstep : xp.step;
    const int minpos = -((step + 2) / 2);
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [HD] q + [Tk] weights
    float *qs = lds;
    float *ws = lds + HD;
    const int tq = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int lane = threadIdx.x;
    const int kr = r / kv_div;
    const float *q = Q + (int64_t)r * q_rs + (int64_t)tq * q_ts + h * HD;
    const int HP = HD / 2;
    if (xp.cos_t) {  // the query of position `step`, rotated on the way into LDS (xpos_rotate_kernel's expression, scale up)
        if (dstep) q += (int64_t)step * xp.q_dyn;
        const int pp = step + minpos + xp.pmax;
        for (int j = lane; j < HP; j += 64) {
            const float sc = xp.scale_t[pp * HP + j];
            const float c = xp.cos_t[step * HP + j] * sc, sn = xp.sin_t[step * HP + j] * sc;
            const float2 x = *reinterpret_cast<const float2 *>(q + 2 * j);
            qs[2 * j] = x.x * c + (-x.y) * sn;
            qs[2 * j + 1] = x.y * c + x.x * sn;
        }
    } else {
        for (int d = lane; d < HD; d += 64) qs[d] = q[d];
    }
    __syncthreads();
    const int valid = klen ? min(klen[kr], Tk) : Tk;
    const float *kb = K + (int64_t)kr * k_rs + h * HD;
    float mx = -INFINITY;
    for (int t = lane; t < Tk; t += 64) {
        float dot = -INFINITY;
        if (t < valid) {
            const float4 *kp = reinterpret_cast<const float4 *>(kb + (int64_t)t * k_ts);
            dot = 0.f;
            if (xp.cos_t && xp.rot_k) {  // key t of the raw history, rotated (scale down) as it is read
                const float4 *cp = reinterpret_cast<const float4 *>(xp.cos_t + t * HP);
                const float4 *sp = reinterpret_cast<const float4 *>(xp.sin_t + t * HP);
                const float4 *ip = reinterpret_cast<const float4 *>(xp.iscale_t + (minpos + t + xp.pmax) * HP);
                for (int d8 = 0; d8 < HD / 8; ++d8) {
                    const float4 k0 = kp[2 * d8], k1 = kp[2 * d8 + 1], cc = cp[d8], ss = sp[d8], ii = ip[d8];
                    float c, sn;
                    c = cc.x * ii.x, sn = ss.x * ii.x;
                    dot += qs[d8 * 8 + 0] * (k0.x * c + (-k0.y) * sn);
                    dot += qs[d8 * 8 + 1] * (k0.y * c + k0.x * sn);
                    c = cc.y * ii.y, sn = ss.y * ii.y;
                    dot += qs[d8 * 8 + 2] * (k0.z * c + (-k0.w) * sn);
                    dot += qs[d8 * 8 + 3] * (k0.w * c + k0.z * sn);
                    c = cc.z * ii.z, sn = ss.z * ii.z;
                    dot += qs[d8 * 8 + 4] * (k1.x * c + (-k1.y) * sn);
                    dot += qs[d8 * 8 + 5] * (k1.y * c + k1.x * sn);
                    c = cc.w * ii.w, sn = ss.w * ii.w;
                    dot += qs[d8 * 8 + 6] * (k1.z * c + (-k1.w) * sn);
                    dot += qs[d8 * 8 + 7] * (k1.w * c + k1.z * sn);
                }
            } else {
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    const float4 kv = kp[d4];
                    dot += qs[d4 * 4 + 0] * kv.x;
                    dot += qs[d4 * 4 + 1] * kv.y;
                    dot += qs[d4 * 4 + 2] * kv.z;
                    dot += qs[d4 * 4 + 3] * kv.w;
                }
            }
        }
        ws[t] = dot;
        mx = fmaxf(mx, dot);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int t = lane; t < Tk; t += 64) {
        const float e = expf(ws[t] - mx);
        ws[t] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int t = lane; t < Tk; t += 64) ws[t] *= inv;
    __syncthreads();
    const float *vb = V + (int64_t)kr * v_rs + h * HD;
    float *ob = O + (int64_t)r * o_rs + (int64_t)tq * o_ts + h * HD;
    constexpr int U = 8;  // values of V in flight per lane; the sum itself stays t-ordered
    for (int d = lane; d < HD; d += 64) {
        float acc = 0.f;
        for (int t0 = 0; t0 < valid; t0 += U) {
            float vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) vv[u] = (t0 + u < valid) ? vb[(int64_t)(t0 + u) * v_ts + d] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (t0 + u < valid) acc += ws[t0 + u] * vv[u];
        }
        if (opl.p) qs[d] = acc;   // planar output (Tq == 1): the head's HD values become HD / 8 cells of row r (qs is free by now)
        else ob[d] = acc;
    }
    if (opl.p) {
        wave_lds_fence();
        for (int c = lane; c < (HD >> 3); c += 64) store_cells(opl, h * (HD >> 3) + c, r, qs + c * 8);
    }
}

// ---- the decoder's SELF-attention (one query = the step's token, keys = the row's own history 0 .. step, rotated as they are read):
// one workgroup per row, one wave per head.  attention_kernel gives every (head, row) a workgroup of one wave whose lane t walks key
// t's 320-byte slice of a 1280-byte row by itself — 64 scattered streams per wave, 40 960 one-wave workgroups per launch, 27 % of the
// HBM rate.  Here the row's whole history [Tk][heads * HD] (contiguous: <= 40 KB) is brought in by all 256 threads as coalesced float4
// runs, parked in LDS with a pitch of heads * HD + 4 floats (lane t's float4 reads then fall on distinct banks), and every head's wave
// evaluates exactly attention_kernel's expressions on it — per-key dot products in d order with the rotation folded in, the
// lane-strided softmax, the t-ordered weighted sum — so the results are bitwise the same.
// Round 6 (scripts/dev/att_stamps.py: 12.3 us inside the kernel at Tk = 32, of which 4.6 staging the keys, 2.2 scores, 4.7 weighted sum):
// every global load of a phase is now in flight at once — the keys and, behind them, the values travel to REGISTERS first (up to
// SELF_ST float4 per thread; the values are parked in the key area once the scores are done), and the rotation factors cos * iscale,
// sin * iscale of (key t, pair j) — the same for all four heads, read by every lane from three tables inside its dot-product loop
// before — are formed once per workgroup into LDS.  Same expressions, same order per (query, key) and per (query, d): same bits.
constexpr int SELF_ST = 11;    // float4 per thread of a staged [Tk][E] block: Tk * E / 4 <= 256 * SELF_ST (Tk <= 35 at E = 320); past it: a plain loop
constexpr int SELF_TP_PAD = 4; // table row pitch HP + 4 floats: lane t's float4 reads fall on distinct 16-byte slots (HP = 40)
template <int heads, int HD>   // compile-time: the staging loops divide by E / 4 and HD / 2 some forty times per thread
// (argument order: what the first batch of loads needs comes first — twelve dwords are preloaded into SGPRs with the wave, build.py;
// the token stride of K and V is E, checked by the launcher)
__global__ __launch_bounds__(256) void attention_self_kernel(const float *__restrict__ K, int64_t k_rs, const float *__restrict__ V, int64_t v_rs,
                                                             const float *__restrict__ Q, int64_t q_rs, int TkCap, const int *__restrict__ dstep,
                                                             float *__restrict__ O, int64_t o_rs, OcrAttXpos xp, OcrPlanes opl) {
    constexpr int64_t k_ts = (int64_t)heads * HD, v_ts = k_ts;
    static_assert(heads * 64 == 256 && HD % 8 == 0, "one wave per head");
    MIT_ATT_STAMP2(4);
    const int Tk = dstep ? *dstep + 1 : TkCap;
    const int step = dstep ? *dstep : xp.step;
    const int minpos = -((step + 2) / 2);
    constexpr int E = heads * HD, KP = E + 4, HP = HD / 2, TP = HP + SELF_TP_PAD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *qs_all = lds;                      // [heads][HD]
    float *ws_all = qs_all + E;               // [heads][TkCap]
    float *ks = ws_all + heads * TkCap;       // [TkCap][KP]   keys, then values   (E + heads * TkCap is a multiple of 4: 16-byte aligned rows)
    float *tc = ks + TkCap * KP;              // [TkCap][TP]   cos * iscale of (key t, pair j)
    float *tsn = tc + TkCap * TP;             // [TkCap][TP]   sin * iscale
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, r = blockIdx.x;
    float *qs = qs_all + h * HD, *ws = ws_all + h * TkCap;
    constexpr int E4 = E >> 2;
    const int n4 = Tk * E4;
    // ---- every global load of the kernel is requested here, in one batch: the key history, the value history (it waits in registers
    // until the scores are done), the query and the table entries of the rotations.  The pins below keep them here: the loads are from
    // read-only memory, so left alone the optimiser sinks each one to its use — the values' became eleven round trips in a row inside
    // the store loop behind the scores (3.3 us of a 12 us workgroup, scripts/dev/att_stamps.py).
    f32x4 stk[SELF_ST], stv[SELF_ST];
    const float *kb = K + (int64_t)r * k_rs, *vbase = V + (int64_t)r * v_rs;
    auto stage_load = [&](f32x4 (&st)[SELF_ST], const float *base, const int64_t ts) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SELF_ST; ++j) {
            const int i = tid + j * 256, ii = i < n4 ? i : 0, t = ii / E4, c4 = ii - t * E4;
            st[j] = *reinterpret_cast<const f32x4 *>(base + (int64_t)t * ts + c4 * 4);
        }
    };
    auto stage_store = [&](f32x4 (&st)[SELF_ST], const float *base, const int64_t ts) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SELF_ST; ++j) {
            const int i = tid + j * 256;
            if (i < n4) {
                const int t = i / E4, c4 = i - t * E4;
                *reinterpret_cast<f32x4 *>(ks + t * KP + c4 * 4) = st[j];
            }
        }
        for (int i = tid + SELF_ST * 256; i < n4; i += 256) {   // (histories longer than the register stage holds)
            const int t = i / E4, c4 = i - t * E4;
            *reinterpret_cast<f32x4 *>(ks + t * KP + c4 * 4) = *reinterpret_cast<const f32x4 *>(base + (int64_t)t * ts + c4 * 4);
        }
    };
    stage_load(stk, kb, k_ts);
    stage_load(stv, vbase, v_ts);
    // the query of position `step` (lane j < HP: pair j of this wave's head) and its rotation factors
    static_assert(HP <= 64, "one pair per lane");
    const int jq = lane < HP ? lane : 0;
    const int pp = step + minpos + xp.pmax;
    float q_sc = xp.scale_t[pp * HP + jq], q_c = xp.cos_t[step * HP + jq], q_s = xp.sin_t[step * HP + jq];
    float2 q_x = *reinterpret_cast<const float2 *>(Q + (int64_t)r * q_rs + h * HD + (dstep ? (int64_t)step * xp.q_dyn : 0) + 2 * jq);
    // rotation factors of the raw key history: entries i = t * HP + j of the tables
    constexpr int TT = 6;   // entries per thread held in registers (Tk * HP <= 256 * TT: Tk <= 38 at HP = 40); past it: a plain loop
    float fc[TT], fs[TT], fi[TT];
    const int nt = Tk * HP;
#pragma unroll
    for (int u = 0; u < TT; ++u) {
        const int i = tid + u * 256, ii = i < nt ? i : 0, t = ii / HP, j = ii - t * HP;
        fi[u] = xp.iscale_t[(minpos + t + xp.pmax) * HP + j];
        fc[u] = xp.cos_t[ii];      // (t * HP + j == ii)
        fs[u] = xp.sin_t[ii];
    }
#pragma unroll
    for (int j = 0; j < SELF_ST; ++j) asm volatile("" : "+v"(stk[j]), "+v"(stv[j]));
#pragma unroll
    for (int u = 0; u < TT; ++u) asm volatile("" : "+v"(fc[u]), "+v"(fs[u]), "+v"(fi[u]));
    asm volatile("" : "+v"(q_sc), "+v"(q_c), "+v"(q_s), "+v"(q_x.x), "+v"(q_x.y));
    if (lane < HP) {   // the query, rotated on the way into LDS (attention_kernel's expression, scale up)
        const float c = q_c * q_sc, sn = q_s * q_sc;
        qs[2 * lane] = q_x.x * c + (-q_x.y) * sn;
        qs[2 * lane + 1] = q_x.y * c + q_x.x * sn;
    }
#pragma unroll
    for (int u = 0; u < TT; ++u) {   // c = cos * iscale, sn = sin * iscale (scale down), once per workgroup for its four heads
        const int i = tid + u * 256;
        if (i < nt) {
            const int t = i / HP, j = i - t * HP;
            tc[t * TP + j] = fc[u] * fi[u];
            tsn[t * TP + j] = fs[u] * fi[u];
        }
    }
    for (int i = tid + TT * 256; i < nt; i += 256) {
        const int t = i / HP, j = i - t * HP;
        const float is = xp.iscale_t[(minpos + t + xp.pmax) * HP + j];
        tc[t * TP + j] = xp.cos_t[t * HP + j] * is;
        tsn[t * TP + j] = xp.sin_t[t * HP + j] * is;
    }
    stage_store(stk, kb, k_ts);
    __syncthreads();
    MIT_ATT_STAMP2(5);
    float mx = -INFINITY;
    for (int t = lane; t < Tk; t += 64) {
        const float4 *kp = reinterpret_cast<const float4 *>(ks + t * KP + h * HD);
        const float4 *cp = reinterpret_cast<const float4 *>(tc + t * TP);
        const float4 *sp = reinterpret_cast<const float4 *>(tsn + t * TP);
        float dot = 0.f;
        for (int d8 = 0; d8 < HD / 8; ++d8) {  // key t of the raw history, rotated (scale down) as it is read
            const float4 k0 = kp[2 * d8], k1 = kp[2 * d8 + 1], cc = cp[d8], ss = sp[d8];
            dot += qs[d8 * 8 + 0] * (k0.x * cc.x + (-k0.y) * ss.x);
            dot += qs[d8 * 8 + 1] * (k0.y * cc.x + k0.x * ss.x);
            dot += qs[d8 * 8 + 2] * (k0.z * cc.y + (-k0.w) * ss.y);
            dot += qs[d8 * 8 + 3] * (k0.w * cc.y + k0.z * ss.y);
            dot += qs[d8 * 8 + 4] * (k1.x * cc.z + (-k1.y) * ss.z);
            dot += qs[d8 * 8 + 5] * (k1.y * cc.z + k1.x * ss.z);
            dot += qs[d8 * 8 + 6] * (k1.z * cc.w + (-k1.w) * ss.w);
            dot += qs[d8 * 8 + 7] * (k1.w * cc.w + k1.z * ss.w);
        }
        ws[t] = dot;
        mx = fmaxf(mx, dot);
    }
    MIT_ATT_STAMP2(6);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int t = lane; t < Tk; t += 64) {
        const float e = expf(ws[t] - mx);
        ws[t] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int t = lane; t < Tk; t += 64) ws[t] *= inv;
    MIT_ATT_STAMP2(10);
    __syncthreads();           // every head is done with the keys (and the weights are visible)
    MIT_ATT_STAMP2(11);
    stage_store(stv, vbase, v_ts);  // the values take their place
    MIT_ATT_STAMP2(12);
    __syncthreads();
    MIT_ATT_STAMP2(7);
    float *ob = O ? O + (int64_t)r * o_rs + h * HD : nullptr;
    const float *vs = ks + h * HD;
    for (int d = lane; d < HD; d += 64) {
        float acc = 0.f;
        for (int t = 0; t < Tk; ++t) acc += ws[t] * vs[t * KP + d];   // t-ordered
        if (opl.p) qs[d] = acc;
        else ob[d] = acc;
    }
    MIT_ATT_STAMP2(8);
    if (opl.p) {
        wave_lds_fence();
        for (int c = lane; c < (HD >> 3); c += 64) store_cells(opl, h * (HD >> 3) + c, r, qs + c * 8);
    }
    MIT_ATT_STAMP2(9);
}

// ---- the same attention for ONE query position of G = kv_div consecutive rows that share a K / V block (the beams of a line
// in cross-attention): one workgroup per (head, K/V block) stages the keys through LDS once instead of once per beam.
// Per (query, key) dot products, the lane-strided softmax sums and the t-ordered weighted sum are evaluated in exactly the
// order of attention_kernel, so both give bitwise identical results.
constexpr int ATT_G_MAX = 8;
constexpr int ATT_THREADS = 256;
// keys per LDS chunk / float4 loads per thread per chunk (head_dim 80): the throughput form (many workgroups per CU) and the latency form
// for launches that leave CUs idle (one page: 4 heads x 32 lines) — a line's keys in ONE chunk, so one load round trip per pass
constexpr int ATT_KCHUNK = 64, ATT_STAGE = 8;
constexpr int ATT_KCHUNK_L = 160, ATT_STAGE_L = 13;

// chunk [nk keys x HD] of one head, global -> registers (coalesced: consecutive threads walk a key row)
template <int STAGE>
__device__ __forceinline__ void att_chunk_load(float4 (&reg)[STAGE], const float *base, int64_t ts, int t0, int nk, int HD4, int tid) {
#pragma unroll
    for (int j = 0; j < STAGE; ++j) {
        const int i = tid + j * ATT_THREADS;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < nk * HD4) {
            const int t = i / HD4, d4 = i - t * HD4;
            v = *reinterpret_cast<const float4 *>(base + (int64_t)(t0 + t) * ts + d4 * 4);
        }
        reg[j] = v;
    }
}

template <int STAGE>
__device__ __forceinline__ void att_chunk_store(const float4 (&reg)[STAGE], float *ks, int nk, int HD4, int KP, int tid) {
#pragma unroll
    for (int j = 0; j < STAGE; ++j) {
        const int i = tid + j * ATT_THREADS;
        if (i < nk * HD4) {
            const int t = i / HD4, d4 = i - t * HD4;
            *reinterpret_cast<float4 *>(ks + t * KP + d4 * 4) = reg[j];
        }
    }
}

// QF (the decoder's cross-attention at few rows): the queries are not read from Q but COMPUTED here — q = LayerNorm(x) @ Wq + bias for
// the line's G beams and this head's HD columns (TransformerDecoderLayer norm2 + multihead_attn's q projection, model_48px.py:548-572):
// wave 3 normalises the G rows (ln_rows8.h: layernorm_kernel's bits) and parks their planes in LDS, waves 0 .. 2 run pgemm_rows_kernel's
// K loop on 32 columns each (same cells, same pair and k order, same bias add) while the line's keys travel, and the results go
// through the rotation into qs as the loaded queries would.  One launch less per layer and step, bit for bit the two-launch form.
constexpr int ATT_QF_K8 = mitln::LN_K / 8, ATT_QF_ROWS = 8, ATT_QF_DEPTH = 6;
constexpr size_t ATT_QF_LDS = (size_t)3 * ATT_QF_K8 * ATT_QF_ROWS * 16;

template <int KCHUNK, int STAGE, int HD, int G, bool QF = false>
__global__ __launch_bounds__(ATT_THREADS) void attention_shared_kv_kernel(const float *__restrict__ Q, int64_t q_rs,
                                                                          const float *__restrict__ K, int64_t k_rs, int64_t k_ts,
                                                                          const float *__restrict__ V, int64_t v_rs, int64_t v_ts,
                                                                          float *__restrict__ O, int64_t o_rs,
                                                                          const int *__restrict__ klen, int Tk,
                                                                          const int *__restrict__ dstep, OcrAttXpos xp, OcrPlanes opl, OcrAttQProj qp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KP = HD + 4;                // 16-byte aligned rows; lane t reads row t as float4s (pitch 84: conflict-free per 16 lanes)
    constexpr int HD4 = HD / 4;
    float *qs = lds;                          // [G][HD]
    float *ks = qs + G * HD;                  // [KCHUNK][KP]   keys, then values
    float *ws = ks + KCHUNK * KP;             // [G][Tk]
    // QF: [3][K8][ATT_QF_ROWS] cells of the normalised rows, in the key area — free until the first chunk (in registers so far) is parked
    // there, behind the barrier that also publishes qs
    mitcg::u32x4 *apl = reinterpret_cast<mitcg::u32x4 *>(ks);
    static_assert(!QF || (ATT_QF_LDS <= (size_t)KCHUNK * KP * 4 && (G * HD) % 4 == 0), "the planes fit the key area, 16-byte aligned");
    const int h = blockIdx.x, kr = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = kr * G;
    float4 stage[STAGE];
    const float *kb = K + (int64_t)kr * k_rs + h * HD;
    const float *vb = V + (int64_t)kr * v_rs + h * HD;
    // the first keys travel while the queries are prepared; the latency form does not even wait for the line's length (rows past it are
    // padding inside the line's block: loaded, stored, never used)
    constexpr bool EAGER = KCHUNK > 64;
    MIT_ATT_STAMP(0);
    if (EAGER) att_chunk_load<STAGE>(stage, kb, k_ts, 0, min(KCHUNK, Tk), HD4, tid);
    const int valid = klen ? min(klen[kr], Tk) : Tk;
    if (!EAGER) att_chunk_load<STAGE>(stage, kb, k_ts, 0, min(KCHUNK, valid), HD4, tid);
    if constexpr (QF) {
        using namespace mitcg;
        static_assert(G <= ATT_QF_ROWS && HD % 8 == 0 && HD <= 96, "three 32-column blocks per head, one 8-row slab");
        constexpr int K8 = ATT_QF_K8, KTS = mitln::LN_K / 16, D = ATT_QF_DEPTH;
        const int li = lane & 31, lh = lane >> 5;
        const unsigned int w_step = (unsigned int)qp.ldw * 32u, w_plane = (unsigned int)K8 * (unsigned int)qp.ldw * 16u;
        const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(qp.w_planes), 0, 3 * w_plane, 0x00020000);
        // (columns past the head's HD, or past the matrix: cells of other columns / zeros from the descriptor — computed, never used)
        const unsigned int w_off = ((unsigned int)lh * (unsigned int)qp.ldw + (unsigned int)(h * HD + 32 * wave + li)) * 16u;
        u32x4 fw[D][3];
        auto issue = [&](const int d, const int kstep) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fw[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off, pl * w_plane + (unsigned int)kstep * w_step, 0);
        };
        const int step = dstep ? *dstep : xp.step;
        constexpr int HP = HD / 2;
        const int pp = step + -((step + 2) / 2) + xp.pmax;
        // what the epilogue needs from global memory — the Linear's scale / bias and the rotation's table entries of this lane's columns —
        // is requested here, ahead of the K loop (behind it, it was a dependent round trip of its own)
        float e_sc[2][8], e_bi[2][8], e_c[2][4], e_s[2][4];
        if (wave < 3) {
#pragma unroll
            for (int d = 0; d < D; ++d) issue(d, d);
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {
                const int c0 = 32 * wave + 8 * (2 * pq + lh), cc = c0 < HD ? c0 : 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    e_sc[pq][e] = qp.scale ? qp.scale[h * HD + cc + e] : 1.f;
                    e_bi[pq][e] = qp.bias[h * HD + cc + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = cc / 2 + e;
                    const float sc = xp.cos_t ? xp.scale_t[pp * HP + j] : 1.f;
                    e_c[pq][e] = xp.cos_t ? xp.cos_t[step * HP + j] * sc : 1.f;
                    e_s[pq][e] = xp.cos_t ? xp.sin_t[step * HP + j] * sc : 0.f;
                }
            }
        }
        if (wave == 3) {   // rows r0 .. r0 + G - 1, eight lanes each (the groups past G repeat row G - 1 and store nothing)
            const int rr = lane >> 3, q = lane & 7, row = r0 + (rr < G ? rr : G - 1);
            f32x4 v[5][2];
            mitln::ln_row_cells(qp.x + (int64_t)row * qp.ldx + 8 * q, qp.ln_w + 8 * q, qp.ln_b + 8 * q, qp.eps, v);
            if (rr < G) {
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    u32x4 ph, pm, pl_;
                    split8(v[b][0], v[b][1], ph, pm, pl_);
                    apl[(0 * K8 + 8 * b + q) * ATT_QF_ROWS + rr] = ph;
                    apl[(1 * K8 + 8 * b + q) * ATT_QF_ROWS + rr] = pm;
                    apl[(2 * K8 + 8 * b + q) * ATT_QF_ROWS + rr] = pl_;
                }
            }
        }
        __syncthreads();
        MIT_ATT_STAMP2(0);
        if (wave < 3) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            const u32x4 *ya = apl + lh * ATT_QF_ROWS + (li < G ? li : 0);
            u32x4 fa[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[0][pl] = li < G ? ya[(pl * K8) * ATT_QF_ROWS] : zero;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kstep = 0; kstep < KTS; ++kstep) {
                const int d = kstep % D, cur = kstep & 1;
                if (kstep + 1 < KTS) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fa[cur ^ 1][pl] = li < G ? ya[(pl * K8 + 2 * (kstep + 1)) * ATT_QF_ROWS] : zero;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pr = 3; pr < 9; ++pr)   // the six plane pairs, transposed result (rows = output columns), as pgemm_rows_kernel
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[d][kSplitPB[pr]]), __builtin_bit_cast(bf16x8, fa[cur][kSplitPA[pr]]), acc, 0, 0, 0);
                if (kstep + D < KTS) issue(d, kstep + D);
                __builtin_amdgcn_sched_barrier(0);
            }
            MIT_ATT_STAMP2(1);
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {   // after the swap: beam li, the head's columns c0 .. c0 + 7
                float val[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * pq + e]), __float_as_uint(acc[8 * pq + 4 + e]), false, false);
                    val[e] = __uint_as_float(sw[0]);
                    val[4 + e] = __uint_as_float(sw[1]);
                }
                const int c0 = 32 * wave + 8 * (2 * pq + lh);
                if (li < G && c0 < HD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = val[e] * e_sc[pq][e] + e_bi[pq][e];   // the Linear's epilogue: acc * scale + bias
                    if (xp.cos_t) {   // (xpos_rotate_kernel's expression with c = cos * scale, sn = sin * scale)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = c0 / 2 + e;
                            qs[li * HD + 2 * j] = val[2 * e] * e_c[pq][e] + (-val[2 * e + 1]) * e_s[pq][e];
                            qs[li * HD + 2 * j + 1] = val[2 * e + 1] * e_c[pq][e] + val[2 * e] * e_s[pq][e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) qs[li * HD + c0 + e] = val[e];
                    }
                }
            }
        }
    } else if (xp.cos_t) {  // the beams' queries of position `step`, rotated on the way into LDS (xpos_rotate_kernel's expression, scale up)
        const int step = dstep ? *dstep : xp.step;
        constexpr int HP = HD / 2;
        const int pp = step + -((step + 2) / 2) + xp.pmax;
        const int64_t qo = dstep ? (int64_t)step * xp.q_dyn : 0;
        for (int i = tid; i < G * HP; i += ATT_THREADS) {
            const int g = i / HP, j = i - g * HP;
            const float sc = xp.scale_t[pp * HP + j];
            const float c = xp.cos_t[step * HP + j] * sc, sn = xp.sin_t[step * HP + j] * sc;
            const float2 x = *reinterpret_cast<const float2 *>(Q + qo + (int64_t)(r0 + g) * q_rs + h * HD + 2 * j);
            qs[g * HD + 2 * j] = x.x * c + (-x.y) * sn;
            qs[g * HD + 2 * j + 1] = x.y * c + x.x * sn;
        }
    } else {
        for (int i = tid; i < G * HD; i += ATT_THREADS) qs[i] = Q[(int64_t)(r0 + i / HD) * q_rs + h * HD + (i % HD)];
    }

    MIT_ATT_STAMP(1);
    // ---- pass 1: scores.  The next chunk's keys travel to registers while this chunk's dot products run; behind the last chunk of keys
    // the first chunk of VALUES does (its latency is hidden by the last dot products and the softmax).  A thread owns one key of the
    // chunk — its row read from LDS once, into registers — and every KT-th ... query of the line.
    constexpr int KT = KCHUNK > 128 ? 256 : (KCHUNK > 64 ? 128 : 64), NGRP = ATT_THREADS / KT;
    // (NGRP == 1: compile-time zero — with a run-time tid / KT the compiler cannot see that every query index is below G and wraps each
    // query's sums in an exec-masked branch of its own: five dependent chains one after the other instead of interleaved)
    const int tl = NGRP == 1 ? tid : tid % KT, gsub = NGRP == 1 ? 0 : tid / KT;
    bool v_ahead = false;
    for (int t0 = 0; t0 < Tk; t0 += KCHUNK) {
        const int nk = max(0, min(KCHUNK, valid - t0));
        __syncthreads();  // previous chunk consumed (and qs visible)
        if (!v_ahead) att_chunk_store<STAGE>(stage, ks, (EAGER && t0 == 0) ? min(KCHUNK, Tk) : nk, HD4, KP, tid);
        __syncthreads();
        MIT_ATT_STAMP(2);
        if (t0 + KCHUNK < valid) {
            att_chunk_load<STAGE>(stage, kb, k_ts, t0 + KCHUNK, min(KCHUNK, valid - t0 - KCHUNK), HD4, tid);
        } else if (!v_ahead) {
            att_chunk_load<STAGE>(stage, vb, v_ts, 0, min(KCHUNK, valid), HD4, tid);
            v_ahead = true;
        }
        const int t = t0 + tl;
        if (tl < KCHUNK && t < Tk) {
            if (t < valid) {
                float4 kv[HD4];
#pragma unroll
                for (int d4 = 0; d4 < HD4; ++d4) kv[d4] = *reinterpret_cast<const float4 *>(ks + tl * KP + d4 * 4);
                // the thread's queries side by side: their (d-ordered, dependent) sums are independent of each other and interleave
                constexpr int NQ = (G + NGRP - 1) / NGRP;
                float dot[NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) dot[qi] = 0.f;
                // The queries' values (one address for every lane: a broadcast read) run ONE d-group ahead of their products, fenced: left
                // to itself the compiler issues each read right before its use and waits for it — a hundred LDS round trips in a row, 4.6 us
                // of a 12 us workgroup (scripts/dev/att_stamps.py) — although the five sums of a group are 40 independent VALU
                // instructions that cover the next group's latency.
                float4 qv[2][NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) qv[0][qi] = *reinterpret_cast<const float4 *>(qs + min(gsub + qi * NGRP, G - 1) * HD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d4 = 0; d4 < HD4; ++d4) {  // d ascending, one rounding per product and per sum, as in attention_kernel
                    if (d4 + 1 < HD4) {
#pragma unroll
                        for (int qi = 0; qi < NQ; ++qi)
                            qv[(d4 + 1) & 1][qi] = *reinterpret_cast<const float4 *>(qs + min(gsub + qi * NGRP, G - 1) * HD + (d4 + 1) * 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        const int g = gsub + qi * NGRP;
                        if (g < G) {
                            const float4 q4 = qv[d4 & 1][qi];
                            dot[qi] += q4.x * kv[d4].x;
                            dot[qi] += q4.y * kv[d4].y;
                            dot[qi] += q4.z * kv[d4].z;
                            dot[qi] += q4.w * kv[d4].w;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const int g = gsub + qi * NGRP;
                    if (g < G) ws[(int64_t)g * Tk + t] = dot[qi];
                }
            } else {
                for (int g = gsub; g < G; g += NGRP) ws[(int64_t)g * Tk + t] = -INFINITY;
            }
        }
    }
    __syncthreads();
    MIT_ATT_STAMP(3);
    {   // softmax: one wave per query, lanes strided over the keys; a wave with two queries (G = 5: wave 0) runs them side by side so
        // that their shuffle and exp latencies overlap (2.0 -> 1 us of the workgroup); per query the same operations in the same order
        constexpr int NW = ATT_THREADS / 64, QW = (G + NW - 1) / NW;
        float *wq[QW];
        bool on[QW];
        float mx[QW], sum[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            on[i] = wave + i * NW < G;
            wq[i] = ws + (int64_t)(on[i] ? wave + i * NW : min(wave, G - 1)) * Tk;   // (an idle slot re-reads a row of this wave's, stores nothing)
            mx[i] = -INFINITY;
            sum[i] = 0.f;
        }
        for (int t = lane; t < Tk; t += 64)
#pragma unroll
            for (int i = 0; i < QW; ++i) mx[i] = fmaxf(mx[i], wq[i][t]);
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < QW; ++i) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
        for (int t = lane; t < Tk; t += 64)
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const float e = expf(wq[i][t] - mx[i]);
                if (on[i]) wq[i][t] = e;
                sum[i] += e;
            }
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < QW; ++i) sum[i] += __shfl_xor(sum[i], o);
        float iv[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) iv[i] = 1.0f / sum[i];
        for (int t = lane; t < Tk; t += 64)
#pragma unroll
            for (int i = 0; i < QW; ++i)
                if (on[i]) wq[i][t] *= iv[i];  // the (weight * 1/sum) factor of the weighted sum, formed once
    }

    MIT_ATT_STAMP(4);
    // ---- pass 2: weighted sum of the values, t-ordered per (query, d); thread (d, half) owns the queries g = half, half + 2, ...
    // Waves 0 / 1: half = wave, d = lane; wave 2: d = 64 .. 79 of both halves (a wave's weight reads are then one address: a broadcast).
    // Branch-free inner loop: a (d, half) whose last query index falls past G sums a duplicate of query G - 1 that is never stored.
    static_assert(HD == 80, "thread mapping of the weighted sum");
    constexpr int J = (G + 1) / 2;
    const int half = wave < 2 ? wave : ((lane >> 4) & 1);
    const int d = wave < 2 ? lane : 64 + (lane & 15);
    const bool owner = wave < 2 || (wave == 2 && lane < 32);
    float acc[J];
    const float *wrow[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        acc[j] = 0.f;
        wrow[j] = ws + (int64_t)min(half + 2 * j, G - 1) * Tk;
    }
    for (int t0 = 0; t0 < valid; t0 += KCHUNK) {
        const int nk = min(KCHUNK, valid - t0);
        __syncthreads();  // previous chunk consumed (and the softmax weights visible)
        att_chunk_store<STAGE>(stage, ks, nk, HD4, KP, tid);
        __syncthreads();
        MIT_ATT_STAMP(5);
        if (t0 + KCHUNK < valid) att_chunk_load<STAGE>(stage, vb, v_ts, t0 + KCHUNK, min(KCHUNK, valid - t0 - KCHUNK), HD4, tid);
        if (owner) {
            constexpr int U = 8;  // values and weights of U keys read ahead of their (t-ordered) multiply-adds
            const float *vcol = ks + d;
            int tb = 0;
            for (; tb + U <= nk; tb += U) {
                float vv[U], wv[J][U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    vv[u] = vcol[(tb + u) * KP];
#pragma unroll
                    for (int j = 0; j < J; ++j) wv[j][u] = wrow[j][t0 + tb + u];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < J; ++j) acc[j] += wv[j][u] * vv[u];
            }
            for (; tb < nk; ++tb) {
                const float v = vcol[tb * KP];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] += wrow[j][t0 + tb] * v;
            }
        }
    }
    MIT_ATT_STAMP(6);
    if (opl.p) {  // planar output: the G x HD block passes through LDS (qs: last read in pass 1) to become cells of 8
        if (owner) {
#pragma unroll
            for (int j = 0; j < (G + 1) / 2; ++j) {
                const int g = half + 2 * j;
                if (g < G) qs[g * HD + d] = acc[j];
            }
        }
        __syncthreads();
        const int HC = HD >> 3;
        for (int i = tid; i < G * HC; i += ATT_THREADS) {
            const int g = i / HC, c = i - g * HC;
            store_cells(opl, h * HC + c, r0 + g, qs + g * HD + c * 8);
        }
        MIT_ATT_STAMP(7);
        return;
    }
    if (owner) {
#pragma unroll
        for (int j = 0; j < (G + 1) / 2; ++j) {
            const int g = half + 2 * j;
            if (g < G) O[(int64_t)(r0 + g) * o_rs + h * HD + d] = acc[j];
        }
    }
}

// ---- embedding rows: out[r] = E[tok[r]] ----
// dstep (optional): the token of step s is column s of the history buffer the previous step wrote — tok for s <= 1 and odd s, tok1 else
// (the two buffers alternate, see beam_dyn_kernel)
__global__ void embed_kernel(const int *__restrict__ tok, int64_t tok_stride, const float *__restrict__ E,
                             float *__restrict__ out, int R, int D, const int *__restrict__ tok1, const int *__restrict__ dstep) {
    if (dstep) {
        const int step = *dstep;
        tok = ((step == 0 || !((step - 1) & 1)) ? tok : tok1) + step;
    }
    const int64_t total = (int64_t)R * (D / 4);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int d4 = (int)(i % (D / 4));
        const int r = (int)(i / (D / 4));
        const int t = tok[(int64_t)r * tok_stride];
        reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(E + (int64_t)t * D)[d4];
    }
}

// ---- log_softmax + top-5 over the dictionary, one block per row ----
// ties resolve to the lower index.
// NJ > 0 (D <= 256 NJ): a thread's NJ elements (d = tid, tid + 256, ...) are loaded ONCE, all loads in flight together, and every pass
// runs on registers — the loop form (NJ == 0) re-reads the row per pass and its candidate insertion serialises the loads (24 dependent
// round trips per row at the 48px dictionary: 24 us per launch, 8 of it now).  Per thread the same elements in the same order, across
// threads the same pairing (tid, tid + s) for s = 128 .. 1: the max, the sum, the log-probabilities and the five winners are bitwise those
// of the loop form.
__device__ __forceinline__ bool top5_better(const float a, const int da, const float b, const int db) { return a > b || (a == b && da < db); }

template <int NJ>
__global__ __launch_bounds__(256) void logsoftmax_top5_kernel(const float *__restrict__ logits, int64_t ld, int D,
                                                               int suppress_tok, float *__restrict__ vals,
                                                               int *__restrict__ idx, float *__restrict__ logp_out) {
    __shared__ float red[256];
    __shared__ float cv[256 * 5];
    __shared__ int ci[256 * 5];
    __shared__ float bc[2];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *x = logits + (int64_t)r * ld;
    float tv[5];
    int ti[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        tv[j] = -INFINITY;
        ti[j] = 0x7fffffff;
    }
    float mx = -INFINITY, lse;
    if constexpr (NJ > 0) {
        float v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = tid + 256 * j;
            v[j] = d < D ? x[d] : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(v[j]));   // all NJ loads requested before the first use (read-only loads are sunk to their uses otherwise)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (tid + 256 * j == suppress_tok) v[j] = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = tid + 256 * j;
            if (d < D) {
                mx = fmaxf(mx, v[j]);
                if (top5_better(v[j], d, tv[4], ti[4])) {   // insertion = replace the last, bubble up (static register indices)
                    tv[4] = v[j], ti[4] = d;
#pragma unroll
                    for (int q = 4; q > 0; --q) {
                        if (top5_better(tv[q], ti[q], tv[q - 1], ti[q - 1])) {
                            const float fv = tv[q]; tv[q] = tv[q - 1]; tv[q - 1] = fv;
                            const int fi = ti[q]; ti[q] = ti[q - 1]; ti[q - 1] = fi;
                        }
                    }
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (tid + 256 * j < D) sum += expf(v[j] - mx);
        red[tid] = sum;
        __syncthreads();
        if (wave == 0) {   // the tree red[t] += red[t + s], s = 128 .. 1, with the last six levels inside the wave
            float sm = (red[lane] + red[lane + 128]) + (red[lane + 64] + red[lane + 192]);
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_down(sm, o);
            if (lane == 0) bc[0] = logf(sm);
        }
        __syncthreads();
        lse = bc[0];
        if (logp_out) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int d = tid + 256 * j;
                if (d < D) logp_out[(int64_t)r * D + d] = (v[j] - mx) - lse;
            }
        }
    } else {
        for (int d = tid; d < D; d += 256) {
            float v = x[d];
            if (d == suppress_tok) v = -INFINITY;
            mx = fmaxf(mx, v);
            if (v > tv[4] || (v == tv[4] && d < ti[4])) {
                int j = 4;
                while (j > 0 && (v > tv[j - 1] || (v == tv[j - 1] && d < ti[j - 1]))) {
                    tv[j] = tv[j - 1];
                    ti[j] = ti[j - 1];
                    --j;
                }
                tv[j] = v;
                ti[j] = d;
            }
        }
        red[tid] = mx;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
            __syncthreads();
        }
        mx = red[0];
        __syncthreads();
        float sum = 0.f;
        for (int d = tid; d < D; d += 256) {
            const float v = (d == suppress_tok) ? -INFINITY : x[d];
            sum += expf(v - mx);
        }
        red[tid] = sum;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        lse = logf(red[0]);
        if (logp_out) {
            for (int d = tid; d < D; d += 256) {
                const float v = (d == suppress_tok) ? -INFINITY : x[d];
                logp_out[(int64_t)r * D + d] = (v - mx) - lse;
            }
        }
    }
    for (int j = 0; j < 5; ++j) {
        cv[tid * 5 + j] = tv[j];
        ci[tid * 5 + j] = ti[j];
    }
    __syncthreads();
    // pairwise merges of the per-thread sorted candidate lists (value descending, lower index first on ties): the order is
    // total, so the surviving five are the same whatever the merge tree
    for (int s = 128; s > 0; s >>= 1) {
        float ov[5];
        int oi[5];
        if (tid < s) {
            const float *av = cv + tid * 5, *bv = cv + (tid + s) * 5;
            const int *ai = ci + tid * 5, *bi = ci + (tid + s) * 5;
            int ia = 0, ib = 0;
            for (int j = 0; j < 5; ++j) {
                const float va = av[ia], vb = bv[ib];
                const int da = ai[ia], db = bi[ib];
                const bool take_a = va > vb || (va == vb && da <= db);
                ov[j] = take_a ? va : vb;
                oi[j] = take_a ? da : db;
                ia += take_a ? 1 : 0;
                ib += take_a ? 0 : 1;
            }
        }
        __syncthreads();
        if (tid < s) {
            for (int j = 0; j < 5; ++j) {
                cv[tid * 5 + j] = ov[j];
                ci[tid * 5 + j] = oi[j];
            }
        }
        __syncthreads();
    }
    if (tid < 5) {
        vals[r * 5 + tid] = (cv[tid] - mx) - lse;  // log_softmax value
        idx[r * 5 + tid] = ci[tid];
    }
}

// ---- beam bookkeeping: one wave per sample (25 candidates on lanes 0 .. 24) ----
// The reference's bookkeeping (:693-699, :716-771) is integer work plus one float add per candidate; a thread per sample walked the
// 25 candidates and copied five histories by itself (42 us per step at one page).  Here the lanes of a wave hold the candidates, the
// top-5 are five wave-wide arg-max rounds (value descending, lower flat index first on ties — torch.topk's order on this path), the
// histories are copied lane-parallel.  `next` (optional): the embedding rows of the tokens just chosen, written as the residual stream of
// the NEXT step (what embed_kernel would do as the first launch of that step).
struct BeamNextEmbed {
    const float *E;   // embedding table [dict][D]; NULL: none
    float *out;       // [R][D]
    int D;
};

__device__ __forceinline__ void beam_embed_rows(const BeamNextEmbed &next, const int n, const int (&tok)[5], const int lane) {
    if (!next.E) return;
    const int D4 = next.D >> 2;
    for (int i = lane; i < 5 * D4; i += 64) {
        const int k = i / D4, d4 = i - k * D4;
        const int t = k == 0 ? tok[0] : k == 1 ? tok[1] : k == 2 ? tok[2] : k == 3 ? tok[3] : tok[4];
        reinterpret_cast<float4 *>(next.out + (int64_t)(n * 5 + k) * next.D)[d4] = reinterpret_cast<const float4 *>(next.E + (int64_t)t * next.D)[d4];
    }
}

// step 0 (:693-699): beam j of a sample takes the j-th best token of its (identical) row.
__device__ __forceinline__ void beam_init_body(const float *__restrict__ vals, const int *__restrict__ idx, int *__restrict__ hist,
                                               int hist_ld, float *__restrict__ logp, int N, int start_tok, const BeamNextEmbed next) {
    const int n = blockIdx.x, lane = threadIdx.x;
    if (n >= N) return;
    int tok[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) tok[j] = idx[(n * 5) * 5 + j];
    if (lane < 5) {
        const int row = n * 5 + lane;
        hist[(int64_t)row * hist_ld + 0] = start_tok;
        hist[(int64_t)row * hist_ld + 1] = idx[(n * 5) * 5 + lane];
        logp[row] = vals[(n * 5) * 5 + lane];
    }
    beam_embed_rows(next, n, tok, lane);
}

__global__ __launch_bounds__(64) void beam_init_kernel(const float *__restrict__ vals, const int *__restrict__ idx, int *__restrict__ hist,
                                                       int hist_ld, float *__restrict__ logp, int N, int start_tok, BeamNextEmbed next) {
    beam_init_body(vals, idx, hist, hist_ld, logp, N, start_tok, next);
}

// steps >= 1 (:716-771). hist_in/out [R][hist_ld]: tokens 0..step valid on input, 0..step+1 on output.
__device__ __forceinline__ void beam_step_body(const float *__restrict__ vals, const int *__restrict__ idx,
                                               const int *__restrict__ hist_in, int *__restrict__ hist_out, int hist_ld,
                                               const float *__restrict__ logp_in, float *__restrict__ logp_out, int *__restrict__ done,
                                               int *__restrict__ res_row, int *__restrict__ res_len, float *__restrict__ res_prob,
                                               int *__restrict__ res_tok, int *__restrict__ done_count, int N, int step, int end_tok,
                                               int max_finished, const BeamNextEmbed next) {
    const int n = blockIdx.x, lane = threadIdx.x;
    if (n >= N) return;
    const bool was_done = done[n] != 0;   // (every lane reads it before lane 0 of this wave — its only writer — may set it)
    float cl = -INFINITY;   // candidate c = b * 5 + j on lane c: beam b's j-th best continuation
    int ct = 0;
    if (lane < 25) {
        const int b = lane / 5, j = lane - b * 5, row = n * 5 + b;
        const bool fin = hist_in[(int64_t)row * hist_ld + step] == end_tok;
        const float v = fin ? 0.f : vals[row * 5 + j];
        ct = fin ? end_tok : idx[row * 5 + j];
        cl = logp_in[row] + v;
    }
    bool used = lane >= 25;
    int sel[5], tok[5];
    float sl[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // topk, descending, lower flat index first on ties
        float v = used ? -INFINITY : cl;
        int c = used ? 1 << 20 : lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o);
            const int oc = __shfl_xor(c, o);
            if (ov > v || (ov == v && oc < c)) v = ov, c = oc;
        }
        sel[k] = c;
        if (lane == c) used = true;
        sl[k] = __shfl(cl, c);
        tok[k] = __shfl(ct, c);
    }
    int fcount = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int row = n * 5 + k;
        const int src = n * 5 + sel[k] / 5;
        for (int t = lane; t <= step; t += 64) hist_out[(int64_t)row * hist_ld + t] = hist_in[(int64_t)src * hist_ld + t];
        if (lane == 0) {
            hist_out[(int64_t)row * hist_ld + step + 1] = tok[k];
            logp_out[row] = sl[k];
        }
        fcount += tok[k] == end_tok;
    }
    if (!was_done && fcount >= max_finished) {
        const int row = n * 5 + 0;  // argmax of the (descending) top-k
        const int src = n * 5 + sel[0] / 5;
        for (int t = lane; t <= step; t += 64) res_tok[(int64_t)n * hist_ld + t] = hist_in[(int64_t)src * hist_ld + t];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            res_tok[(int64_t)n * hist_ld + step + 1] = tok[0];
            res_row[n] = row;
            res_len[n] = step + 2;
            res_prob[n] = expf(sl[0]);
            done[n] = 1;
            atomicAdd(done_count, 1);
        }
    }
    beam_embed_rows(next, n, tok, lane);
}

__global__ __launch_bounds__(64) void beam_step_kernel(const float *__restrict__ vals, const int *__restrict__ idx,
                                 const int *__restrict__ hist_in, int *__restrict__ hist_out, int hist_ld,
                                 const float *__restrict__ logp_in, float *__restrict__ logp_out, int *__restrict__ done,
                                 int *__restrict__ res_row, int *__restrict__ res_len, float *__restrict__ res_prob,
                                 int *__restrict__ res_tok, int *__restrict__ done_count, int N, int step, int end_tok,
                                 int max_finished, BeamNextEmbed next) {
    beam_step_body(vals, idx, hist_in, hist_out, hist_ld, logp_in, logp_out, done, res_row, res_len, res_prob, res_tok, done_count, N, step,
                   end_tok, max_finished, next);
}

// The same bookkeeping with the step read from device memory (one launch sequence serves every step, so it can be replayed from a
// hipGraph): step 0 initialises buffer 0; step s >= 1 reads buffer (s - 1) & 1 and writes buffer s & 1.
__global__ __launch_bounds__(64) void beam_dyn_kernel(const float *__restrict__ vals, const int *__restrict__ idx, int *__restrict__ hist0, int *__restrict__ hist1,
                                int hist_ld, float *__restrict__ logp0, float *__restrict__ logp1, int *__restrict__ done,
                                int *__restrict__ res_row, int *__restrict__ res_len, float *__restrict__ res_prob, int *__restrict__ res_tok,
                                int *__restrict__ done_count, int N, const int *__restrict__ dstep, int start_tok, int end_tok, int max_finished,
                                BeamNextEmbed next) {
    const int step = *dstep;
    if (step == 0) {
        beam_init_body(vals, idx, hist0, hist_ld, logp0, N, start_tok, next);
        return;
    }
    const bool odd_in = ((step - 1) & 1) != 0;
    beam_step_body(vals, idx, odd_in ? hist1 : hist0, odd_in ? hist0 : hist1, hist_ld, odd_in ? logp1 : logp0, odd_in ? logp0 : logp1, done,
                   res_row, res_len, res_prob, res_tok, done_count, N, step, end_tok, max_finished, next);
}

__global__ void step_advance_kernel(int *__restrict__ dstep) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *dstep += 1;
}

// fallback (:774-784): samples that never finished take the first row of their beam.
__global__ void beam_finalize_kernel(const int *__restrict__ hist, int hist_ld, const float *__restrict__ logp,
                                     int *__restrict__ done, int *__restrict__ res_row, int *__restrict__ res_len,
                                     float *__restrict__ res_prob, int *__restrict__ res_tok, int N, int len) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N || done[n]) return;
    const int row = n * 5;
    res_row[n] = row;
    res_len[n] = len;
    res_prob[n] = expf(logp[row]);
    for (int t = 0; t < len; ++t) res_tok[(int64_t)n * hist_ld + t] = hist[(int64_t)row * hist_ld + t];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// internal launch helpers (shared with ocr_decoder.hip) + C-ABI wrappers
// ---------------------------------------------------------------------------------------------

int ocrk_layernorm(const float *in, int64_t in_rs, const float *w, const float *b, float *out, int64_t out_rs, int rows,
                   int D, float eps, hipStream_t s, const OcrPlanes *planes) {
    if (D <= 0 || D > 512) return mit_set_error("ocrk_layernorm: D out of range (got %d; a lane holds 8 values)", D);
    if (planes && (!planes->p || (D & 7)))  // the planar form stages a row in 512 floats of LDS and writes cells of 8
        return mit_set_error("ocrk_layernorm: planar output needs a plane buffer and D %% 8 == 0 (got %d)", D);
    const int waves = 4;
    MitProbeScope probe("layernorm_kernel", s, (planes ? 10.0 : 8.0) * (double)rows * D);
    if (planes)
        hipLaunchKernelGGL(layernorm_kernel<true>, dim3((rows + waves - 1) / waves), dim3(64 * waves), 0, s, in, in_rs, w, b, out, out_rs,
                           rows, D, eps, *planes);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, dim3((rows + waves - 1) / waves), dim3(64 * waves), 0, s, in, in_rs, w, b, out, out_rs,
                           rows, D, eps, OcrPlanes{nullptr, 0, 0});
    return 0;
}

void ocrk_xpos_rotate(const float *in, int64_t in_rs, int64_t in_ts, float *out, int64_t out_rs, int64_t out_ts, int R, int T,
                      int i0, int p0, int downscale, const MitXposTables &tb, hipStream_t s, const int *dstep, int dyn_mode,
                      int64_t dyn_in) {
    const int64_t total = (int64_t)R * T * 160;
    MitProbeScope probe("xpos_rotate_kernel", s, 8.0 * (double)R * T * 320);
    hipLaunchKernelGGL(xpos_rotate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, in_rs, in_ts, out, out_rs, out_ts, R,
                       T, i0, p0, downscale, tb.cos_t, tb.sin_t, tb.scale_t, tb.iscale_t, tb.pmax, dstep, dyn_mode, dyn_in);
}

// ---- the same attention for ALL query positions of a row (the encoder's self-attention: Tq = Tk = the line's memory length): one
// workgroup per (head, row, block of 32 queries) stages the row's keys and values in LDS once — attention_kernel re-reads them
// from L2 for every query (32 x the bytes) — and each wave takes eight queries, so a key / value element read from LDS feeds eight
// dot products / weighted sums.  Per (query, key) dot product, per query the lane-strided softmax sums, per (query, d) the t-ordered
// weighted sum: evaluated in exactly the order of attention_kernel, so both give bitwise identical results.
constexpr int ATTR_GQ = 8;
constexpr int ATTR_THREADS = 256;

__global__ __launch_bounds__(ATTR_THREADS) void attention_rows_kernel(const float *__restrict__ Q, int64_t q_rs, int64_t q_ts,
                                                                      const float *__restrict__ K, int64_t k_rs, int64_t k_ts,
                                                                      const float *__restrict__ V, int64_t v_rs, int64_t v_ts,
                                                                      float *__restrict__ O, int64_t o_rs, int64_t o_ts,
                                                                      const int *__restrict__ klen, int Tq, int Tk, int HD) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KP = HD + 4;  // 16-byte aligned rows; lane t reads row t as float4s
    const int HD4 = HD / 4;
    float *kv = lds;                                   // [Tk][KP]: the keys, then (same buffer) the values
    float *qs_all = kv + (size_t)Tk * KP;              // [waves][GQ][HD]
    float *ws_all = qs_all + (ATTR_THREADS / 64) * ATTR_GQ * HD;  // [waves][GQ][Tk]
    const int h = blockIdx.x, r = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int valid = klen ? min(klen[r], Tk) : Tk;
    auto stage = [&](const float *base, int64_t ts) {
        for (int i = tid; i < valid * HD4; i += ATTR_THREADS) {
            const int t = i / HD4, d4 = i - t * HD4;
            *reinterpret_cast<float4 *>(kv + t * KP + d4 * 4) = *reinterpret_cast<const float4 *>(base + (int64_t)t * ts + d4 * 4);
        }
    };
    stage(K + (int64_t)r * k_rs + h * HD, k_ts);
    float *qs = qs_all + wave * ATTR_GQ * HD;
    float *ws = ws_all + (size_t)wave * ATTR_GQ * Tk;
    // blockIdx.z: this workgroup's block of 4 x 8 query positions (the grid, not a loop, covers the row's queries — a chunk of 16 lines
    // x 4 heads alone would leave three quarters of the CUs without a workgroup)
    const int qa = ((int)blockIdx.z * (ATTR_THREADS / 64) + wave) * ATTR_GQ;
    const int ng = max(0, min(ATTR_GQ, Tq - qa));
    for (int i = lane; i < ATTR_GQ * HD; i += 64) {
        const int g = i / HD, d = i - g * HD;
        qs[i] = g < ng ? Q[(int64_t)r * q_rs + (int64_t)(qa + g) * q_ts + h * HD + d] : 0.f;
    }
    __syncthreads();  // keys and queries staged
    float mx[ATTR_GQ];
#pragma unroll
    for (int g = 0; g < ATTR_GQ; ++g) mx[g] = -INFINITY;
    for (int t = lane; t < Tk; t += 64) {
        float dot[ATTR_GQ];
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g) dot[g] = t < valid ? 0.f : -INFINITY;
        if (t < valid) {
            const float4 *kp = reinterpret_cast<const float4 *>(kv + t * KP);
            for (int d4 = 0; d4 < HD4; ++d4) {  // d ascending, one rounding per product and per sum, as in attention_kernel
                const float4 kk = kp[d4];
#pragma unroll
                for (int g = 0; g < ATTR_GQ; ++g) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qs + g * HD + d4 * 4);  // same address in every lane: broadcast
                    dot[g] += qv.x * kk.x;
                    dot[g] += qv.y * kk.y;
                    dot[g] += qv.z * kk.z;
                    dot[g] += qv.w * kk.w;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g) {
            ws[(size_t)g * Tk + t] = dot[g];
            mx[g] = fmaxf(mx[g], dot[g]);
        }
    }
    __syncthreads();  // every wave is done with the keys; this wave's scores are visible to all its lanes
    stage(V + (int64_t)r * v_rs + h * HD, v_ts);  // the values take the keys' place (their loads fly while the softmax runs)
#pragma unroll
    for (int g = 0; g < ATTR_GQ; ++g) {  // softmax: lanes strided over the keys exactly as attention_kernel's single wave
        float *w = ws + (size_t)g * Tk;
        float m = mx[g];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int t = lane; t < Tk; t += 64) {
            const float e = expf(w[t] - m);
            w[t] = e;
            sum += e;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        for (int t = lane; t < Tk; t += 64) w[t] *= inv;
    }
    __syncthreads();
    for (int d = lane; d < HD; d += 64) {  // weighted sum of the values, t-ordered per (query, d)
        float acc[ATTR_GQ];
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g) acc[g] = 0.f;
        for (int t = 0; t < valid; ++t) {
            const float v = kv[t * KP + d];
#pragma unroll
            for (int g = 0; g < ATTR_GQ; ++g) acc[g] += ws[(size_t)g * Tk + t] * v;
        }
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g)
            if (g < ng) O[(int64_t)r * o_rs + (int64_t)(qa + g) * o_ts + h * HD + d] = acc[g];
    }
}

// ---- the encoder's self-attention for ALL lines of a page group in one launch, with the XPOS rotation of q and k folded in:
// attention_rows_kernel's arithmetic on rows that are rotated while they are staged in LDS instead of by two xpos_rotate_kernel
// launches per chunk (the rotated value of an element is the same fp32 expression, so the result is bitwise the one of
// rotate + rotate + attention_rows_kernel chunk by chunk).  Lines are ragged: line r of the group owns L_r consecutive rows of the flat
// [rows, heads * HD] q / k / v / o tensors starting at row start_r (its chunk's padded memory length L_r is its query AND key count;
// positions are centred per chunk: p0 = -((L_r + 1) / 2), model_48px.py:327-394, xpos_relative_position.py:44-71).
struct OcrLine {
    int start, L;
};
__global__ __launch_bounds__(ATTR_THREADS) void attention_lines_xpos_kernel(const float *__restrict__ Q, const float *__restrict__ K,
                                                                            const float *__restrict__ V, float *__restrict__ O, int64_t ts,
                                                                            const OcrLine *__restrict__ lines, const int *__restrict__ klen,
                                                                            int Tmax, int HD, const float *__restrict__ cosT,
                                                                            const float *__restrict__ sinT, const float *__restrict__ scaleT,
                                                                            const float *__restrict__ iscaleT, int pmax) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KP = HD + 4;
    const int HD4 = HD / 4, HP = HD / 2;  // HP: rotation pairs per head (table row length)
    const int h = blockIdx.x, r = blockIdx.y;
    const OcrLine ln = lines[r];
    const int Tk = ln.L, Tq = ln.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qa = ((int)blockIdx.z * (ATTR_THREADS / 64) + wave) * ATTR_GQ;
    if ((int)blockIdx.z * (ATTR_THREADS / 64) * ATTR_GQ >= Tq) return;  // whole workgroup: this line is shorter than the group's longest
    float *kv = lds;                                              // [Tmax][KP]: the keys, then (same buffer) the values
    float *qs_all = kv + (size_t)Tmax * KP;                       // [waves][GQ][HD]
    float *ws_all = qs_all + (ATTR_THREADS / 64) * ATTR_GQ * HD;  // [waves][GQ][Tmax]
    const int valid = klen ? min(klen[r], Tk) : Tk;
    const int p0 = -((ln.L + 1) / 2);
    const float *kb = K + (int64_t)ln.start * ts + h * HD, *vb = V + (int64_t)ln.start * ts + h * HD, *qb = Q + (int64_t)ln.start * ts + h * HD;
    auto rot2 = [&](const float2 x, const int t, const int j, const float *sct) {  // xpos_rotate_kernel's expression for pair j at position t
        const int pp = p0 + t + pmax;
        const float sc = sct[pp * HP + j];
        const float c = cosT[t * HP + j] * sc;
        const float sn = sinT[t * HP + j] * sc;
        float2 o;
        o.x = x.x * c + (-x.y) * sn;
        o.y = x.y * c + x.x * sn;
        return o;
    };
    for (int i = tid; i < valid * HD4; i += ATTR_THREADS) {  // keys, rotated with the inverse scale
        const int t = i / HD4, d4 = i - t * HD4;
        const float4 x = *reinterpret_cast<const float4 *>(kb + (int64_t)t * ts + d4 * 4);
        const float2 a = rot2(make_float2(x.x, x.y), t, 2 * d4, iscaleT), b = rot2(make_float2(x.z, x.w), t, 2 * d4 + 1, iscaleT);
        *reinterpret_cast<float4 *>(kv + t * KP + d4 * 4) = make_float4(a.x, a.y, b.x, b.y);
    }
    float *qs = qs_all + wave * ATTR_GQ * HD;
    float *ws = ws_all + (size_t)wave * ATTR_GQ * Tmax;
    const int ng = max(0, min(ATTR_GQ, Tq - qa));
    for (int i = lane; i < ATTR_GQ * HP; i += 64) {  // this wave's queries, rotated with the scale
        const int g = i / HP, j = i - g * HP;
        float2 o = make_float2(0.f, 0.f);
        if (g < ng) o = rot2(*reinterpret_cast<const float2 *>(qb + (int64_t)(qa + g) * ts + 2 * j), qa + g, j, scaleT);
        *reinterpret_cast<float2 *>(qs + g * HD + 2 * j) = o;
    }
    __syncthreads();  // keys and queries staged
    float mx[ATTR_GQ];
#pragma unroll
    for (int g = 0; g < ATTR_GQ; ++g) mx[g] = -INFINITY;
    for (int t = lane; t < Tk; t += 64) {
        float dot[ATTR_GQ];
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g) dot[g] = t < valid ? 0.f : -INFINITY;
        if (t < valid) {
            const float4 *kp = reinterpret_cast<const float4 *>(kv + t * KP);
            for (int d4 = 0; d4 < HD4; ++d4) {  // d ascending, one rounding per product and per sum, as in attention_kernel
                const float4 kk = kp[d4];
#pragma unroll
                for (int g = 0; g < ATTR_GQ; ++g) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qs + g * HD + d4 * 4);
                    dot[g] += qv.x * kk.x;
                    dot[g] += qv.y * kk.y;
                    dot[g] += qv.z * kk.z;
                    dot[g] += qv.w * kk.w;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g) {
            ws[(size_t)g * Tmax + t] = dot[g];
            mx[g] = fmaxf(mx[g], dot[g]);
        }
    }
    __syncthreads();  // every wave is done with the keys
    for (int i = tid; i < valid * HD4; i += ATTR_THREADS) {  // the values take the keys' place
        const int t = i / HD4, d4 = i - t * HD4;
        *reinterpret_cast<float4 *>(kv + t * KP + d4 * 4) = *reinterpret_cast<const float4 *>(vb + (int64_t)t * ts + d4 * 4);
    }
#pragma unroll
    for (int g = 0; g < ATTR_GQ; ++g) {  // softmax: lanes strided over the keys exactly as attention_kernel's single wave
        float *w = ws + (size_t)g * Tmax;
        float m = mx[g];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int t = lane; t < Tk; t += 64) {
            const float e = expf(w[t] - m);
            w[t] = e;
            sum += e;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        for (int t = lane; t < Tk; t += 64) w[t] *= inv;
    }
    __syncthreads();
    for (int d = lane; d < HD; d += 64) {  // weighted sum of the values, t-ordered per (query, d)
        float acc[ATTR_GQ];
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g) acc[g] = 0.f;
        for (int t = 0; t < valid; ++t) {
            const float v = kv[t * KP + d];
#pragma unroll
            for (int g = 0; g < ATTR_GQ; ++g) acc[g] += ws[(size_t)g * Tmax + t] * v;
        }
#pragma unroll
        for (int g = 0; g < ATTR_GQ; ++g)
            if (g < ng) O[(int64_t)(ln.start + qa + g) * ts + h * HD + d] = acc[g];
    }
}

// ---- cross-attention memory of one decoder layer for all lines of a group: mem_k[line, t, :] = XPOS-rotated (inverse scale, centred per
// chunk) k rows, mem_v[line, t, :] = v rows, t < L_line; replaces one xpos_rotate_kernel launch and one copy per chunk.  One thread per
// (row of the flat k / v, rotation pair).
__global__ void memory_kv_lines_kernel(const float *__restrict__ Kf, const float *__restrict__ Vf, int64_t ts, float *__restrict__ mem_k,
                                       float *__restrict__ mem_v, int64_t line_stride, const OcrLine *__restrict__ lines, int n_lines,
                                       int first_line, int HP, int pairs_per_row, const float *__restrict__ cosT,
                                       const float *__restrict__ sinT, const float *__restrict__ iscaleT, int pmax, int64_t total) {
    const int row0 = lines[0].start;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pair = (int)(i % pairs_per_row);
        const int64_t row = i / pairs_per_row;  // row of the flat tensors, relative to the group's first row
        int lo = 0, hi = n_lines - 1;           // the line owning this row (lines are consecutive row ranges)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (lines[mid].start - row0 <= row) lo = mid; else hi = mid - 1;
        }
        const OcrLine ln = lines[lo];
        const int t = (int)(row - (ln.start - row0));
        const int j = pair % HP;
        const int pp = -((ln.L + 1) / 2) + t + pmax;
        const float sc = iscaleT[pp * HP + j];
        const float c = cosT[t * HP + j] * sc, sn = sinT[t * HP + j] * sc;
        const float2 x = *reinterpret_cast<const float2 *>(Kf + (row0 + row) * ts + pair * 2);
        float2 o;
        o.x = x.x * c + (-x.y) * sn;
        o.y = x.y * c + x.x * sn;
        const int64_t dst = (int64_t)(first_line + lo) * line_stride + (int64_t)t * ts + pair * 2;
        *reinterpret_cast<float2 *>(mem_k + dst) = o;
        *reinterpret_cast<float2 *>(mem_v + dst) = *reinterpret_cast<const float2 *>(Vf + (row0 + row) * ts + pair * 2);
    }
}

// the decoder's self-attention on attention_self_kernel (1, default; MIT_ATT_NO_SELF in the environment starts with 0) or on
// attention_kernel (0): same bits either way (tests/test_ocr_gpu.py), a switch for A/B runs
static std::atomic<int> g_att_self_rows{getenv("MIT_ATT_NO_SELF") ? 0 : 1};
extern "C" int mit_attention_self_rows_set(int on) {
    const int prev = g_att_self_rows.load(std::memory_order_relaxed);
    if (on >= 0) g_att_self_rows.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

void ocrk_attention(const float *Q, int64_t q_rs, int64_t q_ts, const float *K, int64_t k_rs, int64_t k_ts, const float *V,
                    int64_t v_rs, int64_t v_ts, float *O, int64_t o_rs, int64_t o_ts, const int *klen, int R, int Tq, int Tk,
                    int kv_div, hipStream_t s, int heads, int head_dim, const int *dstep, const OcrAttXpos *xpos, const OcrPlanes *o_planes) {
    OcrAttXpos xp{};
    if (xpos) xp = *xpos;
    const OcrPlanes opl = (o_planes && Tq == 1) ? *o_planes : OcrPlanes{nullptr, 0, 0};
    // (the shared-K/V form never shortens Tk by the step counter: with a step counter it is only used for the rotated query)
    if ((!dstep || (xp.cos_t && !xp.rot_k)) && Tq == 1 && kv_div == 5 && R % kv_div == 0 && R / kv_div <= 65535 && head_dim == 80) {  // the 48px decoder's cross-attention: 5 beams, 4 x 80
        // algorithmic bytes: K and V of each line read once for its kv_div beams (+ q in, o out); FLOPs 4 Tk d per query row and head
        const double bytes = 4.0 * heads * head_dim * ((double)(R / kv_div) * 2.0 * Tk + 2.0 * R), flops = 4.0 * (double)R * heads * Tk * head_dim;
        auto lds_bytes = [&](const int kchunk) { return ((size_t)kv_div * head_dim + (size_t)kchunk * (head_dim + 4) + (size_t)kv_div * Tk) * sizeof(float); };
        // few workgroups (one page .. a few): the latency form, a line's keys in one chunk; else the throughput form.  Same arithmetic.
        static const int64_t lat_max = getenv("MIT_ATT_LATENCY_MAX_WGS") ? atoll(getenv("MIT_ATT_LATENCY_MAX_WGS")) : 512;
        if ((int64_t)heads * (R / kv_div) <= lat_max && ATT_KCHUNK_L * (head_dim / 4) <= ATT_STAGE_L * ATT_THREADS && lds_bytes(ATT_KCHUNK_L) <= 64 * 1024) {
            MitProbeScope probe("attention_shared_kv_kernel", s, bytes, flops);
            hipLaunchKernelGGL((attention_shared_kv_kernel<ATT_KCHUNK_L, ATT_STAGE_L, 80, 5>), dim3(heads, R / kv_div), dim3(ATT_THREADS), lds_bytes(ATT_KCHUNK_L), s, Q, q_rs,
                               K, k_rs, k_ts, V, v_rs, v_ts, O, o_rs, klen, Tk, dstep, xp, opl, OcrAttQProj{});
            return;
        }
        if (ATT_KCHUNK * (head_dim / 4) <= ATT_STAGE * ATT_THREADS && lds_bytes(ATT_KCHUNK) <= 64 * 1024) {
            MitProbeScope probe("attention_shared_kv_kernel", s, bytes, flops);
            hipLaunchKernelGGL((attention_shared_kv_kernel<ATT_KCHUNK, ATT_STAGE, 80, 5>), dim3(heads, R / kv_div), dim3(ATT_THREADS), lds_bytes(ATT_KCHUNK), s, Q, q_rs, K,
                               k_rs, k_ts, V, v_rs, v_ts, O, o_rs, klen, Tk, dstep, xp, opl, OcrAttQProj{});
            return;
        }
    }
    static const bool no_rows = getenv("MIT_ATT_NO_ROWS") != nullptr;
    if (!no_rows && !dstep && !xp.cos_t && !opl.p && kv_div == 1 && Tq >= 2 * ATTR_GQ && R <= 65535 && Tq <= 65535 * 32 && head_dim <= 128 && ((q_rs | q_ts | k_rs | k_ts | v_rs | v_ts) & 3) == 0) {
        const size_t sm = ((size_t)Tk * (head_dim + 4) + (size_t)(ATTR_THREADS / 64) * ATTR_GQ * (head_dim + Tk)) * sizeof(float);
        if (sm <= 150 * 1024 && head_dim % 8 == 0) {
            // algorithmic bytes: q, k, v read once and o written once per row; FLOPs 4 Tk d per query and head
            MitProbeScope probe("attention_rows_kernel", s, 4.0 * (double)R * heads * head_dim * (2.0 * Tq + 2.0 * Tk),
                                4.0 * (double)R * Tq * heads * Tk * head_dim);
            static size_t granted[16] = {0};
            if (sm > 64 * 1024) {
                int dev = 0;
                (void)hipGetDevice(&dev);
                if (granted[dev & 15] < sm) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attention_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
                    granted[dev & 15] = sm;
                }
            }
            hipLaunchKernelGGL(attention_rows_kernel, dim3(heads, R, (Tq + (ATTR_THREADS / 64) * ATTR_GQ - 1) / ((ATTR_THREADS / 64) * ATTR_GQ)), dim3(ATTR_THREADS), sm, s, Q, q_rs, q_ts, K, k_rs, k_ts, V, v_rs, v_ts, O, o_rs,
                               o_ts, klen, Tq, Tk, head_dim);
            return;
        }
    }
    if (g_att_self_rows.load(std::memory_order_relaxed) && Tq == 1 && kv_div == 1 && !klen && xp.cos_t && xp.rot_k && heads == 4 && head_dim == 80 && k_ts == (int64_t)heads * head_dim && v_ts == k_ts &&
        !((q_rs | k_rs | k_ts | v_rs | v_ts) & 3) && !((reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(Q)) & 15) && ((heads * Tk) & 3) == 0) {
        const size_t sm = ((size_t)heads * head_dim + (size_t)heads * Tk + (size_t)Tk * (heads * head_dim + 4) + (size_t)2 * Tk * (head_dim / 2 + SELF_TP_PAD)) * sizeof(float);
        if (sm <= 64 * 1024) {
            // the decoder's self-attention: the row's key history staged once for its four heads (bitwise attention_kernel's results)
            MitProbeScope probe("attention_self_kernel", s, 4.0 * heads * head_dim * ((double)R * 2.0 * Tk + 2.0 * (double)R),
                                4.0 * (double)R * heads * Tk * head_dim);
            hipLaunchKernelGGL((attention_self_kernel<4, 80>), dim3(R), dim3(256), sm, s, K, k_rs, V, v_rs, Q, q_rs, Tk, dstep, O, o_rs, xp, opl);
            return;
        }
    }
    const size_t smem = ((size_t)head_dim + (size_t)Tk) * sizeof(float);
    MitProbeScope probe("attention_kernel", s, 4.0 * heads * head_dim * ((double)(R / kv_div) * 2.0 * Tk + 2.0 * (double)R * Tq),
                        4.0 * (double)R * Tq * heads * Tk * head_dim);
    hipLaunchKernelGGL(attention_kernel, dim3(Tq, heads, R), dim3(64), smem, s, Q, q_rs, q_ts, K, k_rs, k_ts, V, v_rs, v_ts, O, o_rs,
                       o_ts, klen, Tk, kv_div, head_dim, dstep, xp, opl);   // dstep: Tk = the LDS capacity, the kernel attends to *dstep + 1 keys
}

// The decoder's cross-attention with its q projection inside (QF form above).  Returns false — nothing launched — when the problem is
// not the few-row case the form exists for (the caller then runs the Linear and ocrk_attention).
bool ocrk_cross_attention_qproj(const OcrAttQProj &qp, const float *K, int64_t k_rs, int64_t k_ts, const float *V, int64_t v_rs, int64_t v_ts,
                                const int *klen, int R, int Tk, hipStream_t s, const int *dstep, const OcrAttXpos *xpos, const OcrPlanes *o_planes) {
    constexpr int heads = 4, head_dim = 80, kv_div = 5;
    OcrAttXpos xp{};
    if (xpos) xp = *xpos;
    if (!o_planes || !o_planes->p || !qp.x || !qp.w_planes || !qp.bias || !qp.ln_w || !qp.ln_b || (qp.ldx & 3) || xp.rot_k) return false;
    if ((reinterpret_cast<uintptr_t>(qp.x) | reinterpret_cast<uintptr_t>(qp.w_planes) | reinterpret_cast<uintptr_t>(qp.ln_w) | reinterpret_cast<uintptr_t>(qp.ln_b)) & 15) return false;
    if ((uint64_t)3 * (mitln::LN_K / 8) * (uint64_t)qp.ldw * 16u >= (1ull << 32)) return false;
    if (R % kv_div || R / kv_div > 65535) return false;
    static const int64_t lat_max = getenv("MIT_ATT_LATENCY_MAX_WGS") ? atoll(getenv("MIT_ATT_LATENCY_MAX_WGS")) : 512;
    const size_t lds_bytes = ((size_t)kv_div * head_dim + (size_t)ATT_KCHUNK_L * (head_dim + 4) + (size_t)kv_div * Tk) * sizeof(float);
    if ((int64_t)heads * (R / kv_div) > lat_max || lds_bytes > 64 * 1024) return false;
    const double bytes = 4.0 * heads * head_dim * ((double)(R / kv_div) * 2.0 * Tk + 2.0 * R), flops = 4.0 * (double)R * heads * Tk * head_dim;
    MitProbeScope probe("attention_shared_kv_kernel", s, bytes, flops + 2.0 * R * 320.0 * 320.0);
    hipLaunchKernelGGL((attention_shared_kv_kernel<ATT_KCHUNK_L, ATT_STAGE_L, 80, 5, true>), dim3(heads, R / kv_div), dim3(ATT_THREADS), lds_bytes, s, nullptr, 0,
                       K, k_rs, k_ts, V, v_rs, v_ts, nullptr, 0, klen, Tk, dstep, xp, *o_planes, qp);
    return true;
}

void ocrk_embed(const int *tok, int64_t tok_stride, const float *E, float *out, int R, int D, hipStream_t s, const int *tok1,
                const int *dstep) {
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for((int64_t)R * D / 4, 256)), dim3(256), 0, s, tok, tok_stride, E, out, R, D, tok1, dstep);
}

void ocrk_beam_dyn(const float *vals, const int *idx, int *hist0, int *hist1, int hist_ld, float *logp0, float *logp1, int *done,
                   int *res_row, int *res_len, float *res_prob, int *res_tok, int *done_count, int N, const int *dstep, int start_tok,
                   int end_tok, int max_finished, hipStream_t s, const float *next_E, float *next_out, int next_D) {
    hipLaunchKernelGGL(beam_dyn_kernel, dim3(N), dim3(64), 0, s, vals, idx, hist0, hist1, hist_ld, logp0, logp1, done, res_row,
                       res_len, res_prob, res_tok, done_count, N, dstep, start_tok, end_tok, max_finished, BeamNextEmbed{next_E, next_out, next_D});
}

void ocrk_step_advance(int *dstep, hipStream_t s) { hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, dstep); }

void ocrk_logsoftmax_top5(const float *logits, int64_t ld, int R, int D, int suppress_tok, float *vals, int *idx,
                          float *logp_out, hipStream_t s) {
    MitProbeScope probe("logsoftmax_top5_kernel", s, 4.0 * (double)R * D + (logp_out ? 4.0 * (double)R * D : 0.0));
    const char *loop_form = getenv("MIT_OCR_TOP5_LOOP");   // the loop form for every D (tests: both forms bit for bit)
    if (loop_form && atoi(loop_form))
        hipLaunchKernelGGL(logsoftmax_top5_kernel<0>, dim3(R), dim3(256), 0, s, logits, ld, D, suppress_tok, vals, idx, logp_out);
    else if (D <= 256 * 8)
        hipLaunchKernelGGL(logsoftmax_top5_kernel<8>, dim3(R), dim3(256), 0, s, logits, ld, D, suppress_tok, vals, idx, logp_out);
    else if (D <= 256 * 24)
        hipLaunchKernelGGL(logsoftmax_top5_kernel<24>, dim3(R), dim3(256), 0, s, logits, ld, D, suppress_tok, vals, idx, logp_out);
    else
        hipLaunchKernelGGL(logsoftmax_top5_kernel<0>, dim3(R), dim3(256), 0, s, logits, ld, D, suppress_tok, vals, idx, logp_out);
}

void ocrk_beam_init(const float *vals, const int *idx, int *hist, int hist_ld, float *logp, int N, int start_tok,
                    hipStream_t s, const float *next_E, float *next_out, int next_D) {
    hipLaunchKernelGGL(beam_init_kernel, dim3(N), dim3(64), 0, s, vals, idx, hist, hist_ld, logp, N, start_tok, BeamNextEmbed{next_E, next_out, next_D});
}

void ocrk_beam_step(const float *vals, const int *idx, const int *hist_in, int *hist_out, int hist_ld, const float *logp_in,
                    float *logp_out, int *done, int *res_row, int *res_len, float *res_prob, int *res_tok, int *done_count,
                    int N, int step, int end_tok, int max_finished, hipStream_t s, const float *next_E, float *next_out, int next_D) {
    hipLaunchKernelGGL(beam_step_kernel, dim3(N), dim3(64), 0, s, vals, idx, hist_in, hist_out, hist_ld, logp_in,
                       logp_out, done, res_row, res_len, res_prob, res_tok, done_count, N, step, end_tok, max_finished,
                       BeamNextEmbed{next_E, next_out, next_D});
}

void ocrk_beam_finalize(const int *hist, int hist_ld, const float *logp, int *done, int *res_row, int *res_len,
                        float *res_prob, int *res_tok, int N, int len, hipStream_t s) {
    hipLaunchKernelGGL(beam_finalize_kernel, dim3((N + 63) / 64), dim3(64), 0, s, hist, hist_ld, logp, done, res_row, res_len,
                       res_prob, res_tok, N, len);
}

extern "C" int mit_ocr_prep(const uint8_t *lines_dev, float *out_dev, int N, int H, int Wp, void *stream) {
    if (!lines_dev || !out_dev) return mit_set_error("mit_ocr_prep: null pointer");
    const int64_t npix = (int64_t)N * H * Wp;
    hipLaunchKernelGGL(ocr_prep_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, lines_dev,
                       reinterpret_cast<float4 *>(out_dev), npix);
    MIT_CHECK_LAUNCH("mit_ocr_prep");
    return 0;
}

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

extern "C" int mit_attention_heads(const float *q_dev, int64_t q_rs, int64_t q_ts, const float *k_dev, int64_t k_rs, int64_t k_ts,
                                   const float *v_dev, int64_t v_rs, int64_t v_ts, float *out_dev, int64_t o_rs, int64_t o_ts,
                                   const int *klen_dev, int R, int Tq, int Tk, int kv_div, int heads, int head_dim, void *stream) {
    if (!q_dev || !k_dev || !v_dev || !out_dev) return mit_set_error("mit_attention_heads: null pointer");
    if (R <= 0 || Tq <= 0 || Tk <= 0 || kv_div <= 0) return mit_set_error("mit_attention_heads: empty problem");
    if (Tk > 8192 || R > 65535 || Tq > 65535) return mit_set_error("mit_attention_heads: problem too large");
    if (heads <= 0 || heads > 64 || head_dim <= 0 || (head_dim & 3) || head_dim > 256)
        return mit_set_error("mit_attention_heads: heads in [1, 64], head_dim a multiple of 4 up to 256");
    ocrk_attention(q_dev, q_rs, q_ts, k_dev, k_rs, k_ts, v_dev, v_rs, v_ts, out_dev, o_rs, o_ts, klen_dev, R, Tq, Tk, kv_div,
                   (hipStream_t)stream, heads, head_dim);
    MIT_CHECK_LAUNCH("mit_attention_heads");
    return 0;
}

extern "C" int mit_avgpool_nhwc(const float *in_dev, float *out_dev, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                                int ph, int pw, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_avgpool_nhwc: null pointer");
    if ((C & 3) || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || 2 * ph > kh || 2 * pw > kw)
        return mit_set_error("mit_avgpool_nhwc: bad geometry (C %% 4 == 0, pad <= kernel / 2)");
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    if (B <= 0 || Ho <= 0 || Wo <= 0) return mit_set_error("mit_avgpool_nhwc: empty output");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(avgpool_general_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, B, H, W,
                       C / 4, Ho, Wo, kh, kw, sh, sw, ph, pw);
    MIT_CHECK_LAUNCH("mit_avgpool_nhwc");
    return 0;
}

extern "C" int mit_affine_act_nhwc(const float *in_dev, int64_t in_pixstride, const float *scale_dev, const float *bias_dev,
                                   float *out_dev, int64_t out_pixstride, int64_t npix, int C, int relu, void *stream) {
    if (!in_dev || !scale_dev || !bias_dev || !out_dev) return mit_set_error("mit_affine_act_nhwc: null pointer");
    if ((C & 3) || (in_pixstride & 3) || (out_pixstride & 3)) return mit_set_error("mit_affine_act_nhwc: C and pixel strides must be multiples of 4");
    if (npix <= 0) return 0;
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(npix * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, in_dev, in_pixstride,
                       scale_dev, bias_dev, out_dev, out_pixstride, npix, C / 4, relu);
    MIT_CHECK_LAUNCH("mit_affine_act_nhwc");
    return 0;
}

extern "C" int mit_logsoftmax_top5(const float *logits_dev, int64_t ld, int R, int D, int suppress_tok, float *vals_dev, int *idx_dev,
                                   void *stream) {
    if (!logits_dev || !vals_dev || !idx_dev) return mit_set_error("mit_logsoftmax_top5: null pointer");
    if (R <= 0 || D < 5) return mit_set_error("mit_logsoftmax_top5: need R > 0 and D >= 5");
    ocrk_logsoftmax_top5(logits_dev, ld, R, D, suppress_tok, vals_dev, idx_dev, nullptr, (hipStream_t)stream);
    MIT_CHECK_LAUNCH("mit_logsoftmax_top5");
    return 0;
}

extern "C" int mit_u8_to_f32_nhwc4(const uint8_t *in_dev, float *out_dev, int64_t npix, int mode, void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_u8_to_f32_nhwc4: null pointer");
    if (mode < 0 || mode > 2) return mit_set_error("mit_u8_to_f32_nhwc4: bad mode %d", mode);
    if (npix <= 0) return 0;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, in_dev,
                       reinterpret_cast<float4 *>(out_dev), npix, mode);
    MIT_CHECK_LAUNCH("mit_u8_to_f32_nhwc4");
    return 0;
}

extern "C" int mit_sigmoid_inplace(float *x_dev, int64_t n, void *stream) {
    if (!x_dev) return mit_set_error("mit_sigmoid_inplace: null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x_dev, n);
    MIT_CHECK_LAUNCH("mit_sigmoid_inplace");
    return 0;
}

extern "C" int mit_gelu_inplace(float *x_dev, int64_t n, void *stream) {
    if (!x_dev) return mit_set_error("mit_gelu_inplace: null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gelu_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x_dev, n);
    MIT_CHECK_LAUNCH("mit_gelu_inplace");
    return 0;
}

extern "C" int mit_layernorm(const float *in_dev, int64_t in_rowstride, const float *w_dev, const float *b_dev,
                             float *out_dev, int64_t out_rowstride, int rows, int D, float eps, void *stream) {
    if (!in_dev || !w_dev || !b_dev || !out_dev) return mit_set_error("mit_layernorm: null pointer");
    if (D <= 0 || D > 512) return mit_set_error("mit_layernorm: D out of range");
    if (rows <= 0) return 0;
    if (ocrk_layernorm(in_dev, in_rowstride, w_dev, b_dev, out_dev, out_rowstride, rows, D, eps, (hipStream_t)stream)) return 1;
    MIT_CHECK_LAUNCH("mit_layernorm");
    return 0;
}

static int check_tables(const MitXposTables *tb, int imax_needed, int pmin, int pmax_needed, const char *who) {
    if (!tb || !tb->cos_t || !tb->sin_t || !tb->scale_t || !tb->iscale_t) return mit_set_error("%s: null XPOS tables", who);
    if (imax_needed > tb->imax || pmin < -tb->pmax || pmax_needed >= tb->pmax)
        return mit_set_error("%s: XPOS position out of table range", who);
    return 0;
}

extern "C" int mit_xpos_rotate(const float *in_dev, int64_t in_rs, int64_t in_ts, float *out_dev, int64_t out_rs,
                               int64_t out_ts, int R, int T, int i0, int p0, int downscale, const MitXposTables *tables,
                               void *stream) {
    if (!in_dev || !out_dev) return mit_set_error("mit_xpos_rotate: null pointer");
    if (R <= 0 || T <= 0) return 0;
    if (check_tables(tables, i0 + T, p0, p0 + T - 1, "mit_xpos_rotate")) return 1;
    ocrk_xpos_rotate(in_dev, in_rs, in_ts, out_dev, out_rs, out_ts, R, T, i0, p0, downscale, *tables, (hipStream_t)stream);
    MIT_CHECK_LAUNCH("mit_xpos_rotate");
    return 0;
}

static size_t attention_lines_lds_bytes(int Lmax, int head_dim) {
    return ((size_t)Lmax * (head_dim + 4) + (size_t)(ATTR_THREADS / 64) * ATTR_GQ * (head_dim + Lmax)) * sizeof(float);
}
static constexpr size_t ATTENTION_LINES_LDS_MAX = 150 * 1024;

extern "C" int mit_attention_lines_xpos_max_len(int head_dim) {
    if (head_dim <= 0 || head_dim > 128 || (head_dim & 7)) return 0;
    int L = 0;  // the LDS need is linear in Lmax: solve, then step down past any rounding
    const size_t per_l = ((size_t)(head_dim + 4) + (size_t)(ATTR_THREADS / 64) * ATTR_GQ) * sizeof(float);
    const size_t fixed = (size_t)(ATTR_THREADS / 64) * ATTR_GQ * head_dim * sizeof(float);
    if (ATTENTION_LINES_LDS_MAX > fixed) L = (int)((ATTENTION_LINES_LDS_MAX - fixed) / per_l);
    while (L > 0 && attention_lines_lds_bytes(L, head_dim) > ATTENTION_LINES_LDS_MAX) --L;
    return L > 4096 ? 4096 : L;
}

extern "C" int mit_attention_lines_xpos(const float *q_dev, const float *k_dev, const float *v_dev, float *out_dev, int64_t row_stride,
                                        const int32_t *lines_dev, const int *klen_dev, int n_lines, int Lmax, int heads, int head_dim,
                                        const MitXposTables *tables, void *stream) {
    if (!q_dev || !k_dev || !v_dev || !out_dev || !lines_dev) return mit_set_error("mit_attention_lines_xpos: null pointer");
    if (n_lines <= 0) return 0;
    if (n_lines > 65535 || Lmax <= 0 || Lmax > 4096 || heads <= 0 || heads > 64 || head_dim <= 0 || head_dim > 128 || (head_dim & 7) || (row_stride & 3))
        return mit_set_error("mit_attention_lines_xpos: bad sizes (n_lines %d, Lmax %d, heads %d, head_dim %d)", n_lines, Lmax, heads, head_dim);
    if (check_tables(tables, Lmax, -((Lmax + 1) / 2), Lmax, "mit_attention_lines_xpos")) return 1;
    const size_t sm = attention_lines_lds_bytes(Lmax, head_dim);
    if (sm > ATTENTION_LINES_LDS_MAX)
        return mit_set_error("mit_attention_lines_xpos: lines of %d positions do not fit the LDS form (at most %d at head_dim %d: "
                             "mit_attention_lines_xpos_max_len)", Lmax, mit_attention_lines_xpos_max_len(head_dim), head_dim);
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(attention_lines_xpos_kernel), sm);
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: q, k, v read once and o written once per position (upper bound: every line counted at Lmax)
    MitProbeScope probe("attention_lines_xpos_kernel", s, 16.0 * (double)n_lines * Lmax * heads * head_dim, 4.0 * (double)n_lines * Lmax * heads * Lmax * head_dim);
    const int qblocks = (Lmax + (ATTR_THREADS / 64) * ATTR_GQ - 1) / ((ATTR_THREADS / 64) * ATTR_GQ);
    hipLaunchKernelGGL(attention_lines_xpos_kernel, dim3(heads, n_lines, qblocks), dim3(ATTR_THREADS), sm, s, q_dev, k_dev, v_dev, out_dev, row_stride,
                       reinterpret_cast<const OcrLine *>(lines_dev), klen_dev, Lmax, head_dim, tables->cos_t, tables->sin_t, tables->scale_t,
                       tables->iscale_t, tables->pmax);
    MIT_CHECK_LAUNCH("mit_attention_lines_xpos");
    return 0;
}

extern "C" int mit_memory_kv_lines(const float *k_dev, const float *v_dev, int64_t row_stride, float *mem_k_dev, float *mem_v_dev,
                                   int64_t line_stride, const int32_t *lines_dev, int n_lines, int first_line, int64_t n_rows, int Lmax,
                                   int head_dim, const MitXposTables *tables, void *stream) {
    if (!k_dev || !v_dev || !mem_k_dev || !mem_v_dev || !lines_dev) return mit_set_error("mit_memory_kv_lines: null pointer");
    if (n_lines <= 0 || n_rows <= 0) return 0;
    if ((row_stride & 1) || (line_stride & 1) || head_dim <= 0 || (head_dim & 1) || (row_stride % head_dim)) return mit_set_error("mit_memory_kv_lines: bad strides");
    if (check_tables(tables, Lmax, -((Lmax + 1) / 2), Lmax, "mit_memory_kv_lines")) return 1;
    const int pairs = (int)(row_stride / 2);
    const int64_t total = n_rows * pairs;
    hipStream_t s = (hipStream_t)stream;
    MitProbeScope probe("memory_kv_lines_kernel", s, 16.0 * (double)n_rows * row_stride);
    hipLaunchKernelGGL(memory_kv_lines_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, k_dev, v_dev, row_stride, mem_k_dev, mem_v_dev, line_stride,
                       reinterpret_cast<const OcrLine *>(lines_dev), n_lines, first_line, head_dim / 2, pairs, tables->cos_t, tables->sin_t,
                       tables->iscale_t, tables->pmax, total);
    MIT_CHECK_LAUNCH("mit_memory_kv_lines");
    return 0;
}

extern "C" int mit_attention(const float *q_dev, int64_t q_rs, int64_t q_ts, const float *k_dev, int64_t k_rs, int64_t k_ts,
                             const float *v_dev, int64_t v_rs, int64_t v_ts, float *out_dev, int64_t o_rs, int64_t o_ts,
                             const int *klen_dev, int R, int Tq, int Tk, int kv_div, void *stream) {
    if (!q_dev || !k_dev || !v_dev || !out_dev) return mit_set_error("mit_attention: null pointer");
    if (R <= 0 || Tq <= 0 || Tk <= 0 || kv_div <= 0) return mit_set_error("mit_attention: empty problem");
    if (Tk > 8192 || R > 65535 || Tq > 65535) return mit_set_error("mit_attention: problem too large");
    ocrk_attention(q_dev, q_rs, q_ts, k_dev, k_rs, k_ts, v_dev, v_rs, v_ts, out_dev, o_rs, o_ts, klen_dev, R, Tq, Tk, kv_div,
                   (hipStream_t)stream);
    MIT_CHECK_LAUNCH("mit_attention");
    return 0;
}
