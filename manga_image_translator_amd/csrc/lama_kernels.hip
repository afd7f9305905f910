// lama_kernels.hip — the memory-bound pieces of the LaMa inpainting stage (HBM-bound, fused
// to one read + one write each): uint8 page/mask -> fp32 NHWC network input, the masked
// positional encoding (MPE) index maps and their embedding add, and the final composite.
//
// Reference: manga_translator/inpainting/inpainting_lama_mpe.py
//   _infer :56-118, LamaFourier.__call__ :713-726, load_masked_position_encoding :751-815,
//   MPE.forward :625-632, FFCResNetGenerator.forward :603-613.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr int MPE_S = 256;  // str_size :758

// ---- (1) u8 page + u8 mask -> fp32 [B,H,W,4] = (rgb/255 * (1-m), m),  m = (mask/255 >= 0.5) ----
__global__ void lama_prep_kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask,
                                 float4 *__restrict__ out, int64_t npix) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < npix; i += stride) {
        const float m = ((float)mask[i] / 255.0f >= 0.5f) ? 1.f : 0.f;  // :85-87
        const float k = 1.f - m;
        float4 v;
        v.x = ((float)img[3 * i + 0] / 255.0f) * k;  // :82, :92
        v.y = ((float)img[3 * i + 1] / 255.0f) * k;
        v.z = ((float)img[3 * i + 2] / 255.0f) * k;
        v.w = m;  // torch.cat([img * (1 - mask), mask]) :604
        out[i] = v;
    }
}

// ---- (1b) the same, written as the reflect-padded image the row-packed stem convolution reads: [B, H + 2 pad, Wp, 4] with
// ReflectionPad2d(pad) materialised (inpainting_lama_mpe.py:560) and the columns beyond W + 2 pad zeroed (they meet zero weights,
// but must be finite) ----
__global__ void lama_prep_padded_kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask, float4 *__restrict__ out, int B,
                                        int H, int W, int pad, int Wp) {
    const int Hp = H + 2 * pad;
    const int64_t total = (int64_t)B * Hp * Wp;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int x = (int)(i % Wp);
        const int64_t r = i / Wp;
        const int y = (int)(r % Hp);
        const int64_t b = r / Hp;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (x < W + 2 * pad) {
            int yy = y - pad, xx = x - pad;
            yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
            xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
            const int64_t s = (b * H + yy) * W + xx;
            const float m = ((float)mask[s] / 255.0f >= 0.5f) ? 1.f : 0.f;
            const float k = 1.f - m;
            v.x = ((float)img[3 * s + 0] / 255.0f) * k;
            v.y = ((float)img[3 * s + 1] / 255.0f) * k;
            v.z = ((float)img[3 * s + 2] / 255.0f) * k;
            v.w = m;
        }
        out[i] = v;
    }
}

// ---- (2a) area-resize of the binary mask to 256x256, "> 0" after rounding (:764-765) ----
// Separable weights come from the host (same formula as the oracle restatement of cv2 INTER_AREA):
// for destination index d, taps [start[d], start[d]+cnt[d]) with weights w[d*maxtaps + j].
__global__ void mpe_downsample_kernel(const uint8_t *__restrict__ mask, int H, int W, const int *__restrict__ ys,
                                      const int *__restrict__ yc, const double *__restrict__ yw, int ymax,
                                      const int *__restrict__ xs, const int *__restrict__ xc,
                                      const double *__restrict__ xw, int xmax, uint8_t *__restrict__ hole) {
    const int b = blockIdx.y;
    const int dy = blockIdx.x;
    const uint8_t *mb = mask + (int64_t)b * H * W;
    for (int dx = threadIdx.x; dx < MPE_S; dx += blockDim.x) {
        double acc = 0.0;
        for (int j = 0; j < yc[dy]; ++j) {
            const uint8_t *row = mb + (int64_t)(ys[dy] + j) * W;
            double racc = 0.0;
            for (int i = 0; i < xc[dx]; ++i) {
                const float m = ((float)row[xs[dx] + i] / 255.0f >= 0.5f) ? 255.0 : 0.0;
                racc += xw[dx * xmax + i] * m;
            }
            acc += yw[dy * ymax + j] * racc;
        }
        // saturate_cast<uchar>(round) > 0
        hole[((int64_t)b * MPE_S + dy) * MPE_S + dx] = (floor(acc + 0.5) >= 1.0) ? 1 : 0;
    }
}

// ---- (2b) ring distance + direction bits on the 256x256 grid, one workgroup per page ----
// Equivalent to the reference's iterative cv2.filter2D loop (:775-802): iteration i marks the
// unknown cells that touch the known region (3x3, BORDER_REFLECT_101) with pos = i, and sets
// direction bit d when the d-th 2x2 sub-window holds a known cell *before* the update.
__device__ __forceinline__ int refl101(int v) { return v < 0 ? -v : (v >= MPE_S ? 2 * MPE_S - 2 - v : v); }

__global__ __launch_bounds__(1024) void mpe_rings_kernel(const uint8_t *__restrict__ hole, uint8_t *__restrict__ relpos,
                                                          uint8_t *__restrict__ direct) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *cur = lds;                 // known map (1 = known), MPE_S*MPE_S
    uint8_t *nxt = lds + MPE_S * MPE_S;
    const int b = blockIdx.x;
    const uint8_t *hb = hole + (int64_t)b * MPE_S * MPE_S;
    uint8_t *rp = relpos + (int64_t)b * MPE_S * MPE_S;
    uint8_t *db = direct + (int64_t)b * MPE_S * MPE_S;
    int any_known = 0, unknown = 0;
    for (int i = threadIdx.x; i < MPE_S * MPE_S; i += blockDim.x) {
        const uint8_t k = hb[i] ? 0 : 1;  // mask3 = 1 - mask/255 :768
        cur[i] = k;
        rp[i] = 0;
        db[i] = 0;
        any_known |= k;
        unknown += !k;
    }
    any_known = __syncthreads_or(any_known);
    int remaining = __syncthreads_count(unknown > 0) > 0;
    if (!any_known) return;  // "otherwise it will cause infinity loop" :773-774
    for (int it = 1; it < 4 * MPE_S && remaining; ++it) {
        int still = 0;
        for (int i = threadIdx.x; i < MPE_S * MPE_S; i += blockDim.x) {
            const int y = i / MPE_S, x = i - y * MPE_S;
            uint8_t k = cur[i];
            if (!k) {
                const int ym = refl101(y - 1), yp = refl101(y + 1), xm = refl101(x - 1), xp = refl101(x + 1);
                const uint8_t a00 = cur[ym * MPE_S + xm], a01 = cur[ym * MPE_S + x], a02 = cur[ym * MPE_S + xp];
                const uint8_t a10 = cur[y * MPE_S + xm], a12 = cur[y * MPE_S + xp];
                const uint8_t a20 = cur[yp * MPE_S + xm], a21 = cur[yp * MPE_S + x], a22 = cur[yp * MPE_S + xp];
                if (a00 | a01 | a02 | a10 | a12 | a20 | a21 | a22) {
                    k = 1;
                    rp[i] = (uint8_t)(it > 127 ? 127 : it);  // clip(int(pos/128*128), 0, 127) :805-807
                    uint8_t bits = 0;
                    if (a00 | a01 | a10) bits |= 1;  // d_filter1 :754
                    if (a10 | a20 | a21) bits |= 2;  // d_filter2 :755
                    if (a01 | a02 | a12) bits |= 4;  // d_filter3 :756
                    if (a12 | a21 | a22) bits |= 8;  // d_filter4 :757
                    db[i] = bits;
                } else {
                    still = 1;
                }
            }
            nxt[i] = k;
        }
        remaining = __syncthreads_or(still);
        uint8_t *t = cur;
        cur = nxt;
        nxt = t;
    }
}

// ---- (2c) x_l += alpha5 * emb[rel] + alpha6 * (direct @ Wd), nearest-resized from the 256 grid ----
__global__ void mpe_add_kernel(float *__restrict__ x, const uint8_t *__restrict__ mask, const uint8_t *__restrict__ relpos,
                               const uint8_t *__restrict__ direct, const int *__restrict__ ymap,
                               const int *__restrict__ xmap, const float *__restrict__ emb /*[128][64]*/,
                               const float *__restrict__ dw /*[4][64]*/, float alpha5, float alpha6, int B, int H,
                               int W) {
    // one thread = one pixel x 4 channels (float4); 16 threads per pixel
    const int64_t total = (int64_t)B * H * W * 16;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c4 = (int)(i & 15);
        const int64_t pix = i >> 4;
        const int xw = (int)(pix % W);
        const int64_t r = pix / W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const bool in_mask = ((float)mask[pix] / 255.0f >= 0.5f);  // ori_mask != 0 :811,813
        int rel = 0, bits = 0;
        if (in_mask) {
            const int64_t cell = ((int64_t)b * MPE_S + ymap[y]) * MPE_S + xmap[xw];
            rel = relpos[cell];
            bits = direct[cell];
        }
        float4 v = reinterpret_cast<float4 *>(x)[i];
        const float4 e = reinterpret_cast<const float4 *>(emb)[rel * 16 + c4];
        float4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (bits & (1 << k)) {
                const float4 w = reinterpret_cast<const float4 *>(dw)[k * 16 + c4];
                d.x += w.x; d.y += w.y; d.z += w.z; d.w += w.w;
            }
        }
        v.x = (v.x + e.x * alpha5) + d.x * alpha6;  // x_l += rel_pos; x_l += direct :611-612
        v.y = (v.y + e.y * alpha5) + d.y * alpha6;
        v.z = (v.z + e.z * alpha5) + d.z * alpha6;
        v.w = (v.w + e.w * alpha5) + d.w * alpha6;
        reinterpret_cast<float4 *>(x)[i] = v;
    }
}

// ---- (2d) the same lookup as packed table rows for the stem's epilogue: rows[pix] = rel | direction bits << 16 (0 outside the mask) ----
__global__ void mpe_rows_kernel(const uint8_t *__restrict__ mask, const uint8_t *__restrict__ relpos, const uint8_t *__restrict__ direct,
                                const int *__restrict__ ymap, const int *__restrict__ xmap, int32_t *__restrict__ rows, int B, int H, int W) {
    const int64_t total = (int64_t)B * H * W;
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; pix < total; pix += stride) {
        const int xw = (int)(pix % W);
        const int64_t r = pix / W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        int32_t v = 0;
        if ((float)mask[pix] / 255.0f >= 0.5f) {  // ori_mask != 0 :811,813 (the test of mpe_add_kernel)
            const int64_t cell = ((int64_t)b * MPE_S + ymap[y]) * MPE_S + xmap[xw];
            v = (int32_t)relpos[cell] | ((int32_t)direct[cell] << 16);
        }
        rows[pix] = v;
    }
}

// ---- (3) composite: sigmoid output [B,H,W,3] f32 + page + mask -> inpainted page u8 ----
__global__ void lama_post_kernel(const float *__restrict__ pred, int64_t pred_pixstride, const uint8_t *__restrict__ img,
                                 const uint8_t *__restrict__ mask, uint8_t *__restrict__ out, int64_t npix, int composite) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < npix; i += stride) {
        const uint8_t mk = mask[i];
        const float m = ((float)mk / 255.0f >= 0.5f) ? 1.f : 0.f;
        const bool keep_inpainted = !composite || mk >= 127;  // mask_original :59-60 (composite == 0: img_inpainted itself, :111)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint8_t px = img[3 * i + c];
            const float im = ((float)px / 255.0f) * (1.f - m);
            const float p = pred[i * pred_pixstride + c];
            const float v = p * m + (1.f - m) * im;                 // :726
            const uint8_t q = (uint8_t)(int)(v * 255.0f);           // astype(np.uint8): truncation :111
            out[3 * i + c] = keep_inpainted ? q : px;               // :117
        }
    }
}

inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mit_lama_prep(const uint8_t *img_dev, const uint8_t *mask_dev, float *out_dev, int B, int H, int W,
                             void *stream) {
    if (!img_dev || !mask_dev || !out_dev) return mit_set_error("mit_lama_prep: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return mit_set_error("mit_lama_prep: empty page");
    const int64_t npix = (int64_t)B * H * W;
    MitProbeScope probe("lama_prep_kernel", (hipStream_t)stream, (double)npix * (3 + 1 + 16));
    hipLaunchKernelGGL(lama_prep_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, img_dev, mask_dev,
                       reinterpret_cast<float4 *>(out_dev), npix);
    MIT_CHECK_LAUNCH("mit_lama_prep");
    return 0;
}

extern "C" int mit_lama_prep_padded(const uint8_t *img_dev, const uint8_t *mask_dev, float *out_dev, int B, int H, int W, int pad, int Wp,
                                    void *stream) {
    if (!img_dev || !mask_dev || !out_dev) return mit_set_error("mit_lama_prep_padded: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return mit_set_error("mit_lama_prep_padded: empty page");
    if (pad < 0 || pad >= H || pad >= W || Wp < W + 2 * pad) return mit_set_error("mit_lama_prep_padded: bad padding (pad %d, Wp %d for %d x %d)", pad, Wp, H, W);
    const int64_t npix = (int64_t)B * (H + 2 * pad) * Wp;
    MitProbeScope probe("lama_prep_kernel", (hipStream_t)stream, (double)B * H * W * 4 + (double)npix * 16);
    hipLaunchKernelGGL(lama_prep_padded_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, img_dev, mask_dev,
                       reinterpret_cast<float4 *>(out_dev), B, H, W, pad, Wp);
    MIT_CHECK_LAUNCH("mit_lama_prep_padded");
    return 0;
}

extern "C" int mit_lama_mpe_index(const uint8_t *mask_dev, int B, int H, int W, const int *ys, const int *yc,
                                  const double *yw, int ymax, const int *xs, const int *xc, const double *xw, int xmax,
                                  uint8_t *hole_dev, uint8_t *relpos_dev, uint8_t *direct_dev, void *stream) {
    if (!mask_dev || !hole_dev || !relpos_dev || !direct_dev || !ys || !yc || !yw || !xs || !xc || !xw)
        return mit_set_error("mit_lama_mpe_index: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return mit_set_error("mit_lama_mpe_index: empty page");
    hipLaunchKernelGGL(mpe_downsample_kernel, dim3(MPE_S, B), dim3(256), 0, (hipStream_t)stream, mask_dev, H, W, ys, yc,
                       yw, ymax, xs, xc, xw, xmax, hole_dev);
    MIT_CHECK_LAUNCH("mit_lama_mpe_index(downsample)");
    const size_t smem = 2 * MPE_S * MPE_S;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(mpe_rings_kernel), smem);
    hipLaunchKernelGGL(mpe_rings_kernel, dim3(B), dim3(1024), smem, (hipStream_t)stream, hole_dev, relpos_dev, direct_dev);
    MIT_CHECK_LAUNCH("mit_lama_mpe_index(rings)");
    return 0;
}

extern "C" int mit_lama_mpe_add(float *x_dev, const uint8_t *mask_dev, const uint8_t *relpos_dev,
                                const uint8_t *direct_dev, const int *ymap_dev, const int *xmap_dev,
                                const float *emb_dev, const float *dirw_dev, float alpha5, float alpha6, int B, int H,
                                int W, void *stream) {
    if (!x_dev || !mask_dev || !relpos_dev || !direct_dev || !ymap_dev || !xmap_dev || !emb_dev || !dirw_dev)
        return mit_set_error("mit_lama_mpe_add: null pointer");
    const int64_t total = (int64_t)B * H * W * 16;
    // algorithmic bytes: the 64-channel stem output read and written once (+ 1 mask byte per pixel; the 256x256 index maps stay in cache)
    MitProbeScope probe("mpe_add_kernel", (hipStream_t)stream, (double)B * H * W * (2.0 * 64 * 4 + 1));
    hipLaunchKernelGGL(mpe_add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x_dev, mask_dev,
                       relpos_dev, direct_dev, ymap_dev, xmap_dev, emb_dev, dirw_dev, alpha5, alpha6, B, H, W);
    MIT_CHECK_LAUNCH("mit_lama_mpe_add");
    return 0;
}

extern "C" int mit_lama_mpe_rows(const uint8_t *mask_dev, const uint8_t *relpos_dev, const uint8_t *direct_dev, const int *ymap_dev,
                                 const int *xmap_dev, int32_t *rows_dev, int B, int H, int W, void *stream) {
    if (!mask_dev || !relpos_dev || !direct_dev || !ymap_dev || !xmap_dev || !rows_dev) return mit_set_error("mit_lama_mpe_rows: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return mit_set_error("mit_lama_mpe_rows: bad size");
    const int64_t total = (int64_t)B * H * W;
    MitProbeScope probe("mpe_rows_kernel", (hipStream_t)stream, (double)total * 5.0);   // one mask byte read, one row word written
    hipLaunchKernelGGL(mpe_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, mask_dev, relpos_dev, direct_dev,
                       ymap_dev, xmap_dev, rows_dev, B, H, W);
    MIT_CHECK_LAUNCH("mit_lama_mpe_rows");
    return 0;
}

extern "C" int mit_lama_post(const float *pred_dev, int64_t pred_pixstride, const uint8_t *img_dev,
                             const uint8_t *mask_dev, uint8_t *out_dev, int B, int H, int W, int composite, void *stream) {
    if (!pred_dev || !img_dev || !mask_dev || !out_dev) return mit_set_error("mit_lama_post: null pointer");
    const int64_t npix = (int64_t)B * H * W;
    MitProbeScope probe("lama_post_kernel", (hipStream_t)stream, (double)npix * (12 + 3 + 1 + 3));
    hipLaunchKernelGGL(lama_post_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, pred_dev,
                       pred_pixstride, img_dev, mask_dev, out_dev, npix, composite);
    MIT_CHECK_LAUNCH("mit_lama_post");
    return 0;
}
