// conv_small_cout.hip — k x k stride-1 convolution with <= 4 output channels (LaMa's 7x7 64->3 output conv).
//
// An implicit-GEMM tile would waste its N dimension on such a layer (32 MFMA columns for 3 channels: 598 GFLOP
// executed per 2048x1456 page for 56 GFLOP of work, 7.4 ms).  Here every thread owns one output pixel and all
// (<= 4) output channels on the fp32 VALU: the input halo tile is staged through LDS in 16-channel slices laid out
// [c4][y][x] so that the 32 threads of a row read consecutive float4 (conflict-free ds_read_b128), and the weights
// are wave-uniform, i.e. scalar loads feeding v_fma_f32 from SGPRs.  Accumulation order: channel slice, tap (ky, kx),
// channel — a single fmaf chain per output, like the MFMA kernel's (k order differs: slice-major instead of tap-major).
//
// Cout <= 3 (the LaMa layer) takes conv_small_cout3_kernel: packed fp32 math (v_pk_fma_f32, two FMAs per lane per issue — the plain
// kernel sat at 85 % of the unpacked VALU peak) on 2 x 2 output pixels per thread (columns x, x + 1; rows y, y + 8).  The two halves of
// every packed operand are two ADJACENT INPUT CHANNELS: an accumulator pair is (even-channel sum, odd-channel sum) of one output, a
// ds_read_b128 of a pixel's 4-channel slice yields two ready-made input pairs, the matching weight pairs come from one scalar load of
// the channel-fastest table w_pairs — every operand is a naturally aligned register pair, so no v_pk_fma_f32 carries an op_sel / neg
// modifier (the earlier row-pair form broadcast each weight with op_sel: 1440 such instructions, the form that misbehaves beside
// MFMA co-tenants; tests/test_build_flags.py).  The columns share a sliding window of K + 1 reads per kernel row and
// tile row.  Even / odd pixels of a tile row sit in separate halves of the LDS row so that the 64 lanes of a wave (stride 2 pixels)
// read consecutive 16-byte slots.  Each thread stages a 16-channel group (whole 64-byte pieces of its tile cells) in registers and
// feeds four 4-channel LDS slices from it, so a 128-byte input line is touched by 4 groups, not by 16 slices.
// Accumulation order per half: 4-channel slice, tap (ky, kx), channel pair; out = (even half + odd half) + bias.
//
// Reference op: FFCResNetGenerator.model[-2:] = ReflectionPad2d(3) + Conv2d(64, 3, 7) + sigmoid
// (manga_translator/inpainting/inpainting_lama_mpe.py:597-600).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mit_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TW = 32, TH = 8, CCH = 16;

__device__ __forceinline__ float act_fn(float v, int act, float alpha) {
    switch (act) {
        case MIT_ACT_RELU: return v > 0.f ? v : 0.f;
        case MIT_ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case MIT_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

template <int K>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const float *__restrict__ in, int64_t in_pix, const f32x4 *__restrict__ w4,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               int64_t out_pix, int H, int W, int Cin, int Cout, int reflect,
                                                               int act, float alpha) {
    constexpr int R = K / 2;
    constexpr int HW_ = TW + 2 * R, HH_ = TH + 2 * R;
    __shared__ f32x4 tile[CCH / 4][HH_][HW_];
    const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const float *ib = in + (int64_t)b * H * W * in_pix;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < Cin; c0 += CCH) {
        __syncthreads();
        for (int i = threadIdx.x; i < (CCH / 4) * HH_ * HW_; i += 256) {
            const int q = i % (CCH / 4);
            const int p = i / (CCH / 4);
            const int px = p % HW_, py = p / HW_;
            int yy = y0 + py - R, xx = x0 + px - R;
            bool ok = true;
            if (reflect) {
                yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
                xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
                ok = yy >= 0 && yy < H && xx >= 0 && xx < W;  // far outside the image (tile overhang): unused
            } else {
                ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4 *>(ib + ((int64_t)yy * W + xx) * in_pix + c0 + q * 4);
            tile[q][py][px] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 *wt = w4 + (int64_t)(ky * K + kx) * Cin + c0;  // wave-uniform -> scalar loads (staging them in LDS measured 3 % slower)
#pragma unroll
                for (int q = 0; q < CCH / 4; ++q) {
                    const f32x4 v = tile[q][ty + ky][tx + kx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 ww = wt[q * 4 + e];
                        acc.x = fmaf(v[e], ww.x, acc.x);
                        acc.y = fmaf(v[e], ww.y, acc.y);
                        acc.z = fmaf(v[e], ww.z, acc.z);
                        acc.w = fmaf(v[e], ww.w, acc.w);  // kept for Cout = 3 as well: dropping the chain measured 1.5 % SLOWER (A/B in one call)
                    }
                }
            }
        }
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
        float *o = out + (((int64_t)b * H + y) * W + x) * out_pix;
        for (int n = 0; n < Cout; ++n) o[n] = act_fn(acc[n] + (bias ? bias[n] : 0.f), act, alpha);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TW3 = 64, TH3 = 16;

template <int K>
__global__ __launch_bounds__(256) void conv_small_cout3_kernel(const float *__restrict__ in, int64_t in_pix, int64_t in_plane /* floats between 16-channel planes; 0: channels interleaved per pixel */,
                                                                const f32x2 *__restrict__ wq /* [taps][Cin / 4][4 out (3 used)][2 channel pairs] */,
                                                                const float *__restrict__ bias, float *__restrict__ out,
                                                                int64_t out_pix, int H, int W, int Cin, int Cout, int reflect,
                                                                int act, float alpha) {
    // Packed operands are CHANNEL pairs: an accumulator pair holds (sum over even channels, sum over odd channels) of one output, a
    // pixel's channels (c, c + 1) are adjacent in the NHWC input and so are a kernel tap's weights for them in wq — every operand of
    // every v_pk_fma_f32 is a naturally aligned register pair (VGPR pair from one ds_read_b128, SGPR pair from one scalar load), no
    // broadcast, no modifier.  The two halves are added once at the end.
    constexpr int R = K / 2;
    constexpr int TR = TH3 + 2 * R;      // tile rows incl. halo
    constexpr int HW_ = TW3 + 2 * R;     // even for odd K
    constexpr int HALF = HW_ / 2;
    __shared__ f32x4 tile[TR][HW_];      // [row][even pixels | odd pixels] = the 4 channels of the current slice
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * TW3, y0 = blockIdx.y * TH3, b = blockIdx.z;
    const float *ib = in + (int64_t)b * H * W * in_pix;
    f32x2 acc[2][2][3];                  // [row r: y, y + 8][column j: x, x + 1][output channel]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[r][j][n] = f32x2{0.f, 0.f};
    constexpr int ITEMS = (TR * HW_ + 255) / 256;  // pixel cells of the tile per thread
    for (int g0 = 0; g0 < Cin; g0 += 16) {
        // stage a 16-channel group of every cell in registers: whole 64-byte pieces per pixel, so each 128-byte line of the NHWC input
        // is touched by 4 such groups instead of by 16 four-channel slices
        f32x4 rg[ITEMS][4];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int i = threadIdx.x + it * 256;
            const int ly = i / HW_, lx = i - ly * HW_;
            int yy = y0 + ly - R, xx = x0 + lx - R;
            if (reflect) {
                yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
                xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
            }
            const bool ok = i < TR * HW_ && xx >= 0 && xx < W && yy >= 0 && yy < H;  // still outside after one reflection: tile overhang, never used
            // planar input: the 16-channel group g0 / 16 is a plane of its own whose pixels are 64 contiguous bytes each — the group's
            // loads are whole 128-byte lines shared by two neighbouring pixels; interleaved input: 64 bytes of every 256-byte pixel
            const float *pa = ib + ((int64_t)yy * W + xx) * in_pix + (in_plane ? (int64_t)(g0 >> 4) * in_plane : (int64_t)g0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rg[it][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) rg[it][q] = *reinterpret_cast<const f32x4 *>(pa + q * 4);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s4 = (g0 >> 2) + q;  // 4-channel slice index
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int i = threadIdx.x + it * 256;
                if (i < TR * HW_) {
                    const int ly = i / HW_, lx = i - ly * HW_;
                    tile[ly][(lx >> 1) + (lx & 1) * HALF] = rg[it][q];
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                f32x4 win[2][K + 1];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i <= K; ++i) win[r][i] = tile[ty + r * (TH3 / 2) + ky][tx + (i >> 1) + (i & 1) * HALF];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x2 *wt = wq + ((int64_t)(ky * K + kx) * (Cin >> 2) + s4) * 8;  // wave-uniform -> one scalar load of 16 floats
                    const f32x2 w00 = wt[0], w01 = wt[1], w10 = wt[2], w11 = wt[3], w20 = wt[4], w21 = wt[5];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 v = win[r][kx + j];
                            const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
                            acc[r][j][0] = __builtin_elementwise_fma(lo, w00, acc[r][j][0]);
                            acc[r][j][1] = __builtin_elementwise_fma(lo, w10, acc[r][j][1]);
                            acc[r][j][2] = __builtin_elementwise_fma(lo, w20, acc[r][j][2]);
                            acc[r][j][0] = __builtin_elementwise_fma(hi, w01, acc[r][j][0]);
                            acc[r][j][1] = __builtin_elementwise_fma(hi, w11, acc[r][j][1]);
                            acc[r][j][2] = __builtin_elementwise_fma(hi, w21, acc[r][j][2]);
                        }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = x0 + 2 * tx + j;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + ty + r * (TH3 / 2);
            if (x < W && y < H) {
                float *o = out + (((int64_t)b * H + y) * W + x) * out_pix;
                for (int n = 0; n < Cout; ++n) o[n] = act_fn((acc[r][j][n].x + acc[r][j][n].y) + (bias ? bias[n] : 0.f), act, alpha);
            }
        }
    }
}

// ---- the same packed arithmetic with the halo tile brought in by the LDS-DMA (round 6) ----
// Input as 4-CHANNEL planes [Cin / 4][B][H][W][4] (the producer's column-split output map, nsplit = 4): a slice's tile row is one
// contiguous run of 16-byte pixels, so the whole tile of a slice is 25 global_load_lds_dwordx4 pieces (per-lane source addresses carry
// the reflection and the even | odd split of the LDS row) — no staging registers (the register-staged kernel holds 112 for a
// 16-channel group), 28 KB of LDS per workgroup, four workgroups per CU whose DMA waits and compute phases cover each other.
// Same accumulation order as conv_small_cout3_kernel: bit-identical results.  Reflect padding only (an out-of-image cell has no source).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

template <int K>
__global__ __launch_bounds__(256, 4) void conv_small_cout3_dma_kernel(const float *__restrict__ in, int64_t in_plane /* floats between 4-channel planes */,
                                                                      const f32x2 *__restrict__ wq, const float *__restrict__ bias, float *__restrict__ out,
                                                                      int64_t out_pix, int H, int W, int Cin, int Cout, int act, float alpha, int parity_major) {
    constexpr int R = K / 2;
    constexpr int TR = TH3 + 2 * R, HW_ = TW3 + 2 * R, HALF = HW_ / 2;
    constexpr int CELLS = TR * HW_, PIECES = (CELLS + 63) / 64, PPW = (PIECES + 3) / 4, BUF_CELLS = PPW * 4 * 64;
    __shared__ f32x4 tile[BUF_CELLS];  // [TR][even pixels | odd pixels] of the current 4-channel slice (+ the surplus lanes' pad cells)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * TW3, y0 = blockIdx.y * TH3, b = blockIdx.z;
    int src_off[PPW];  // this lane's source pixel of each of its wave's pieces (floats into a plane's image of batch entry b)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        int c = (wave * PPW + j) * 64 + lane;
        c = c < CELLS ? c : CELLS - 1;  // surplus lanes copy a duplicate into the pad cells
        const int ly = c / HW_, slot = c - ly * HW_;
        const int lx = slot < HALF ? 2 * slot : 2 * (slot - HALF) + 1;
        int yy = y0 + ly - R, xx = x0 + lx - R;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
        xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);  // still outside after one reflection: tile overhang, never used
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        // parity-major planes: [2 (y parity)][2 (x parity)][H / 2][W / 2][4] — the even | odd halves of an LDS row then read consecutive pixels
        src_off[j] = parity_major ? ((((yy & 1) * 2 + (xx & 1)) * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * 4 : (yy * W + xx) * 4;
    }
    f32x2 acc[2][2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[r][j][n] = f32x2{0.f, 0.f};
    const float *ib = in + (int64_t)b * H * W * 4;
    for (int s4 = 0; s4 < (Cin >> 2); ++s4) {
        const float *pl = ib + (int64_t)s4 * in_plane;
        __syncthreads();  // the previous slice has been consumed
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(pl + src_off[j]), (lds_void_t *)(tile + (wave * PPW + j) * 64), 16, 0, 0);
        __syncthreads();  // (its fence waits for the DMA: vmcnt(0))
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            f32x4 win[2][K + 1];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i <= K; ++i) win[r][i] = tile[(ty + r * (TH3 / 2) + ky) * HW_ + tx + (i >> 1) + (i & 1) * HALF];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x2 *wt = wq + ((int64_t)(ky * K + kx) * (Cin >> 2) + s4) * 8;
                const f32x2 w00 = wt[0], w01 = wt[1], w10 = wt[2], w11 = wt[3], w20 = wt[4], w21 = wt[5];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 v = win[r][kx + j];
                        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
                        acc[r][j][0] = __builtin_elementwise_fma(lo, w00, acc[r][j][0]);
                        acc[r][j][1] = __builtin_elementwise_fma(lo, w10, acc[r][j][1]);
                        acc[r][j][2] = __builtin_elementwise_fma(lo, w20, acc[r][j][2]);
                        acc[r][j][0] = __builtin_elementwise_fma(hi, w01, acc[r][j][0]);
                        acc[r][j][1] = __builtin_elementwise_fma(hi, w11, acc[r][j][1]);
                        acc[r][j][2] = __builtin_elementwise_fma(hi, w21, acc[r][j][2]);
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = x0 + 2 * tx + j;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + ty + r * (TH3 / 2);
            if (x < W && y < H) {
                float *o = out + (((int64_t)b * H + y) * W + x) * out_pix;
                for (int n = 0; n < Cout; ++n) o[n] = act_fn((acc[r][j][n].x + acc[r][j][n].y) + (bias ? bias[n] : 0.f), act, alpha);
            }
        }
    }
}

}  // namespace

extern "C" int mit_conv_small_cout(const float *in_dev, int64_t in_pixstride, int64_t in_planestride, const float *w4_dev, const float *w_pairs_dev, const float *bias_dev,
                                   float *out_dev, int64_t out_pixstride, int B, int H, int W, int Cin, int Cout, int k,
                                   int pad_mode, int act, float act_alpha, void *stream) {
    if (!in_dev || !w4_dev || !out_dev) return mit_set_error("mit_conv_small_cout: null pointer");
    if (Cout < 1 || Cout > 4) return mit_set_error("mit_conv_small_cout: 1 <= Cout <= 4 required (got %d)", Cout);
    if (Cin <= 0 || (Cin % CCH)) return mit_set_error("mit_conv_small_cout: Cin must be a multiple of %d (got %d)", CCH, Cin);
    if (B <= 0 || H <= 0 || W <= 0 || B > 65535) return mit_set_error("mit_conv_small_cout: bad size");
    if ((in_pixstride & 3) || (reinterpret_cast<uintptr_t>(in_dev) & 15) || (reinterpret_cast<uintptr_t>(w4_dev) & 15) || (reinterpret_cast<uintptr_t>(w_pairs_dev) & 31))
        return mit_set_error("mit_conv_small_cout: input pixels and weights must be 16-byte aligned");
    if (pad_mode == MIT_PAD_REFLECT && (k / 2 >= H || k / 2 >= W)) return mit_set_error("mit_conv_small_cout: reflect pad larger than input");
    const int parity_major = in_planestride < 0;                    // (a negative plane stride marks the parity-major form of 4-channel planes)
    if (parity_major) in_planestride = -in_planestride;
    const bool planes4 = in_planestride != 0 && in_pixstride == 4;  // 4-channel planes: the LDS-DMA kernel
    if (parity_major && (!planes4 || (H & 1) || (W & 1))) return mit_set_error("mit_conv_small_cout: parity-major planes need 4-channel planes of even height and width");
    if (in_planestride && ((in_planestride & 3) || (in_pixstride < 16 && !planes4) || !(Cout <= 3 && w_pairs_dev)))
        return mit_set_error("mit_conv_small_cout: planar input (16- or 4-channel planes) needs the packed kernel (Cout <= 3 with w_pairs), pixel stride 4 or >= 16 and a plane stride %% 4 == 0");
    if (planes4 && (pad_mode != MIT_PAD_REFLECT || (int64_t)H * W * 4 > 0x7fffffffLL))
        return mit_set_error("mit_conv_small_cout: 4-channel planes need reflect padding and H * W * 4 < 2^31");
    dim3 grid(mit_div_up(W, TW), mit_div_up(H, TH), B), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(w4_dev);
    const int refl = pad_mode == MIT_PAD_REFLECT;
    // VALU-bound: algorithmic FLOPs 2 k^2 Cin Cout per pixel; bytes: input read once + Cout outputs written
    MitProbeScope probe(Cout <= 3 && w_pairs_dev && !getenv("MIT_SMALL_COUT_PLAIN") ? (k == 7 ? "conv_small_cout3_kernel<7>" : k == 5 ? "conv_small_cout3_kernel<5>" : "conv_small_cout3_kernel<3>")
                                                                    : (k == 7 ? "conv_small_cout_kernel<7>" : k == 5 ? "conv_small_cout_kernel<5>" : "conv_small_cout_kernel<3>"), s, 4.0 * (double)B * H * W * (Cin + Cout), 2.0 * k * k * (double)Cin * Cout * (double)B * H * W);
    // Cout <= 3: the packed-FMA kernel (2.7x fewer VALU instructions).  With 4-channel slices loaded straight from HBM it fetched
    // every 128-byte input line 8 times and was slower than the plain kernel (25 vs 17.7 ms per 16 pages); staging 16-channel groups
    // in registers brought it to 15.9 ms (same-box A/B).  MIT_SMALL_COUT_PLAIN=1 selects the plain kernel for comparison.
    static const bool use_pk = getenv("MIT_SMALL_COUT_PLAIN") == nullptr;
    if (in_planestride && !use_pk) return mit_set_error("mit_conv_small_cout: MIT_SMALL_COUT_PLAIN cannot read planar input");
    if (Cout <= 3 && use_pk && w_pairs_dev) {
        dim3 grid3(mit_div_up(W, TW3), mit_div_up(H, TH3), B);
        const f32x2 *wp = reinterpret_cast<const f32x2 *>(w_pairs_dev);
        if (Cin & 3) return mit_set_error("mit_conv_small_cout: Cin %% 4");
        if (planes4) {
            switch (k) {
                case 3: hipLaunchKernelGGL(conv_small_cout3_dma_kernel<3>, grid3, block, 0, s, in_dev, in_planestride, wp, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, act, act_alpha, parity_major); break;
                case 5: hipLaunchKernelGGL(conv_small_cout3_dma_kernel<5>, grid3, block, 0, s, in_dev, in_planestride, wp, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, act, act_alpha, parity_major); break;
                case 7: hipLaunchKernelGGL(conv_small_cout3_dma_kernel<7>, grid3, block, 0, s, in_dev, in_planestride, wp, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, act, act_alpha, parity_major); break;
                default: return mit_set_error("mit_conv_small_cout: k must be 3, 5 or 7 (got %d)", k);
            }
            MIT_CHECK_LAUNCH("mit_conv_small_cout");
            return 0;
        }
        switch (k) {
            case 3: hipLaunchKernelGGL(conv_small_cout3_kernel<3>, grid3, block, 0, s, in_dev, in_pixstride, in_planestride, wp, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, refl, act, act_alpha); break;
            case 5: hipLaunchKernelGGL(conv_small_cout3_kernel<5>, grid3, block, 0, s, in_dev, in_pixstride, in_planestride, wp, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, refl, act, act_alpha); break;
            case 7: hipLaunchKernelGGL(conv_small_cout3_kernel<7>, grid3, block, 0, s, in_dev, in_pixstride, in_planestride, wp, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, refl, act, act_alpha); break;
            default: return mit_set_error("mit_conv_small_cout: k must be 3, 5 or 7 (got %d)", k);
        }
        MIT_CHECK_LAUNCH("mit_conv_small_cout");
        return 0;
    }
    switch (k) {
        case 3: hipLaunchKernelGGL(conv_small_cout_kernel<3>, grid, block, 0, s, in_dev, in_pixstride, w4, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, refl, act, act_alpha); break;
        case 5: hipLaunchKernelGGL(conv_small_cout_kernel<5>, grid, block, 0, s, in_dev, in_pixstride, w4, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, refl, act, act_alpha); break;
        case 7: hipLaunchKernelGGL(conv_small_cout_kernel<7>, grid, block, 0, s, in_dev, in_pixstride, w4, bias_dev, out_dev, out_pixstride, H, W, Cin, Cout, refl, act, act_alpha); break;
        default: return mit_set_error("mit_conv_small_cout: k must be 3, 5 or 7 (got %d)", k);
    }
    MIT_CHECK_LAUNCH("mit_conv_small_cout");
    return 0;
}
