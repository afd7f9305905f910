/*
 * mit_hip.h — C-ABI of the MI355X (gfx950) dense-stage engine for manga-image-translator.
 *
 * This is the drop-in boundary beneath the reference's Python plugin classes
 * (SURVEY.md §8b).  Nothing native exists in the reference for this path: every FLOP goes
 * through stock ATen ops called from `async _infer()` bodies.  Each entry point below names
 * the reference call site whose device work it replaces; the Python shim that binds it
 * (ctypes) is `manga_image_translator_amd/lib.py`, and INTEGRATION.md shows the stub a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / HIP types in signatures.  `stream` is a
 *     hipStream_t passed as void* (0 = default stream).
 *   - every pointer named *_dev is a DEVICE pointer (e.g. torch.Tensor.data_ptr()).
 *   - all functions return 0 on success, non-zero on error; `mit_last_error()` returns a
 *     thread-local, NUL-terminated description (the Python shim raises RuntimeError with it;
 *     reference errors are Python exceptions: manga_translator.py:469-477,496-502,583-590).
 *   - all kernels are asynchronous on `stream`; nothing synchronises unless documented.
 *   - activations are fp32 NHWC ("pixel-major, channel-contiguous").
 */
#ifndef MIT_HIP_H
#define MIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIT_ABI_VERSION 11
#define MIT_MAX_TAPS 64

/* activation codes for fused epilogues */
enum MitAct {
    MIT_ACT_NONE = 0,
    MIT_ACT_RELU = 1,    /* nn.ReLU            (inpainting_lama_mpe.py:372-399, basemodel.py:20-22) */
    MIT_ACT_LEAKY = 2,   /* nn.LeakyReLU(alpha) (yolov5/common.py:39-40: 0.1; esrgan: 0.2)          */
    MIT_ACT_SILU = 3,    /* nn.SiLU             (yolov5/common.py:37)                                */
    MIT_ACT_SIGMOID = 4, /* nn.Sigmoid          (basemodel.py:52,107,136; lama generator :599-600)   */
    MIT_ACT_GELU = 5     /* nn.GELU (erf)       (model_48px.py:199,534)                              */
};
/* OR-ed into MitConvGemm.act: the ``post`` operand is added BEFORE the activation, act(v*scale + bias + post) — the
 * ``out += identity; relu(out)`` of torchvision's BasicBlock (default detector backbone, default_utils/DBNet_resnet34.py:76). */
#define MIT_ACT_POST_FIRST 0x100

enum MitPad {
    MIT_PAD_ZERO = 0,    /* nn.Conv2d default                                   */
    MIT_PAD_REFLECT = 1  /* padding_mode='reflect' / nn.ReflectionPad2d (inpainting_lama_mpe.py:334-340,554,597) */
};

/* Addressing of a logical [z][nb][oy][ox][n] fp32 tensor, element strides.
 * z = z1 * zdiv + z0.  Column n maps to (n / nsplit) * nhi + (n % nsplit); nsplit == 0
 * means plain contiguous columns.  base == NULL disables the operand. */
typedef struct MitTensorMap {
    float *base;
    int64_t zs1, zs0, bs, ys, xs;
    int64_t nhi;
    int32_t nsplit;
    int32_t _pad;
} MitTensorMap;

/* mit_conv_gemm — the one dense contraction kernel of the engine (implicit-GEMM on
 * v_mfma_f32_32x32x2_f32, exact fp32, k-ordered fmaf chain).
 *
 *   C[z][m][n] = epilogue( sum_{t<ntaps} sum_{ci<Cin} A[z][m, t, ci] * W[z][t*Cin+ci][n] )
 *
 *   m = (nb, oy, ox), nb < NB, oy < Ho, ox < Wo
 *   A[z][m,t,ci] = a[z1*a_zs1 + z0*a_zs0 + nb*a_bs + iy*a_ys + ix*a_xs + tap_off[t] + ci]
 *        iy = oy*sy + tap_dy[t], ix = ox*sx + tap_dx[t]; outside [0,Hi)x[0,Wi): zero or reflect
 *   W[z][k][n]   = w[z1*w_zs1 + z0*w_zs0 + k*ldw + n]          (rows k >= Kw, cols n >= Nw read as 0)
 *   epilogue(v)  = act( (v + pre[..]) * scale[n] + bias[n] ) + post[..]      (MIT_ACT_POST_FIRST: post joins inside act)
 *
 * Replaces (with BatchNorm folded into scale/bias by the packer):
 *   nn.Conv2d / nn.ConvTranspose2d (as stride-parity sub-convolutions) / nn.Linear calls in
 *   inpainting_lama_mpe.py:349-369,286-307,603-613; ctd_utils/basemodel.py:56-72,100-119;
 *   ctd_utils/yolov5/common.py:30-49,94-135; ocr/model_48px.py:203-276,327-394,548-572;
 *   and torch.fft.rfftn / irfftn of FourierUnit (inpainting_lama_mpe.py:228,252) expressed as
 *   dense DFT matrices (W operand = activations, A operand = the DFT matrix).
 *
 * Requirements: Cin % 4 == 0, ldw % 4 == 0, Nw % 4 == 0, all bases 16-byte aligned,
 *   a_xs/a_ys/a_bs/tap_off multiples of 4 (float4 loads).
 */
typedef struct MitConvGemm {
    /* A operand */
    const float *a;
    int64_t a_zs1, a_zs0, a_bs, a_ys, a_xs;
    int32_t NB, Hi, Wi, Cin;
    int32_t Ho, Wo, sy, sx;
    int32_t ntaps, pad_mode;
    int8_t tap_dy[MIT_MAX_TAPS];
    int8_t tap_dx[MIT_MAX_TAPS];
    int32_t tap_off[MIT_MAX_TAPS];
    /* W operand */
    const float *w;
    int64_t w_zs1, w_zs0, ldw;
    int32_t Kw, Nw;
    /* problem */
    int32_t N;      /* output columns actually stored */
    int32_t Z, zdiv;
    /* epilogue */
    MitTensorMap c, pre, post;
    const float *scale, *bias;
    int32_t act;
    float act_alpha;
    /* optional: W pre-split into three bf16 planes (mit_gemm_split_pack) for the split-bf16 tiles; NULL = fp32 MFMA only.
     * ws_zs0 = uint16 elements between z0 slices (the z1 stride must be 0 when this is set). */
    const uint16_t *w_split;
    int64_t ws_zs0;
    /* optional: a device-resident step counter, for launch sequences that are replayed from a hipGraph with the same arguments every
     * time (the beam-search steps of mit_ocr48_decode): when non-NULL the A operand starts a_dyn * (*dyn) floats and the C map
     * c_dyn * (*dyn) floats further than the descriptor says.  NULL everywhere else. */
    const int32_t *dyn;
    int64_t a_dyn, c_dyn;
    /* optional: a two-table row lookup joined AFTER the activation and the post residual — y += lut1[r1][n]; y += lut2[r2][n], in that
     * order — where output row m (= its pixel index, batch-major) carries lut_rows[m] = r1 | r2 << 16 and both tables have lut_ld floats
     * per row.  LaMa's masked position encoding rides on it: the 7x7 stem writes relu(bn(conv)) + alpha5 * emb[rel] + alpha6 * dir in
     * one pass instead of a separate read-modify-write of the 64-channel stem output (inpainting_lama_mpe.py:609-613).  NULL = off.
     * Implemented as its own instantiation of the float4 epilogue of the fast / split tiles (every other launch compiles to the code it
     * had without it): needs Cin % 16 == 0 and <= 16 taps, N % 4 == 0, lut_ld % 4 == 0, 16-byte aligned maps and tables, Z == 1, no
     * post residual, act none or relu — anything else is refused with an error. */
    const int32_t *lut_rows;
    const float *lut1, *lut2;
    int64_t lut_ld;
} MitConvGemm;

const char *mit_last_error(void);
int mit_abi_version(void);
/* sha256 over the sources (every .hip / .h under csrc, this header) and compiler flags the library was built from; the Python
 * binding refuses (or rebuilds) a library whose digest differs from the tree's, so stale kernels are never measured. */
const char *mit_source_digest(void);

/* The mitigation that shipped before the co-tenancy failure was understood: mit_rfft_rows / mit_irfft_rows were
 * seen returning wrong workgroups while ANOTHER queue's kernel (a second process on the GPU, a second stream) that issues MFMAs was
 * resident on the same CU; with this on they read LDS with 4-byte loads and take a whole CU's LDS, so nothing fits beside them (about 3 %
 * of a LaMa forward).  The cause was the SLP vectoriser's packed-fp32 instructions; the library is built without them now and gives
 * the one-stream bytes beside any co-tenant in its DEFAULT launch form (tests/test_cotenant_gpu.py), so this switch is no longer needed —
 * it stays for A/B runs.  Off by default; initial value from MIT_COTENANT_SAFE in the environment.  on < 0 only queries.  Returns the
 * previous value.  Nothing in the reference corresponds to it. */
int mit_cotenant_safe_set(int on);

/* device / runtime ------------------------------------------------------------------- */
int mit_device_count(int *count);
int mit_device_name(int device, char *buf, int buflen);

/* dense contraction -------------------------------------------------------------------- */
int mit_conv_gemm(const MitConvGemm *desc, void *stream);
/* tile configuration override for tuning/tests: cfg = -1 auto, else index into the table
 * reported by mit_conv_gemm_config_name(). */
int mit_conv_gemm_cfg(const MitConvGemm *desc, int cfg, void *stream);
const char *mit_conv_gemm_config_name(int cfg);
/* the kernel's template-id as rocprofv3 prints it (without namespace), e.g. "conv_gemm_fast_kernel<128, 128, 16, 1, 4, 4, 4>":
 * lets bench.py join its per-tile probe numbers with the profiler's kernel-trace / PMC rows; NULL past the table. */
const char *mit_conv_gemm_config_kernel(int cfg);

/* Split-bf16 form of the same contraction (the GEMM mode: mit_gemm_mode_set / MIT_GEMM_SPLIT, or an explicit "split*" tile through
 * mit_conv_gemm_cfg).  gfx950's bf16 MFMA runs at 16x the rate of the fp32 one; an fp32 number is EXACTLY the sum of three bf16
 * numbers (x = hi + mid + lo, each the round-to-nearest bf16 of what the previous ones left), so
 *     a * b = sum over plane pairs (p, q) of a_p * b_q        every such product is exact in fp32 (8 x 8 significant bits)
 * and the contraction becomes 9 bf16 MFMA products accumulated in fp32 ("p9": the error is that of the fp32 accumulation alone, as
 * for the fp32 MFMA chain), or 6 when the three pairs with p + q >= 3 (relative weight <= 2^-24 of the product) are dropped ("p6").
 * Activations are split in the kernel while they are staged to LDS; the constant W operand is split once:
 *   mit_gemm_split_pack: w [nz][Kw][ldw] fp32 (slices w_zs floats apart; Kw % 8 == 0, ldw % 4 == 0)
 *                        -> out [nz][3 planes][Kw / 8][ldw][8] bf16 (16-byte cells of 8 consecutive k of one column),
 *                        nz * 3 * Kw * ldw uint16 elements; pass it as MitConvGemm.w_split with ws_zs0 = 3 * Kw * ldw.
 * Nothing in the reference corresponds to it (the reference computes these layers with fp32 torch kernels). */
int mit_gemm_split_pack(const float *w_dev, int64_t w_zs, int nz, int Kw, int64_t ldw, uint16_t *out_dev, void *stream);
/* The GEMM mode of mit_conv_gemm's automatic tile choice, process-wide, switchable at run time (launches already queued keep the tile
 * they were given):
 *   6 (default)  layers that carry w_split and fill the chip run on the split-bf16 tiles with 6 of the 9 plane products ("p6": the
 *                dropped terms are <= 2^-25 |a b| each, below one fp32 multiply rounding);
 *   9            the same with all 9 plane products (no term dropped: fp32 accumulation order is the only difference to the fp32 MFMA);
 *   0            fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere, w_split ignored.
 * The initial value is MIT_GEMM_SPLIT from the environment (0 | 6 | 9) when set, else 6.  mit_gemm_mode_set returns non-zero for any
 * other value.  Nothing in the reference corresponds to it. */
int mit_gemm_mode_set(int mode);
int mit_gemm_mode_get(void);
/* Smallest launch, counted in 128 x 64 output tiles (x Z), that the automatic choice hands to the split tiles (default 0 — every
 * eligible launch, which keeps a page's result independent of the batch it is part of — or MIT_GEMM_SPLIT_MIN_TILES); n >= 0 sets it,
 * n < 0 only queries.  Returns the previous value.  A tuning knob: 1280 = one full wave of workgroups. */
int64_t mit_gemm_split_min_tiles(int64_t n);

/* ---- planar operands: the plain GEMMs of the split-bf16 mode with activations that ARRIVE split ---------------------------------
 * In mit_conv_gemm's split tiles the activations are split into their three bf16 planes by VALU work inside the K loop, once per
 * output-column tile.  When the producer of an activation tensor writes the planes itself (mit_split_planes, the planar forms of
 * mit_dwconv_nhwc_ragged_rows / mit_wino43_input, or mit_pgemm's own planar epilogue) a plain GEMM (1x1 convolution, nn.Linear, the 36
 * Winograd products) needs no VALU work in its K loop at all: both operands go global -> LDS by the LDS-DMA (global_load_lds_dwordx4)
 * in the cell layout the bf16 MFMA consumes, a ring of stages keeps the loads one or two K-tiles ahead behind counted vmcnt waits, and
 * a workgroup walks several output tiles so that the next tile's loads are in flight during an epilogue.
 *
 * "planes" of a row-major fp32 matrix X[R][K] (K % 8 == 0), ld >= R:   P[3][K / 8][ld][8] bf16 (uint16 bit patterns),
 *     X[r][8 c + j] == bf16(P[0][c][r][j]) + bf16(P[1][c][r][j]) + bf16(P[2][c][r][j])      exactly, each plane the
 *     round-to-nearest-even bf16 of what the previous ones left (the same split as conv_gemm_split_kernel and mit_gemm_split_pack:
 *     the weights' planes [3][Kw / 8][ldw][8] ARE this layout with r = output column).
 * mit_pgemm:  C[z][m][n] = epilogue( sum_k A[z][m][k] * W[z][k][n] ),  the sum taken as NPROD (6 | 9) bf16 MFMA products per 16-wide k
 *     step in the order of the split tiles: results are bit-identical to mit_conv_gemm on a "split*p6*" / "split*p9*" tile for the
 *     same operands.  epilogue(v) = act((v + pre) * scale[n] + bias[n]) + post   (MIT_ACT_POST_FIRST as in MitConvGemm).
 *     Output either fp32 row-major (c, ldc) or planes (c_planes, ld_cp: N % 8 == 0; pre / post must be NULL), or both NULL = error.
 * Replaces the same reference calls as mit_conv_gemm for these layers: the pointwise convolutions of ConvNeXtBlock
 * (ocr/model_48px.py:203-214) and the convl2l / convl2g / convg2l products of the FFC blocks (inpainting_lama_mpe.py:349-369). */
typedef struct MitPGemm {
    const uint16_t *a_planes; /* [Z][3][K/8][lda][8] */
    int64_t a_zs;             /* uint16 elements between z slices of A (0 for Z == 1) */
    int64_t lda;              /* rows per (plane, k-cell) slab, >= M */
    const uint16_t *w_planes; /* [Z][3][K/8][ldw][8] (mit_gemm_split_pack) */
    int64_t w_zs;             /* uint16 elements between z slices of W (0: shared) */
    int64_t ldw;              /* columns per slab, >= N */
    int32_t M, N, K, Z;       /* K % 16 == 0 */
    float *c;                 /* fp32 output [Z][M][ldc] or NULL */
    int64_t ldc, c_zs;
    const float *pre;         /* optional fp32 operands of the epilogue, row-major like c (own strides) */
    int64_t ld_pre, pre_zs;
    const float *post;
    int64_t ld_post, post_zs;
    uint16_t *c_planes;       /* planar output [Z][3][N/8][ld_cp][8] or NULL */
    int64_t ld_cp, cp_zs;
    const float *scale, *bias;
    int32_t act;
    float act_alpha;
    int32_t nprod;            /* 6 | 9 (0 = follow mit_gemm_mode_get(), which must then be 6 or 9) */
    int32_t tile;             /* -1 = automatic; else an index into mit_pgemm_tile_name() (tests / tuning) */
} MitPGemm;
int mit_pgemm(const MitPGemm *desc, void *stream);
const char *mit_pgemm_tile_name(int tile); /* NULL past the table */
/* 1 when mit_pgemm accepts the problem (sizes / alignment); 0 otherwise, with the reason in mit_last_error(). */
int mit_pgemm_supported(const MitPGemm *desc);
/* X fp32 [R][K] (row stride ldx floats, ldx % 4 == 0, K % 8 == 0) -> planes [3][K/8][ld][8] as above.  The stand-alone producer
 * (tests, and tensors whose producer has no planar form). */
int mit_split_planes(const float *x_dev, int64_t ldx, int R, int K, uint16_t *planes_dev, int64_t ld, void *stream);
/* planes -> fp32 (the exact sum hi + mid + lo): tests and debugging. */
int mit_join_planes(const uint16_t *planes_dev, int64_t ld, int R, int K, float *x_dev, int64_t ldx, void *stream);

/* k x k (3, 5, 7) stride-1 "same" convolution with 1..4 output channels on the fp32 VALU (an MFMA tile would idle 29 of
 * its 32 columns): out[b,y,x,n] = act(sum in[b,y+dy,x+dx,c] * w4[(ky*k+kx)*Cin + c][n] + bias[n]).  in: NHWC with pixel
 * stride in_pixstride floats (Cin % 16 == 0 channels used); w4: [k*k][Cin][4] (output channel padded to 4, zeros beyond
 * Cout); out: pixel stride out_pixstride, Cout floats written.  w_pairs (optional, Cout <= 3, 32-byte aligned): the same weights
 * channel-fastest, [k*k][Cin / 4][4 outputs (zeros beyond Cout)][4 channels] — the operand pairs (channels c, c + 1 of one output) of
 * the packed-FMA kernel that Cout <= 3 takes when it is given (2.7x fewer VALU instructions; accumulators hold even / odd channel
 * sums, no packed instruction carries an op_sel / neg modifier); NULL = the plain kernel.  in_planestride != 0 (packed kernel only): the
 * input arrives as Cin / 16 planes of [B,H,W,16] (in_pixstride = 16), in_planestride floats apart — what the producing convolution
 * writes through a column-split output map (MitTensorMap.nsplit = 16): a 16-channel group of a tile is then a run of whole 128-byte
 * lines instead of a quarter of every pixel's 256 bytes.
 * Round 6: in_pixstride == 4 with in_planestride != 0 = planes of FOUR channels [Cin / 4][B][H][W][4]: each 4-channel slice's halo tile
 * goes global -> LDS by global_load_lds_dwordx4 (no staging registers, 4 workgroups per CU; reflect padding only); a NEGATIVE
 * in_planestride (-stride) says every plane stores its images parity-major, [2 (y & 1)][2 (x & 1)][H / 2][W / 2][4] — the layout in
 * which the stride-2 transposed convolution that produces them writes consecutive pixels per launch.  Same bits in every layout.
 * Replaces ReflectionPad2d(3) + Conv2d(64, 3, 7) + sigmoid at the end of FFCResNetGenerator (inpainting_lama_mpe.py:597-600). */
int mit_conv_small_cout(const float *in_dev, int64_t in_pixstride, int64_t in_planestride, const float *w4_dev, const float *w_pairs_dev, const float *bias_dev, float *out_dev,
                        int64_t out_pixstride, int B, int H, int W, int Cin, int Cout, int k, int pad_mode, int act,
                        float act_alpha, void *stream);

/* The ConvNeXt block's pointwise pair as ONE launch (round 5): out = post + scale2 * (W2 . gelu(W1 . x + b1)) + bias2 per row, i.e.
 * pwconv1 -> GELU -> pwconv2 -> gamma -> + input of ConvNeXtBlock.forward (manga_translator/ocr/model_48px.py:203-214), in the
 * split-bf16 p6 arithmetic of mit_conv_gemm's tiles; the 4C-wide hidden activations stay in registers (as two launches the C = 80 stage
 * writes and re-reads [M, 320] fp32).  x [M, C] (row stride ldx floats), w1_planes = mit_gemm_split_pack of pwconv1's packed weight
 * [K = C][ldw = 4C]; w2perm_planes = mit_gemm_split_pack of pwconv2's packed weight [K = 4C][ldn2] AFTER permuting its rows to the order
 * in which the first contraction's accumulator registers hold the hidden index (k' = 32 hb + 16 s + 8 lh + j  <-  hidden
 * 32 hb + (j & 3) + 8 (2 s + (j >> 2)) + 4 lh; manga_image_translator_amd/ocr48.py builds it); b1 [4C]; scale2 / bias2 [C] (gamma,
 * gamma * bias; NULL = 1 / 0); post [M, C] or NULL (may be out itself).  mit_convnext_mlp_supported(C): 1 for the instantiated
 * widths (80).  The hidden activations are bit-identical to the two-launch form's; the second contraction adds the same products with
 * the 16 values of an MFMA step in other k slots (last-bit differences in the fp32 sums, same bound). */
int mit_convnext_mlp_supported(int C);
int mit_convnext_mlp(const float *x_dev, int64_t ldx, int M, int C, const uint16_t *w1_planes_dev, const float *b1_dev,
                     const uint16_t *w2perm_planes_dev, int64_t ldn2, const float *scale2_dev, const float *bias2_dev,
                     const float *post_dev, int64_t ldp, float *out_dev, int64_t ldo, void *stream);

/* ---- Winograd F(4x4, 3x3) for the stride-1 3x3 convolutions of the FFC blocks (inpainting_lama_mpe.py:349-369: convl2l,
 * convg2l, convl2g under ReflectionPad; the reference calls nn.Conv2d = 9 multiplies per output, this form 2.25):
 *   mit_wino43_input : x NHWC [B,H,W,C] (strides in floats) -> V [36][T][C], T = B * ceil(H/4) * ceil(W/4), the B^T d B
 *                      transform of the 6x6 patch (pad 1, MIT_PAD_REFLECT or MIT_PAD_ZERO) of every 4x4 output tile;
 *   the 36 products M_z = V_z [T x C] @ U_z [C x N] (U = G g G^T, transformed once on the host) run on mit_conv_gemm, Z = 36;
 *   mit_wino43_output: M [36][T][N] -> y NHWC = act((A^T m A) * scale[n] + bias[n]) + post, outputs past H / W dropped.
 * fp32 throughout; per-layer error ~1e-5 of the output range against 3e-7 for the direct form (tests/test_winograd_gpu.py). */
int mit_wino43_input(const float *x_dev, int64_t x_bs, int64_t x_ys, int64_t x_xs, float *v_dev, int B, int H, int W, int C,
                     int pad_mode, void *stream);
int mit_wino43_output(const float *m_dev, float *y_dev, int64_t y_bs, int64_t y_ys, int64_t y_xs, const float *post_dev,
                      int64_t p_bs, int64_t p_ys, int64_t p_xs, const float *scale_dev, const float *bias_dev, int B, int H,
                      int W, int N, int act, float alpha, void *stream);

/* kernel-time probe (measurement only; bench.py's roofline leg).  While enabled every mit_conv_gemm launch — from the
 * host or from the native decoder loop — is bracketed by HIP events on its own stream; mit_prof_read synchronises those
 * events and returns, per tile configuration, the launch count, the summed kernel time and the summed FLOPs
 * (exec = 2*M*N*K*Z as launched; alg = the caller's figure given through mit_prof_tag_next for the next launch of
 * this thread, else exec).  Nothing in the reference corresponds to it (the reference has no profiler hooks on this path). */
typedef struct MitProfStat {
    int64_t launches;
    double ms, exec_flops, alg_flops;
} MitProfStat;
int mit_prof_enable(int on); /* clears the records; on != 0 starts recording */
int mit_prof_tag_next(double alg_flops);
int mit_prof_read(MitProfStat *stats, int max_cfgs, int *n_cfgs);
/* One CSV line per recorded launch (tile, M, N, K, taps, Z, act, ms, executed / algorithmic FLOPs): the per-layer view behind
 * bench.py's per-tile totals (scripts/ocr_layers.py).  Call before mit_prof_enable() clears the records. */
int mit_prof_dump(const char *path);
/* The same probe for the kernels that are NOT mit_conv_gemm (the HBM-bound transforms / FFTs / element-wise passes, the VALU
 * output convolution, the OCR attention / softmax kernels): per kernel name, launches, summed time and the summed ALGORITHMIC
 * bytes (inputs read once + outputs written once at the stored precision) and FLOPs of its launches.
 * mit_prof_enable() arms and clears it together with the conv probe. */
typedef struct MitProfKernelStat {
    char name[48];
    int64_t launches;
    double ms, alg_bytes, alg_flops;
} MitProfKernelStat;
int mit_prof_kernels_read(MitProfKernelStat *stats, int max_stats, int *n_stats);

/* 8-bit image resizes around the inpainter, on device bytes [B,H,W,C] -> [B,dh,dw,C] (1 <= C <= 4):
 *   mode 0 = cv2.INTER_LINEAR (8-bit path, 11-bit coefficients)   — the resize to a multiple of 8 and back
 *            (inpainting_lama_mpe.py:77-79,112-113), detection/common.py:79-84
 *   mode 1 = exact 2x shrink = 2x2 box mean (what OpenCV substitutes for both linear flavours at scale 1/2)
 *   mode 2 = cv2.INTER_LINEAR_EXACT (8.8 fixed point)             — resize_keep_aspect (utils/generic.py:251-255, _infer :64-66)
 * Tap tables (modes 0, 2) come from the host (manga_image_translator_amd/imgproc.py): per destination index the first source
 * index (int32) and two uint16 weights {w(idx), w(idx+1)}; idx+1 is clamped to the last row / column by the kernel. */
int mit_resize_u8(const uint8_t *src_dev, int B, int H, int W, int C, uint8_t *dst_dev, int dh, int dw, int mode, const int *yidx_dev,
                  const uint16_t *ycoef_dev, const int *xidx_dev, const uint16_t *xcoef_dev, void *stream);
/* cv2.bilateralFilter on 8-bit RGB pages [B,H,W,3] (mask_refinement/text_mask_utils.py:159 and detection/default.py:64 call it
 * with d = 17, sigmaColor = sigmaSpace = 80): BORDER_REFLECT_101, `ntaps` taps of the circular support in row-major order, tap k at
 * (dy, dx) = (tap_ofs[k] >> 16, (int16)(tap_ofs[k] & 0xffff)) with spatial weight tap_w[k]; colour weight
 * color_w[|dr| + |dg| + |db|] (768 floats); fp32 sums in tap order, result = round-half-even(sum / wsum).  The tables are OpenCV's
 * ((float)exp(double)), built by the host (imgproc.bilateral_tables).  radius <= 16.  src and dst must not alias. */
int mit_bilateral_u8c3(const uint8_t *src_dev, uint8_t *dst_dev, int B, int H, int W, int radius, int ntaps, const int *tap_ofs_dev,
                       const float *tap_w_dev, const float *color_w_dev, void *stream);
/* out = mask >= thr ? a : b per pixel (C channels): ``img_inpainted * mask_original + img_original * (1 - mask_original)`` with the
 * original mask thresholded at 127 (inpainting_lama_mpe.py:57-61,116). */
int mit_select_u8(const uint8_t *mask_dev, int thr, const uint8_t *a_dev, const uint8_t *b_dev, uint8_t *out_dev, int64_t npix, int C,
                  void *stream);

/* ctd detector: refine_mask on the GPU ------------------------------------------------------------
 * Reference: manga_translator/detection/ctd_utils/textmask.py:158-174 (refine_mask; :29-132 its helpers), called from
 * detection/ctd.py:177 with refine_mode=None.  Three phases over all text-line windows of a page; the host does numpy's histogram /
 * top-k colour / Otsu arithmetic between them (hostglue.refine_mask_gpu).  Bit-identical to hostglue.refine_mask. */
typedef struct MitRefineWindow {
    int x1, y1, x2, y2; /* crop [y1:y2, x1:x2] of the page (enlarge_window of the line's box) */
} MitRefineWindow;
typedef struct MitRefineCand {
    int kind;   /* 0 = none, 1 = inRange(grey, lo, hi), 2 + c = channel c > lo */
    int lo, hi; /* bounds (already rounded / saturated like cv2.inRange) or the Otsu threshold in lo */
    int invert; /* 1: take the complement (minxor_thresh picked cv2.bitwise_not) */
} MitRefineCand;
int64_t mit_ctd_refine_workspace_bytes(const MitRefineWindow *windows, int n);
/* phase A: hist_host[n][4][256] = grey levels under erode3x3(mask crop) > 127, then the three channel histograms of the crop.
 * page_dev u8 [H,W,3], pred_dev u8 [H,W] (the network's mask at page size), windows: HOST array.  Synchronises the stream. */
int mit_ctd_refine_hist(const uint8_t *page_dev, const uint8_t *pred_dev, int H, int W, const MitRefineWindow *windows, int n, int *hist_host,
                        void *workspace_dev, int64_t workspace_bytes, void *stream);
/* phase B: sums_host[n][6] = sum over the crop of (candidate ^ mask) as bytes for 6 raw candidates per line (cands_host[n][6]). */
int mit_ctd_refine_scores(const uint8_t *page_dev, const uint8_t *pred_dev, int H, int W, const MitRefineWindow *windows, int n,
                          const MitRefineCand *cands_host, uint64_t *sums_host, void *workspace_dev, int64_t workspace_bytes, void *stream);
/* phase C: merge_mask_list per line over its (up to 4) candidates in the given order (order_host[n][4]), hole filling, and
 * out_dev[H,W] |= merged inside each window (out_dev must be zeroed by the caller).  Asynchronous on the stream. */
int mit_ctd_refine_merge(const uint8_t *page_dev, const uint8_t *pred_dev, int H, int W, const MitRefineWindow *windows, int n,
                         const MitRefineCand *order_host, uint8_t *out_dev, void *workspace_dev, int64_t workspace_bytes, void *stream);

/* Mask refinement between OCR and inpainting -----------------------------------------------------
 * Reference: manga_translator/mask_refinement/text_mask_utils.py:68-94 (refine_mask -> pydensecrf). */

typedef struct MitCrfCrop {
    int x, y, w, h; /* crop rectangle inside the page, pixels */
} MitCrfCrop;

/* Elliptical dilation of byte masks, batched over jobs (cv2.dilate with cv2.getStructuringElement(MORPH_ELLIPSE, (k, k)), k odd, default
 * anchor and border: text_mask_utils.py:178-195).  Job j reads the source rectangle (sx, sy, sw, sh) — page coordinates — whose pixels
 * live at src_dev + src_off with row pitch spitch, and writes the window (dx, dy, dw, dh) of dst_dev [H, W]: source pixels outside the
 * source rectangle or outside the window count as 0 (the host form dilates the window as a sub-array).  merge = 0: dst = dilated;
 * merge = 1: dst |= dilated for {0, 255} masks (the union over the lines of a page; windows may overlap).  jobs_host is copied to
 * jobs_dev (n_jobs entries of device scratch) on the stream. */
typedef struct MitDilateJob {
    int32_t sx, sy, sw, sh;
    int32_t dx, dy, dw, dh;
    int32_t k, spitch;
    int64_t src_off;
} MitDilateJob;
int mit_mask_dilate_jobs(const uint8_t *src_dev, const MitDilateJob *jobs_host, int n_jobs, uint8_t *dst_dev, int H, int W, int merge,
                         MitDilateJob *jobs_dev, void *stream);
/* buf[i] = buf[i] ? 255 : 0 in place (mask_refinement/__init__.py:29 after the resize back to page size). */
int mit_binarize_u8(uint8_t *buf_dev, int64_t n, void *stream);

/* Bytes of device workspace mit_densecrf_refine needs for these crops (-1: bad crops / batch too large). */
int64_t mit_densecrf_workspace_bytes(const MitCrfCrop *crops, int n_crops);
/* DenseCRF2D mean-field refinement of every text line's mask crop of one page, batched:
 *   unary = -log(clip([1 - m, m], 1e-5, 1)) through unary_lut_dev (256 x 2 floats, built by the host),
 *   pairwise Gaussian (x, y) / sxy_gauss with Potts weight w_gauss + bilateral (x, y) / sxy_bilateral, rgb / srgb_bilateral with
 *   w_bilateral, both through a permutohedral-lattice filter (DIAG_KERNEL, NO_NORMALIZATION), `iterations` mean-field steps,
 *   out = 255 * argmax(Q).   The reference's call is (1, 3, 23, 7, 20, 5).
 * page_dev: u8 [H,W,3] on the device (the bilateral-filtered page); crops: HOST array; mask_dev / out_dev: the crops' masks back to
 * back (crop c at offset sum_{c'<c} w*h, row-major); q_dev (optional): float [sum w*h][2] final marginals.  Synchronises the stream. */
int mit_densecrf_refine(const uint8_t *page_dev, int H, int W, const MitCrfCrop *crops, int n_crops, const uint8_t *mask_dev,
                        uint8_t *out_dev, float *q_dev, float sxy_gauss, float w_gauss, float sxy_bilateral, float srgb_bilateral,
                        float w_bilateral, int iterations, const float *unary_lut_dev, void *workspace_dev, int64_t workspace_bytes,
                        void *stream);

/* LaMa inpainting stage: memory-bound pieces ----------------------------------------------
 * Reference: manga_translator/inpainting/inpainting_lama_mpe.py. */

/* Complex FFT of length h (power of two <= 512) along the row axis of planar re/im data, ncols independent columns (column
 * stride 1), B batches: element (b, t, r, c) at base + b*bs + t*ts + r*hs + c with t = 0 (re) / 1 (im).
 * out[k] = scale * sum_r in[r] * exp(-/+ 2 pi i k r / h) (inverse != 0: +).  twiddle_dev: h/2 (cos, sin) pairs of 2 pi k / h.
 * In-place (in == out with equal strides) is allowed.  The H-axis half of torch.fft.rfftn / irfftn(norm='ortho') in
 * FourierUnit.forward (inpainting_lama_mpe.py:228,252); the W axis is mit_rfft_rows / mit_irfft_rows (or a dense DFT on
 * mit_conv_gemm for widths those do not cover). */
int mit_fft_cols(const float *in_dev, int64_t in_bs, int64_t in_ts, int64_t in_hs, float *out_dev, int64_t out_bs,
                 int64_t out_ts, int64_t out_hs, const float *twiddle_dev, int B, int h, int64_t ncols, int inverse, float scale,
                 void *stream);

/* Real FFT of length w along the W axis of NHWC rows: x[b, h, :, c] (element at in + b*in_bs + h*in_hs + x*in_ws + c) -> the
 * w/2+1 Hermitian bins, planar: element (b, t, h, k, c) at out + b*out_bs + t*out_ts + h*out_hs + k*out_ks + c, t = 0 (re) / 1 (im),
 * multiplied by `scale` (1/sqrt(w) for norm='ortho').  w must be even, <= 512, with w/2 a product of {2,3,5,7,11,13}
 * (mit_rfft_rows_supported); C % 4 == 0.  tables_dev: w/2 (cos, sin) pairs of 2 pi j / (w/2) followed by w/2+1 pairs of
 * 2 pi k / w (fp32, rounded from float64 on the host).  Mixed-radix Stockham butterflies in LDS on the packed sequence
 * z[n] = x[2n] + i x[2n+1].  The W-axis half of torch.fft.rfftn(norm='ortho') in FourierUnit.forward (inpainting_lama_mpe.py:228). */
int mit_rfft_rows_supported(int w);
int mit_rfft_rows(const float *in_dev, int64_t in_bs, int64_t in_hs, int64_t in_ws, float *out_dev, int64_t out_bs, int64_t out_ts,
                  int64_t out_hs, int64_t out_ks, const float *tables_dev, int B, int h, int w, int C, float scale, void *stream);
/* Inverse of mit_rfft_rows: out[b, h, x, c] = scale * irfft_w(in[b, :, h, :, c]) (+ res[b, h, x, c] when res_dev != NULL), the
 * imaginary parts of the DC and Nyquist bins ignored like pocketfft's c2r.  The W-axis half of torch.fft.irfftn(norm='ortho') and
 * the ``x + fu(x)`` of SpectralTransform.forward (inpainting_lama_mpe.py:252,305). */
int mit_irfft_rows(const float *in_dev, int64_t in_bs, int64_t in_ts, int64_t in_hs, int64_t in_ks, float *out_dev, int64_t out_bs,
                   int64_t out_hs, int64_t out_ws, const float *res_dev, int64_t res_bs, int64_t res_hs, int64_t res_ws,
                   const float *tables_dev, int B, int h, int w, int C, float scale, void *stream);

/* u8 page [B,H,W,3] + u8 mask [B,H,W] -> fp32 NHWC [B,H,W,4] = (rgb/255*(1-m), m), m = (mask/255 >= 0.5).
 * Replaces the host-side tensor prep of LamaMPEInpainter._infer :82-92 and the torch.cat of
 * FFCResNetGenerator.forward :604. */
int mit_lama_prep(const uint8_t *img_dev, const uint8_t *mask_dev, float *out_dev, int B, int H, int W, void *stream);

/* The same network input, written as the reflect-padded image [B, H + 2*pad, Wp, 4] (Wp >= W + 2*pad, the surplus columns zero) that
 * the row-packed stem convolution reads: ReflectionPad2d(3) of FFCResNetGenerator.model[0] (inpainting_lama_mpe.py:560) materialised
 * once, so that each kernel row of the 7x7 4->64 stem is ONE contiguous 32-float read (7 pixels x 4 channels + 4 zero-weight floats). */
int mit_lama_prep_padded(const uint8_t *img_dev, const uint8_t *mask_dev, float *out_dev, int B, int H, int W, int pad, int Wp, void *stream);

/* Masked positional-encoding index maps on the 256x256 structure grid:
 * hole = (INTER_AREA resize of the binary mask) > 0; relpos = clipped ring distance; direct = 4 direction bits.
 * Replaces LamaFourier.load_masked_position_encoding :751-807 (cv2.resize + the cv2.filter2D loop on the CPU).
 * ys/yc/yw (xs/xc/xw): DEVICE arrays describing the separable resize: destination row d averages source rows
 * [ys[d], ys[d]+yc[d]) with weights yw[d*ymax + j] (doubles). */
int mit_lama_mpe_index(const uint8_t *mask_dev, int B, int H, int W, const int *ys_dev, const int *yc_dev,
                       const double *yw_dev, int ymax, const int *xs_dev, const int *xc_dev, const double *xw_dev,
                       int xmax, uint8_t *hole_dev, uint8_t *relpos_dev, uint8_t *direct_dev, void *stream);

/* x[B,H,W,64] += alpha5 * rel_pos_emb[rel] + alpha6 * (direct @ direct_emb), with the 256-grid maps
 * nearest-resized through ymap[H] / xmap[W] (cv2.INTER_NEAREST :810-813) and zeroed outside the mask.
 * Replaces MPE.forward :625-632 and the two adds of FFCResNetGenerator.forward :611-612. */
int mit_lama_mpe_add(float *x_dev, const uint8_t *mask_dev, const uint8_t *relpos_dev, const uint8_t *direct_dev,
                     const int *ymap_dev, const int *xmap_dev, const float *emb_dev, const float *dirw_dev, float alpha5,
                     float alpha6, int B, int H, int W, void *stream);
/* The same lookup as per-pixel TABLE ROWS for the stem convolution's epilogue (MitConvGemm.lut_rows): rows[b][y][x] = rel | bits << 16
 * inside the mask, 0 outside, with rel / bits read from the 256-grid maps through ymap / xmap exactly as mit_lama_mpe_add reads them.
 * With lut1[rel] = alpha5 * rel_pos_emb[rel] (128 rows) and lut2[bits] = alpha6 * sum of the set directions' rows of direct_emb (16 rows,
 * summed in bit order), the stem launch performs the two adds of FFCResNetGenerator.forward :611-612 itself — bit-identical to
 * mit_lama_mpe_add after the stem, without the extra read-modify-write of [B,H,W,64]. */
int mit_lama_mpe_rows(const uint8_t *mask_dev, const uint8_t *relpos_dev, const uint8_t *direct_dev, const int *ymap_dev,
                      const int *xmap_dev, int32_t *rows_dev, int B, int H, int W, void *stream);

/* predicted fp32 (pixel stride pred_pixstride floats, 3 used) + page + mask -> inpainted u8 [B,H,W,3]:
 * pred*m + (1-m)*img (:726), *255 truncated to u8 (:111), composited with the original through mask >= 127 (:59-60,117).
 * composite == 0 stops after the truncation (``img_inpainted`` of :111, every pixel from the network): what the plugin resizes
 * back to the page size before compositing there (:112-117) when the page had to be resized. */
int mit_lama_post(const float *pred_dev, int64_t pred_pixstride, const uint8_t *img_dev, const uint8_t *mask_dev,
                  uint8_t *out_dev, int B, int H, int W, int composite, void *stream);

/* Text-detection stage (ctd): memory-bound pieces and NHWC helpers ---------------------------
 * Reference: manga_translator/detection/ctd.py, ctd_utils/. */

/* u8 pages [B,H,W,3] -> letterboxed fp32 NHWC [B,S,S,4] (rgb/255, 4th channel 0; bottom/right zero pad).
 * Replaces preprocess_img (ctd.py:17-28) + letterbox (ctd_utils/utils/imgproc_utils.py:69-100, cv2.resize
 * INTER_LINEAR + copyMakeBorder).  mode 0: no resize; 1: exact 2x shrink (OpenCV routes it to the 2x2 box
 * mean); 2: OpenCV 11-bit fixed-point bilinear with per-axis DEVICE tables (source index, two coefficients). */
int mit_ctd_prep(const uint8_t *img_dev, int B, int H, int W, int nh, int nw, int S, int mode, const int *yidx_dev,
                 const short *ycoef_dev, const int *xidx_dev, const short *xcoef_dev, float *out_dev, void *stream);

/* NHWC max-pool k x k, stride 1, pad k/2 — nn.MaxPool2d of SPPF (yolov5/common.py:181-197). Pixel strides in floats. */
int mit_maxpool_nhwc(const float *in_dev, int64_t in_pixstride, float *out_dev, int64_t out_pixstride, int B, int H,
                     int W, int C, int k, void *stream);
/* NHWC max-pool k x k, stride s, pad p (implicit -inf padding): torchvision ResNet's MaxPool2d(3, 2, 1)
 * (default detector, default_utils/DBNet_resnet34.py:104).  out [B,Ho,Wo,C] dense. */
int mit_maxpool2d_nhwc(const float *in_dev, float *out_dev, int B, int H, int W, int C, int k, int s, int p, void *stream);
/* NHWC 2x2/2 average pool — nn.AvgPool2d(2, 2) of double_conv_c3 (ctd_utils/basemodel.py:28-39). */
int mit_avgpool2_nhwc(const float *in_dev, int64_t in_pixstride, float *out_dev, int64_t out_pixstride, int B, int Ho,
                      int Wo, int C, void *stream);
/* channel-slice copy between NHWC buffers (torch.cat inputs that need a second home, basemodel.py:62-68,102-103). */
int mit_copy_channels(const float *in_dev, int64_t in_pixstride, float *out_dev, int64_t out_pixstride, int64_t npix,
                      int C, void *stream);
/* fp32 map -> u8. mode 0: (uint8)(v*255) truncation (postprocess_mask ctd.py:30-44); mode 1: v > thr
 * (SegDetectorRepresenter.binarize, ctd_utils/utils/db_utils.py:75); mode 2: (uint8)(clip(v,0,1)*255)
 * (ESRGANUpscalerPytorch._infer, upscaling/esrgan_pytorch.py:545). */
int mit_map_to_u8(const float *in_dev, uint8_t *out_dev, int64_t n, int mode, float thr, void *stream);

/* out = a * x + y over n floats (n % 4 == 0): RRDB.forward's ``out * 0.2 + x`` (upscaling/esrgan_pytorch.py:112). */
int mit_axpy(float *out_dev, float a, const float *x_dev, const float *y_dev, int64_t n, void *stream);

/* Host-side (CPU) detector post-processing: thresholded bitmap -> contours -> min-area boxes -> score -> unclip -> boxes.
 * pred: f32 [H,W] probability map (host), bitmap: u8 [H,W] (pred > thresh, host).  Every contour (outer and hole borders,
 * last-found first, at most max_candidates) yields one slot of boxes_out [n,4,2] (int64 x,y scaled to dest_w x dest_h,
 * rounded, clipped; corner order tl,tr,br,bl, or starting at the smallest x+y when roll_start) and scores_out [n]; skipped
 * contours leave zeros (the callers filter on score / non-zero boxes like the reference).  Replaces
 * SegDetectorRepresenter.boxes_from_bitmap of ctd_utils/utils/db_utils.py:127-171 (min_sside 2, box_thresh 0,
 * min_sside_out 0, roll_start 0, unclip 1.5) and default_utils/dbnet_utils.py:97-144 (min_sside 3, box_thresh, min_sside_out 5,
 * roll_start 1), i.e. cv2.findContours/minAreaRect/boxPoints/fillPoly/mean + pyclipper JT_ROUND offset + shapely area/length.
 * boxes_out / scores_out must hold max_candidates slots. */
int mit_boxes_from_bitmap(const float *pred, const uint8_t *bitmap, int H, int W, int dest_w, int dest_h, int max_candidates,
                          float unclip_ratio, float min_sside, float box_thresh, float min_sside_out, int roll_start,
                          int64_t *boxes_out, float *scores_out, int *n_out);
/* The same chain ON THE GPU for a batch of pages (csrc/ctd_boxes.hip; replaces the per-page host call of
 * SegDetectorRepresenter.boxes_from_bitmap, db_utils.py:127-216 / dbnet_utils.py:97-144): pred_dev f32 [B,H,W] with pred_bs elements
 * between pages (0 = H * W; a channel of an NCHW map is 2 H W apart); the bitmap is bitmap_dev u8 [B,H,W] (non-zero = set, bitmap_bs
 * between pages) or, when bitmap_dev is NULL, pred > thresh.  One union-find labelling of the padded bitmap
 * finds every border's first pixel (outer borders: a foreground component's first pixel; hole borders: the left neighbour of an
 * enclosed background component's first pixel — the pixels Suzuki-Abe's raster scan starts them at), one wave per border walks it and
 * does minAreaRect / score / round-join offset / minAreaRect / scaling.  boxes_dev int64 [B,max_candidates,4,2] and scores_dev f32
 * [B,max_candidates] in OpenCV's list order (last border found first; zeros = skipped, as the reference leaves them); counts_dev i32
 * [B] = borders found (min with max_candidates = slots used); overflow_dev i32 [B] != 0: a border of that page exceeds what a wave
 * holds in LDS (8192 points / 4096 corners) — run mit_boxes_from_bitmap for that page (same results).  Results equal the host
 * routine's (tests/test_ctd_boxes_gpu.py).  workspace_dev: mit_boxes_from_bitmap_dev_workspace_bytes(B, H, W, max_candidates) bytes. */
int64_t mit_boxes_from_bitmap_dev_workspace_bytes(int B, int H, int W, int max_candidates);
/* Development aid: a device buffer of 8 x max_candidates x B int64 in which every border records wall_clock64() (100 MHz) at its phase
 * boundaries (walk, rectangle, score, offset, end; [7] = contour length); NULL switches it off (scripts/bench_boxes.py --stamps). */
int mit_boxes_debug_stamps(void *stamps_dev);
int mit_boxes_from_bitmap_dev(const float *pred_dev, int64_t pred_bs, const uint8_t *bitmap_dev, int64_t bitmap_bs, float thresh, int B, int H, int W, int dest_w, int dest_h,
                              int max_candidates, float unclip_ratio, float min_sside, float box_thresh, float min_sside_out, int roll_start,
                              void *workspace_dev, int64_t workspace_bytes, int64_t *boxes_dev, float *scores_dev, int *counts_dev,
                              int *overflow_dev, void *stream);
/* Minimum distance between the quadrilaterals of index pairs (host code, no GPU): quads [n][4][2] doubles (vertex rings), pairs [m][2]
 * -> out [m]; 0 when the two rings touch, cross or contain one another, else the smallest vertex-to-edge distance.  What shapely's
 * Polygon(a.pts).distance(Polygon(b.pts)) returns in Quadrilateral.can_merge / quadrilateral_can_merge_region (utils/generic.py:660-662),
 * which the OCR direction vote (ocr/common.py:12-39) and the text-line merge graph (textline_merge/__init__.py:112-141) evaluate for
 * every pair of neighbouring lines of a page.  The arithmetic is textline.polygon_distance's, operation for operation (doubles). */
int mit_quad_pair_distances(const double *quads, int n, const int32_t *pairs, int m, double *out);
/* Number of contours / border points cv2.findContours(RETR_LIST) would trace in a 0/1 bitmap (diagnostics, tests). */
int mit_find_contours_count(const uint8_t *bitmap, int H, int W, int *n_contours, int64_t *n_points);

/* Mask refinement, host half ahead of the per-line DenseCRF — complete_mask of mask_refinement/text_mask_utils.py:100-170 (host
 * pointers only; CPU code).  mask: u8 [H,W] at the working scale, MODIFIED like the reference's (every line's bounding box outlined
 * with zeros, cv2.rectangle).  Its 8-connected components (cv2.connectedComponentsWithStats) come back as row runs — runs[r] =
 * {y, x0, x1 (exclusive), comp}, comp = 1..n in raster order of the component's first pixel — and assign[comp] is the text line
 * the component belongs to, or -1 (assign[0] = -1): more than 9 pixels, smaller than the line, and either the line polygon covers
 * more than keep_threshold of min(component pixels, line area) of the component's bounding rectangle (largest such ratio, fp32, first
 * wins) or no line does and the nearest polygon is closer to the rectangle's centre than half of max(min(font size, w, h), 10).
 * polys: f64 [M,V,2] (V = 4 for text lines), boxes_xywh: i32 [M,4] (BBox.xywh), font_size: f64 [M]; line_rects: i32 [M,4] = union
 * (x1, y1, x2, y2) of the bounding rectangles of the line's components, -1 when it has none.  runs_cap <= H * ceil(W / 2) runs and
 * assign_cap <= ceil(H / 2) * ceil(W / 2) + 1 entries always suffice. */
typedef struct MitMaskRun {
    int32_t y, x0, x1, comp;
} MitMaskRun;
int mit_mask_assign_lines(uint8_t *mask, int H, int W, const int32_t *boxes_xywh, const double *polys, const double *font_size, int M, int V,
                          double keep_threshold, MitMaskRun *runs, int64_t runs_cap, int32_t *assign, int64_t assign_cap, int32_t *line_rects,
                          int64_t *n_runs_out, int32_t *n_comp_out);
/* The component image of a line inside a rectangle, for n_jobs (line, x, y, w, h) jobs at once (jobs: i32 [n_jobs,5]): out +
 * offsets[j] receives the u8 [h,w] crop that is 255 on the pixels of the components assigned to the job's line — what the reference
 * reads out of its page-sized per-line images (text_mask_utils.py:145,173). */
int mit_mask_line_crops(const MitMaskRun *runs, int64_t n_runs, const int32_t *assign, const int32_t *jobs, int n_jobs, uint8_t *out,
                        const int64_t *offsets);
/* merge_mask_list of the ctd detector's refine_mask (detection/ctd_utils/textmask.py:74-132; filter_with_lines False), HOST
 * pointers: n_cands candidate masks [n_cands][h*w] (0 / 255) with their xor scores, the network's mask window pred_mask [h*w];
 * merged [h*w] receives the result.  Candidates in ascending score, their 8-connected components in raster order of the first
 * pixel, each joined when that lowers sum(xor(merged, erode(pred) > 60)); inpaint_dilate != 0 adds the 5x5 dilation
 * (REFINEMASK_INPAINT); then small holes are filled the same way. */
int mit_merge_mask_list(const uint8_t *cands, const int64_t *scores, int n_cands, const uint8_t *pred_mask, int h, int w,
                        int inpaint_dilate, uint8_t *merged);
/* cv2.threshold(..., THRESH_OTSU)'s threshold (getThreshVal_Otsu_8u) for n 256-bin histograms (host arrays): the per-channel
 * Otsu splits of get_otsuthresh_masklist (ctd_utils/textmask.py:44-54) on the histograms mit_ctd_refine_hist returns. */
int mit_otsu_from_hist(const int32_t *hist, int n, int32_t *thresholds);


/* 48px OCR stage -----------------------------------------------------------------------------
 * Reference: manga_translator/ocr/model_48px.py, ocr/xpos_relative_position.py. */

/* XPOS tables, computed once on the host with the reference's own fp32 expressions
 * (xpos_relative_position.py:9-16,54-59): cos_t/sin_t [imax][40] for index i = 0..imax-1;
 * scale_t / iscale_t [2*pmax][40]: row (p + pmax) holds scale**(p/320) and its reciprocal. */
typedef struct MitXposTables {
    const float *cos_t, *sin_t, *scale_t, *iscale_t;
    int32_t imax, pmax;
} MitXposTables;

/* A packed nn.Linear: w [Kp][ldw] (k-major), out = act((x @ w) * scale + bias); scale may be NULL.  w_split: the same matrix as
 * three bf16 planes (mit_gemm_split_pack) for the split-bf16 tiles, or NULL (fp32 MFMA only). */
typedef struct MitLinear {
    const float *w, *scale, *bias;
    int64_t ldw;
    int32_t K, N, Kp, Np;
    const uint16_t *w_split;
} MitLinear;

typedef struct MitOcrDecoderLayer {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ln3_w, *ln3_b;
    MitLinear qkv;  /* self_attn q|k|v fused, N = 960; the q columns carry the head_dim**-0.5 scaling */
    MitLinear out;  /* self_attn.out_proj */
    MitLinear q2;   /* multihead_attn.q_proj (scaled) */
    MitLinear out2; /* multihead_attn.out_proj */
    MitLinear ff1, ff2;
} MitOcrDecoderLayer;

typedef struct MitOcr48Decoder {
    MitOcrDecoderLayer layers[5];
    const float *embd; /* [dict][320] */
    MitLinear pred1;   /* + GELU */
    MitLinear pred;    /* tied to embd, N = dict */
    MitLinear color1;  /* 320 -> 64, ReLU */
    MitLinear color_heads; /* 64 -> 10 = fg(3) | bg(3) | fg_ind(2) | bg_ind(2) */
    MitXposTables xpos;
    int32_t dict_size, _pad;
} MitOcr48Decoder;

typedef struct MitOcr48DecodeArgs {
    int32_t N, L;               /* text lines; padded encoder-memory length */
    const float *mem_k;         /* [5][N][L][320] cross-attention keys (k_proj + XPOS) per decoder layer */
    const float *mem_v;         /* [5][N][L][320] */
    const int32_t *mem_len;     /* [N] valid memory length (w+3)/4+2 (model_48px.py:684-688) */
    int32_t max_seq_length;     /* T: decode steps (255 in the reference call, :120) */
    int32_t start_tok, end_tok, max_finished, suppress_eos;
    void *workspace;            /* device scratch of mit_ocr48_decode_workspace_bytes() */
    int64_t workspace_bytes;
    int32_t *res_tok;           /* [N][T+1] tokens incl. the start token */
    int32_t *res_len;           /* [N] number of valid tokens in res_tok */
    float *res_prob;            /* [N] exp(sum of log-probs) (:752) */
    int32_t *res_row;           /* [N] beam row whose activation cache feeds the colour heads */
    float *colors;              /* [N*5][T][12] colour-head outputs for every beam row (10 used) */
    float *trace_logits;        /* optional [T][N*5][dict] raw logits (pred(pred1(decoded)), :713); NULL in production */
    int32_t *trace_hist;        /* optional [T][N*5][T+1] beam tokens after each step */
    int32_t steps_run;          /* out: steps executed */
    int32_t graph_mode;         /* 1: replay the steps from a hipGraph (one launch per step instead of 74); 2: never; 0: as
                                 * MIT_OCR_DECODE_GRAPH says (default off: measured 50.2 vs 50.6 ms per page, the loop is bound by its
                                 * kernels' latency, not by launches).  Same kernels either way: identical results. */
} MitOcr48DecodeArgs;

/* One text line to rectify: cv2.warpPerspective of the page crop [y1:y1+ch, x1:x1+cw] to (dw, dh) with inverse map minv
 * (row-major 3x3, destination -> crop coordinates), then ROTATE_90_COUNTERCLOCKWISE when vertical; the result lands in
 * row out_row of the chunk tensor.  Replaces Quadrilateral.get_transformed_region (utils/generic.py:445-481). */
typedef struct MitWarpLine {
    double minv[9];
    int32_t page, x1, y1, cw, ch, dw, dh, vertical, out_row, _pad;
} MitWarpLine;
/* pages u8 [P,H,W,3] -> chunk u8 [N,Hout,Wp,3], zero beyond each line's width (model_48px.py:83-91: np.zeros + copy).
 * 8-bit bilinear with OpenCV's fixed-point rules (INTER_BITS 5, 15-bit weights), BORDER_CONSTANT 0. */
int mit_ocr_warp_lines(const uint8_t *pages_dev, int H, int W, const MitWarpLine *lines_dev, int n_lines, uint8_t *out_dev,
                       int Hout, int Wp, void *stream);
int mit_ocr_prep(const uint8_t *lines_dev, float *out_dev, int N, int H, int Wp, void *stream);
/* depthwise k x k conv + per-channel scale/bias (ConvNeXtBlock.dwconv + norm, model_48px.py:195-196,205-206). w [k*k][C]. */
int mit_dwconv_nhwc(const float *in_dev, const float *w_dev, const float *scale_dev, const float *bias_dev, float *out_dev,
                    int B, int H, int W, int C, int k, void *stream);
/* Ragged variant: nsegs images [B_s,H_s,W_s,C] stored back to back along the pixel axis (the OCR chunks of a page group,
 * whose padded widths differ, model_48px.py:83-86).  Segment s starts at pixel pixel_start; group_start = running count of
 * (image row, 4-column group) work items = sum over earlier segments of B*H*ceil(W/4); total_groups = that sum over all. */
typedef struct MitRaggedSeg {
    int64_t pixel_start, group_start;
    int32_t B, H, W, _pad;
} MitRaggedSeg;
int mit_dwconv_nhwc_ragged(const float *in_dev, const float *w_dev, const float *scale_dev, const float *bias_dev,
                           float *out_dev, const MitRaggedSeg *segs_dev, int nsegs, int64_t total_groups, int C, int k,
                           void *stream);
/* The same when every segment has height common_H: rows are processed YT = 4 (or 2) at a time per thread with the weights in LDS
 * (bit-identical results, about a third of the L1 loads).  common_H <= 0, odd heights or K*K*C*4 > 64 KB fall back to the call above. */
int mit_dwconv_nhwc_ragged_rows(const float *in_dev, const float *w_dev, const float *scale_dev, const float *bias_dev,
                                float *out_dev, const MitRaggedSeg *segs_dev, int nsegs, int64_t total_groups, int C, int k,
                                int common_H, void *stream);
/* nn.LayerNorm over the last dim (transformer norm1/2/3). */
int mit_layernorm(const float *in_dev, int64_t in_rowstride, const float *w_dev, const float *b_dev, float *out_dev,
                  int64_t out_rowstride, int rows, int D, float eps, void *stream);
/* XPOS.forward on [R, T, 4*80] (row / time strides in floats): index i0+t, centred position p0+t. */
int mit_xpos_rotate(const float *in_dev, int64_t in_rs, int64_t in_ts, float *out_dev, int64_t out_rs, int64_t out_ts, int R,
                    int T, int i0, int p0, int downscale, const MitXposTables *tables, void *stream);
/* softmax(q k^T + key-padding mask) v for 4 heads x 80 (XposMultiheadAttention.forward :369-384); q already scaled and
 * rotated, k rotated.  Row r reads keys/values of row r / kv_div; klen_dev (optional) = valid keys per kv row. */
int mit_attention(const float *q_dev, int64_t q_rs, int64_t q_ts, const float *k_dev, int64_t k_rs, int64_t k_ts,
                  const float *v_dev, int64_t v_rs, int64_t v_ts, float *out_dev, int64_t o_rs, int64_t o_ts,
                  const int *klen_dev, int R, int Tq, int Tk, int kv_div, void *stream);
/* The encoder's self-attention (XposMultiheadAttention.forward, ocr/model_48px.py:327-394) for ALL lines of a page group in one launch
 * with the XPOS rotation of q (scale) and k (inverse scale) folded in (xpos_relative_position.py:44-71): line r owns lines[r] = {first row,
 * L_r} rows of the flat [rows, heads * head_dim] q / k / v / out tensors (row stride in floats), L_r = its chunk's memory length = its
 * query and key count, positions centred per chunk (p0 = -((L_r + 1) / 2)), klen[r] valid keys.  Bitwise equal to mit_xpos_rotate (q),
 * mit_xpos_rotate (k, downscale) and mit_attention chunk by chunk.  Lmax = the longest line (LDS sizing). */
int mit_attention_lines_xpos(const float *q_dev, const float *k_dev, const float *v_dev, float *out_dev, int64_t row_stride,
                             const int32_t *lines_dev, const int *klen_dev, int n_lines, int Lmax, int heads, int head_dim,
                             const MitXposTables *tables, void *stream);
/* The decoder's self-attention kernel: 1 (default) = one workgroup per row stages the row's key history once for its four heads
 * (attention_self_kernel), 0 = one single-wave workgroup per (head, row) (attention_kernel); bitwise the same results.  on < 0 only
 * queries.  Returns the previous value.  Nothing in the reference corresponds to it (A/B switch). */
int mit_attention_self_rows_set(int on);
/* Longest line (Lmax) whose keys + per-wave score rows fit mit_attention_lines_xpos's LDS form for this head_dim (308 at head_dim 80);
 * longer lines take the chunk-by-chunk path (mit_xpos_rotate + mit_attention), which has no such limit.  0 for an unsupported head_dim. */
int mit_attention_lines_xpos_max_len(int head_dim);
/* Cross-attention memory of one decoder layer for the same lines: mem_k[first_line + r][t][:] = the XPOS-rotated (inverse scale) k rows,
 * mem_v[...] = the v rows, t < L_r (line_stride floats between lines of the pooled [n, Lmax, heads * head_dim] memory); n_rows = the
 * rows the lines cover.  Replaces one mit_xpos_rotate and one copy per chunk (OCR.infer_beam_batch_tensor's memory, :678-704). */
int mit_memory_kv_lines(const float *k_dev, const float *v_dev, int64_t row_stride, float *mem_k_dev, float *mem_v_dev, int64_t line_stride,
                        const int32_t *lines_dev, int n_lines, int first_line, int64_t n_rows, int Lmax, int head_dim,
                        const MitXposTables *tables, void *stream);
/* Same with an explicit head layout (heads x head_dim contiguous in the feature axis): 8 x 40 for the 48px_ctc encoder's
 * nn.MultiheadAttention (model_48px_ctc.py:216-217,259-265; q already scaled by head_dim**-0.5, PE already added to q/k). */
int mit_attention_heads(const float *q_dev, int64_t q_rs, int64_t q_ts, const float *k_dev, int64_t k_rs, int64_t k_ts,
                        const float *v_dev, int64_t v_rs, int64_t v_ts, float *out_dev, int64_t o_rs, int64_t o_ts,
                        const int *klen_dev, int R, int Tq, int Tk, int kv_div, int heads, int head_dim, void *stream);
/* nn.AvgPool2d(kernel, stride, padding) with count_include_pad=True on NHWC fp32 — the FAN backbone's pools
 * (model_48px_ctc.py:291,297,303: 2/2/0 twice, then kernel 2, stride (2,1), padding (0,1)). out [B,Ho,Wo,C] dense. */
int mit_avgpool_nhwc(const float *in_dev, float *out_dev, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                     int pw, void *stream);
/* y = relu?(x * scale[c] + bias[c]) per pixel: eval BatchNorm2d (+ ReLU) that cannot ride in a conv epilogue because its
 * input also feeds a residual (pre-activation BasicBlock.forward, model_48px_ctc.py:389-403). */
int mit_affine_act_nhwc(const float *in_dev, int64_t in_pixstride, const float *scale_dev, const float *bias_dev, float *out_dev,
                        int64_t out_pixstride, int64_t npix, int C, int relu, void *stream);
/* u8 RGB [npix,3] -> fp32 [npix,4] (4th channel 0).  mode 0: (x-127.5)/127.5 (model_48px.py:115); mode 1: x/127.5 - 1
 * (det_batch_forward_default, detection/default.py:19); mode 2: x/255.  Each is the reference's own fp32 expression. */
int mit_u8_to_f32_nhwc4(const uint8_t *in_dev, float *out_dev, int64_t npix, int mode, void *stream);
/* x <- sigmoid(x): the ``db.sigmoid()`` that det_batch_forward_default applies on top of DBHead's output (default.py:23). */
int mit_sigmoid_inplace(float *x_dev, int64_t n, void *stream);
/* x <- gelu(x) (erf form) over n floats: the nn.GELU of char_pred_norm (model_48px_ctc.py:435). */
int mit_gelu_inplace(float *x_dev, int64_t n, void *stream);
/* log_softmax over D columns + the 5 largest (value, index) per row, ties to the lower index; suppress_tok < 0: none.
 * Row r: vals_dev[5r..], idx_dev[5r..].  decode_ctc_top1's log_softmax + max (model_48px_ctc.py:477-478) uses entry 0. */
int mit_logsoftmax_top5(const float *logits_dev, int64_t ld, int R, int D, int suppress_tok, float *vals_dev, int *idx_dev,
                        void *stream);
/* The whole beam search of OCR.infer_beam_batch_tensor (:691-784) after the encoder, as one native call: per step
 * embedding -> 5 decoder layers (KV cache instead of the reference's per-step K/V recomputation) -> pred1/pred ->
 * log-softmax/top-5 -> beam bookkeeping, all on `stream`; synchronises the stream every few steps to test for early exit. */
int64_t mit_ocr48_decode_workspace_bytes(int N, int T, int dict_size);
int mit_ocr48_decode(const MitOcr48Decoder *dec, MitOcr48DecodeArgs *args, void *stream);
/* Largest number of decoder rows (5 N: lines x beams) whose steps run in the few-row form — every Linear of a step as one wave per
 * 32 x 32 output block on bf16-plane activations (pgemm_rows_kernel), the LayerNorm / attention kernels producing the planes — instead
 * of the tiled form that pays off on full batches.  Both forms give identical results (tests/test_ocr_gpu.py); the few-row one takes a
 * third of the time per Linear at one page.  Needs GEMM mode 6 | 9.  rows < 0 only queries; 0 = never.  Initial value: MIT_OCR_ROWS_MAX in
 * the environment, else 2560 (16 pages of 32 lines).  Returns the previous value.  Nothing in the reference corresponds to it. */
int mit_ocr48_decode_rows_max_set(int rows);

#ifdef __cplusplus
}
#endif
#endif /* MIT_HIP_H */
