// ocr_kernels.h — internal launch helpers of ocr_kernels.hip used by the native decoder loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mit_hip.h"

// Planar output of a kernel whose result feeds a few-row planar GEMM (pgemm_rows.h): the fp32 value a kernel would store goes out as its
// three bf16 planes [3][K8][ld][8] instead (bf16_split.h: the split a GEMM tile would apply to the fp32 value itself).  p == NULL: none.
struct OcrPlanes {
    uint16_t *p;
    int64_t ld;   // rows per (plane, k-cell) slab
    int K8;       // k-cells per plane (row length / 8)
};
int ocrk_layernorm(const float *in, int64_t in_rs, const float *w, const float *b, float *out, int64_t out_rs, int rows,
                    int D, float eps, hipStream_t s, const OcrPlanes *planes = nullptr);   // planes: instead of out (D % 8 == 0)
// dstep != NULL: step-dependent arguments come from the decoder's device-resident step counter (see the kernels' comments)
void ocrk_xpos_rotate(const float *in, int64_t in_rs, int64_t in_ts, float *out, int64_t out_rs, int64_t out_ts, int R, int T,
                      int i0, int p0, int downscale, const MitXposTables &tb, hipStream_t s, const int *dstep = nullptr, int dyn_mode = 0,
                      int64_t dyn_in = 0);
// XPOS rotation folded into the decoder's one-query attention (Tq == 1): the query is the token of position `step` (scaled up), and
// with rot_k the keys are the raw history 0 .. step, rotated (scaled down) as they are read — positions centred on the history,
// origin -((step + 2) / 2).  The rotated value of an element is xpos_rotate_kernel's fp32 expression, so the result is bitwise the one
// of rotate + (rotate +) attention.  dstep != NULL: step = *dstep and the query row starts step * q_dyn floats further.
struct OcrAttXpos {
    const float *cos_t, *sin_t, *scale_t, *iscale_t;   // cos_t == NULL: no rotation
    int pmax, step, rot_k;
    int64_t q_dyn;
};
void ocrk_attention(const float *Q, int64_t q_rs, int64_t q_ts, const float *K, int64_t k_rs, int64_t k_ts, const float *V,
                    int64_t v_rs, int64_t v_ts, float *O, int64_t o_rs, int64_t o_ts, const int *klen, int R, int Tq, int Tk,
                    int kv_div, hipStream_t s, int heads = 4, int head_dim = 80, const int *dstep = nullptr,
                    const OcrAttXpos *xpos = nullptr, const OcrPlanes *o_planes = nullptr);   // o_planes: instead of O (Tq == 1, head_dim % 8 == 0)
// The q projection of the decoder's cross-attention inside its attention kernel (few rows; heads 4 x 80, 5 beams per line):
// q = LayerNorm(x) @ W + bias, W as the planes of pgemm_rows.h.  Bit for bit mit_pgemm_rows_ln + ocrk_attention.
struct OcrAttQProj {
    const float *x;           // residual stream [R][ldx]
    int64_t ldx;
    const float *ln_w, *ln_b;
    float eps;
    const uint16_t *w_planes; // [3][K / 8][ldw][8], K = 320
    int64_t ldw;
    const float *scale;       // [N] or NULL: the Linear's epilogue is acc * scale + bias
    const float *bias;        // [N]
};
bool ocrk_cross_attention_qproj(const OcrAttQProj &qp, const float *K, int64_t k_rs, int64_t k_ts, const float *V, int64_t v_rs, int64_t v_ts,
                                const int *klen, int R, int Tk, hipStream_t s, const int *dstep, const OcrAttXpos *xpos, const OcrPlanes *o_planes);
void ocrk_embed(const int *tok, int64_t tok_stride, const float *E, float *out, int R, int D, hipStream_t s, const int *tok1 = nullptr,
                const int *dstep = nullptr);
void ocrk_beam_dyn(const float *vals, const int *idx, int *hist0, int *hist1, int hist_ld, float *logp0, float *logp1, int *done,
                   int *res_row, int *res_len, float *res_prob, int *res_tok, int *done_count, int N, const int *dstep, int start_tok,
                   int end_tok, int max_finished, hipStream_t s, const float *next_E = nullptr, float *next_out = nullptr, int next_D = 0);
void ocrk_step_advance(int *dstep, hipStream_t s);
void ocrk_logsoftmax_top5(const float *logits, int64_t ld, int R, int D, int suppress_tok, float *vals, int *idx,
                          float *logp_out, hipStream_t s);
// next_E / next_out / next_D (beam kernels, optional): the embedding rows [R][next_D] of the tokens just chosen = the residual stream of
// the next step, written by the bookkeeping kernel instead of a separate ocrk_embed launch at the head of that step
void ocrk_beam_init(const float *vals, const int *idx, int *hist, int hist_ld, float *logp, int N, int start_tok,
                    hipStream_t s, const float *next_E = nullptr, float *next_out = nullptr, int next_D = 0);
void ocrk_beam_step(const float *vals, const int *idx, const int *hist_in, int *hist_out, int hist_ld, const float *logp_in,
                    float *logp_out, int *done, int *res_row, int *res_len, float *res_prob, int *res_tok, int *done_count,
                    int N, int step, int end_tok, int max_finished, hipStream_t s, const float *next_E = nullptr, float *next_out = nullptr,
                    int next_D = 0);
void ocrk_beam_finalize(const int *hist, int hist_ld, const float *logp, int *done, int *res_row, int *res_len,
                        float *res_prob, int *res_tok, int N, int len, hipStream_t s);
