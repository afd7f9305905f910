// ocr_decoder.hip — native beam-search decoder loop of the 48px OCR (OCR.infer_beam_batch_tensor,
// manga_translator/ocr/model_48px.py:691-784).
//
// The reference runs ~75 tiny torch ops per step from Python and syncs to the host several times per
// step (:741-772).  Here the whole loop is one C call: every step enqueues its kernels on the stream
// from C++ (no Python, no per-step host sync), keeps a real K/V cache per layer (the reference
// re-projects the whole history each step, :561-566 — identical values, since row r's history never
// changes: the reference re-orders hypotheses but not caches, :730-735), and does the top-k / beam
// bookkeeping on the device.  Finished samples stay in place (frozen results) instead of being
// compacted away; the host polls a device counter every few steps for the early exit.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/mit_hip.h"
#include <atomic>
#include "common.h"
#include "ocr_kernels.h"
#include "pgemm_rows.h"

namespace {

constexpr int E = 320;
constexpr int FF = 2048;

constexpr int FF2_SPLITK_MAX_ROWS = 640;   // rows (lines x beams) up to which the few-row FFN output Linear cuts K across four waves

struct Ws {
    float *tgt, *nrm, *qkv, *att, *q2, *ffh, *decoded, *p1, *logits, *vals, *logp, *cfeat, *part;
    int *idx, *hist, *done, *done_count, *dstep;
    // the few-row form (rows_path): activations that only feed a Linear live as bf16 planes [3][K / 8][Rp][8] (pgemm_rows.h)
    uint16_t *nrm_p, *att_p, *ffh_p, *dec_p, *p1_p;
    int64_t Rp;
};

// Few rows, long contraction (the FFN's second Linear, K = 2048, at one page: R = lines x beams = 160 rows): a 64-row tiling is
// (R / 64) x (320 / 64) = 15 workgroups that each walk all 128 K-tiles — 45 us of a 256-CU chip for 0.2 GFLOP.  Up to SPLITK_MAX_M
// rows such a GEMM is cut along K into slices of SPLITK_SLICE, computed as batch entries of ONE launch into a partial buffer
// [S][M][Np], and summed in slice order by splitk_reduce_kernel, which applies the epilogue.  Deterministic; the sum order differs
// from the k-sequential chain of the single-launch form (fp32 rounding only) — which is why it is an opt-in experiment
// (MIT_OCR_SPLITK=1, see gemm()): measured 44 -> 50 ms per page of the B = 1 OCR call without it.
constexpr int SPLITK_MAX_M = 2048, SPLITK_MIN_K = 1024, SPLITK_SLICE = 128;

inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

int64_t carve(Ws *w, char *base, int N, int T, int D) {
    const int64_t R = (int64_t)N * 5;
    const int64_t Dp = (D + 3) / 4 * 4;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    float *tgt = (float *)take(R * E * 4);
    float *nrm = (float *)take(R * E * 4);
    float *qkv = (float *)take(5 * 3 * R * T * E * 4);
    float *att = (float *)take(R * E * 4);
    float *q2 = (float *)take(R * E * 4);
    float *ffh = (float *)take(R * FF * 4);
    float *decoded = (float *)take(R * T * E * 4);
    float *p1 = (float *)take(R * E * 4);
    float *logits = (float *)take(R * Dp * 4);
    float *vals = (float *)take(R * 5 * 4);
    float *logp = (float *)take(2 * R * 4);
    float *cfeat = (float *)take(R * T * 64 * 4);
    float *part = (float *)take(R <= SPLITK_MAX_M ? (int64_t)(FF / SPLITK_SLICE) * R * E * 4 : 0);
    int *idx = (int *)take(R * 5 * 4);
    int *hist = (int *)take(2 * R * (T + 1) * 4);
    int *done = (int *)take((int64_t)N * 4);
    int *done_count = (int *)take(256);
    int *dstep = (int *)take(256);
    const int64_t Rp = (R + 31) / 32 * 32;
    uint16_t *nrm_p = (uint16_t *)take(3 * E * Rp * 2);
    uint16_t *att_p = (uint16_t *)take(3 * E * Rp * 2);
    uint16_t *ffh_p = (uint16_t *)take(3 * FF * Rp * 2);
    uint16_t *dec_p = (uint16_t *)take(3 * E * Rp * 2);
    uint16_t *p1_p = (uint16_t *)take(3 * E * Rp * 2);
    if (w) *w = Ws{tgt, nrm, qkv, att, q2, ffh, decoded, p1, logits, vals, logp, cfeat, part, idx, hist, done, done_count, dstep,
                   nrm_p, att_p, ffh_p, dec_p, p1_p, Rp};
    return off;
}

// sum of the S partial products in slice order + the epilogue of mit_conv_gemm: C = act(sum * scale + bias) + post
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, const int S, const int M, const int N4, const int Np,
                                                           float *C, const int64_t ldc, const float *__restrict__ scale,
                                                           const float *__restrict__ bias, const int act, const float *post, const int64_t ldpost) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N4) return;
    const int m = idx / N4, n = (idx - m * N4) * 4;
    const int64_t slice = (int64_t)M * Np;
    const float *p = part + (int64_t)m * Np + n;
    float4 v = *reinterpret_cast<const float4 *>(p);
    for (int z = 1; z < S; ++z) {
        const float4 t = *reinterpret_cast<const float4 *>(p + z * slice);
        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
    }
    if (scale) {
        const float4 sc = *reinterpret_cast<const float4 *>(scale + n);
        v.x *= sc.x, v.y *= sc.y, v.z *= sc.z, v.w *= sc.w;
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + n);
        v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
    }
    if (act == MIT_ACT_RELU) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    if (post) {
        const float4 r = *reinterpret_cast<const float4 *>(post + (int64_t)m * ldpost + n);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
    }
    *reinterpret_cast<float4 *>(C + (int64_t)m * ldc + n) = v;
}

// C[M x N] = act((A[M x K] @ W) * scale + bias) + post, rows of A / C / post strided.  ``part``: partial-sum scratch for the
// split-K form (NULL = never split).
// dyn / a_dyn / c_dyn: device-resident step counter and the per-step strides of A and C (MitConvGemm.dyn), for the graph-replayed steps.
int gemm(const MitLinear &lin, const float *A, int64_t lda, float *Cp, int64_t ldc, int M, int act, const float *post,
         int64_t ldpost, hipStream_t s, int nsplit = 0, int64_t nhi = 0, float *part = nullptr, const int *dyn = nullptr,
         int64_t a_dyn = 0, int64_t c_dyn = 0) {
    MitConvGemm d;
    memset(&d, 0, sizeof(d));
    d.dyn = dyn; d.a_dyn = a_dyn; d.c_dyn = c_dyn;
    d.a = A;
    d.a_xs = lda;
    d.NB = 1; d.Hi = 1; d.Wi = M; d.Ho = 1; d.Wo = M; d.sy = 1; d.sx = 1;
    d.ntaps = 1; d.pad_mode = MIT_PAD_ZERO;
    d.w = lin.w; d.ldw = lin.ldw; d.Nw = lin.Np;
    d.N = lin.N;
    // Opt-in (MIT_OCR_SPLITK=1): the slice-wise sum is deterministic but rounds differently from the k-sequential chain of the
    // single-launch form, and with it a page decoded alone would no longer give bit for bit the logits it gives inside a batch — beam
    // search turns such last-bit differences into different tokens whenever two hypotheses score within them.  Off by default.
    static const bool splitk_on = getenv("MIT_OCR_SPLITK") != nullptr && atoi(getenv("MIT_OCR_SPLITK")) != 0;
    if (part && splitk_on && !dyn && M <= SPLITK_MAX_M && lin.K >= SPLITK_MIN_K && lin.K % SPLITK_SLICE == 0 && lin.Kp == lin.K && !nsplit &&
        (lin.N & 3) == 0 && (act == MIT_ACT_NONE || act == MIT_ACT_RELU) && !(ldc & 3) && !(ldpost & 3)) {
        const int S = lin.K / SPLITK_SLICE;
        d.Cin = SPLITK_SLICE; d.Kw = SPLITK_SLICE;
        d.Z = S; d.zdiv = 1 << 30;                       // z1 = 0, z0 = slice
        d.a_zs0 = SPLITK_SLICE;                          // A rows are k-contiguous: slice z starts SPLITK_SLICE floats further
        d.w_zs0 = (int64_t)SPLITK_SLICE * lin.ldw;
        d.c.base = part; d.c.xs = lin.Np; d.c.zs0 = (int64_t)M * lin.Np;
        d.act = MIT_ACT_NONE;
        if (mit_conv_gemm(&d, s)) return 1;
        const int N4 = lin.N / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((M * N4 + 255) / 256), dim3(256), 0, s, part, S, M, N4, lin.Np, Cp, ldc, lin.scale,
                           lin.bias, act, post, ldpost);
        return 0;
    }
    d.Cin = lin.K; d.Kw = lin.Kp;
    d.Z = 1; d.zdiv = 1;
    d.w_split = lin.w_split;  // planes attached by the packer in a split GEMM mode (NULL otherwise); the launcher decides by the mode of the moment
    d.c.base = Cp; d.c.xs = ldc; d.c.nsplit = nsplit; d.c.nhi = nhi;
    if (post) {
        d.post.base = const_cast<float *>(post);
        d.post.xs = ldpost;
    }
    d.scale = lin.scale; d.bias = lin.bias; d.act = act;
    return mit_conv_gemm(&d, s);
}

// The same Linear on planar activations (pgemm_rows.h): C fp32 (optional, with the column split / step offset of gemm()) and / or planes.
int pgemm(const MitLinear &lin, const uint16_t *a_planes, int64_t lda, int M, float *Cp, int64_t ldc, int act, const float *post,
          int64_t ldpost, uint16_t *c_planes, int64_t ld_cp, hipStream_t s, int nsplit = 0, int64_t nhi = 0, const int *dyn = nullptr,
          int64_t c_dyn = 0, int splitk = 0) {
    MitPGemm d;
    memset(&d, 0, sizeof(d));
    d.a_planes = a_planes; d.lda = lda;
    d.w_planes = lin.w_split; d.ldw = lin.ldw;
    d.M = M; d.N = lin.N; d.K = lin.K; d.Z = 1;
    d.c = Cp; d.ldc = ldc;
    d.post = post; d.ld_post = ldpost;
    d.scale = lin.scale; d.bias = lin.bias; d.act = act;
    d.nprod = 0;  // the GEMM mode of the moment
    PgRowsExt x;
    memset(&x, 0, sizeof(x));
    x.nsplit = nsplit; x.nhi = nhi; x.dyn = dyn; x.c_dyn = c_dyn; x.splitk = splitk;
    if (Cp) x.also_planes = c_planes, x.also_ld = ld_cp;
    else d.c_planes = c_planes, d.ld_cp = ld_cp;
    return mit_pgemm_rows(d, x, s);
}
// The same with A = LayerNorm(x) computed by the GEMM's own waves (pgemm_rows_ln.hip; K == 320): bit for bit ocrk_layernorm + pgemm.
int pgemm_ln(const MitLinear &lin, const float *xin, int64_t ldx, const float *ln_w, const float *ln_b, int M, float *Cp, int64_t ldc, int act,
             uint16_t *c_planes, int64_t ld_cp, hipStream_t s, int nsplit = 0, int64_t nhi = 0, const int *dyn = nullptr, int64_t c_dyn = 0) {
    MitPGemm d;
    memset(&d, 0, sizeof(d));
    d.w_planes = lin.w_split; d.ldw = lin.ldw;
    d.M = M; d.N = lin.N; d.K = lin.K; d.Z = 1;
    d.c = Cp; d.ldc = ldc;
    d.scale = lin.scale; d.bias = lin.bias; d.act = act;
    d.nprod = 0;
    PgRowsExt x;
    memset(&x, 0, sizeof(x));
    x.nsplit = nsplit; x.nhi = nhi; x.dyn = dyn; x.c_dyn = c_dyn;
    if (Cp) x.also_planes = c_planes, x.also_ld = ld_cp;
    else d.c_planes = c_planes, d.ld_cp = ld_cp;
    const PgRowsLn ln{xin, ldx, ln_w, ln_b, 1e-5f};
    return mit_pgemm_rows_ln(d, x, ln, s);
}
inline bool rows_ok(const MitLinear &l) { return l.w_split && l.Kp == l.K && (l.K % 16) == 0 && (l.N % 8) == 0; }

__global__ void fill_int_kernel(int *p, int64_t n, int v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

__global__ void copy_hist_kernel(const int *src, int *dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}


// instantiated step graphs whose launches may still be in flight; destroyed once their event has completed
struct PendingGraph {
    hipGraphExec_t exec;
    hipGraph_t graph;
    hipEvent_t done;
};
thread_local std::vector<PendingGraph> g_pending;

void reap_graphs() {
    size_t k = 0;
    for (size_t i = 0; i < g_pending.size(); ++i) {
        PendingGraph &p = g_pending[i];
        if (p.done && hipEventQuery(p.done) == hipSuccess) {
            (void)hipGraphExecDestroy(p.exec);
            (void)hipGraphDestroy(p.graph);
            (void)hipEventDestroy(p.done);
        } else {
            g_pending[k++] = p;
        }
    }
    g_pending.resize(k);
}

}  // namespace

namespace {
std::atomic<int> g_rows_max{-1};
int rows_max_now() {
    int v = g_rows_max.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("MIT_OCR_ROWS_MAX");
        v = (e && *e) ? atoi(e) : 2560;
        if (v < 0) v = 0;
        g_rows_max.store(v, std::memory_order_relaxed);
    }
    return v;
}
}  // namespace

extern "C" int mit_ocr48_decode_rows_max_set(int rows) {
    const int prev = rows_max_now();
    if (rows >= 0) g_rows_max.store(rows, std::memory_order_relaxed);
    return prev;
}

extern "C" int64_t mit_ocr48_decode_workspace_bytes(int N, int T, int dict_size) {
    if (N <= 0 || T <= 0 || dict_size <= 0) return 0;
    return carve(nullptr, nullptr, N, T, dict_size);
}

extern "C" int mit_ocr48_decode(const MitOcr48Decoder *dec, MitOcr48DecodeArgs *a, void *stream) {
    if (!dec || !a) return mit_set_error("mit_ocr48_decode: null argument");
    const int N = a->N, L = a->L, T = a->max_seq_length, D = dec->dict_size;
    if (N <= 0 || L <= 0 || T <= 0) return mit_set_error("mit_ocr48_decode: empty problem");
    if (!a->mem_k || !a->mem_v || !a->mem_len || !a->workspace || !a->res_tok || !a->res_len || !a->res_prob || !a->res_row || !a->colors)
        return mit_set_error("mit_ocr48_decode: null buffer");
    if (a->workspace_bytes < carve(nullptr, nullptr, N, T, D)) return mit_set_error("mit_ocr48_decode: workspace too small");
    if (T + 1 > dec->xpos.imax || (T + 1) / 2 + 1 >= dec->xpos.pmax) return mit_set_error("mit_ocr48_decode: XPOS tables too small for T");
    hipStream_t s = (hipStream_t)stream;
    Ws w;
    carve(&w, (char *)a->workspace, N, T, D);
    const int R = N * 5;
    const int64_t Dp = (D + 3) / 4 * 4;
    const int64_t TE = (int64_t)T * E;
    const int hist_ld = T + 1;
    int *hist[2] = {w.hist, w.hist + (int64_t)R * hist_ld};
    float *logp[2] = {w.logp, w.logp + R};

    hipLaunchKernelGGL(fill_int_kernel, dim3(64), dim3(256), 0, s, hist[0], (int64_t)2 * R * hist_ld, a->start_tok);
    MIT_CHECK_HIP(hipMemsetAsync(w.done, 0, (size_t)N * 4, s));
    MIT_CHECK_HIP(hipMemsetAsync(w.done_count, 0, 4, s));
    MIT_CHECK_HIP(hipMemsetAsync(w.decoded, 0, (size_t)R * TE * 4, s));
    MIT_CHECK_HIP(hipMemsetAsync(a->res_len, 0, (size_t)N * 4, s));
    ocrk_embed(hist[0], hist_ld, dec->embd, w.tgt, R, E, s);   // step 0: the start tokens; every later step's rows come from the beam kernel

    int cur = 0, steps = 0;
    // One beam-search step as a launch sequence.  ``dyn`` == nullptr: the step-dependent arguments are host values (the classic form);
    // ``dyn`` != nullptr: every kernel takes them from the device-resident counter w.dstep, so the SAME sequence serves every step and
    // can be replayed from a hipGraph (one graph launch instead of 74 kernel launches per step: at one page — R = 160 rows — the loop
    // was bound by launch cost, not by its kernels).  Both forms run the same kernels on the same operands: identical results.
    // rows_path: the few-row form of a step (see body).  mit_ocr48_decode_rows_max_set: largest R = 5 N it is used for; measured equal to
    // the 64 x 64 split tiles at R = 2560 (16 pages) and 2-3.5x faster per Linear at R = 160 .. 640 (profiles/r04u_pgemm_rows.log)
    const int rows_max = rows_max_now();
    static const bool splitk_env = getenv("MIT_OCR_SPLITK") != nullptr && atoi(getenv("MIT_OCR_SPLITK")) != 0;
    const int gmode = mit_gemm_mode_get();
    bool rows_path = (gmode == 6 || gmode == 9) && R <= rows_max && !splitk_env && rows_ok(dec->pred1) && dec->pred.w_split &&
                     dec->pred.Kp == dec->pred.K && (dec->pred.K % 16) == 0 && (dec->pred.N % 4) == 0;
    for (int l = 0; l < 5 && rows_path; ++l) {
        const MitOcrDecoderLayer &ly = dec->layers[l];
        rows_path = rows_ok(ly.qkv) && rows_ok(ly.out) && rows_ok(ly.q2) && rows_ok(ly.out2) && rows_ok(ly.ff1) && rows_ok(ly.ff2);
    }
    // LayerNorm inside the Linear that consumes it (read per call: tests switch it in-process; MIT_OCR_LN_FUSED=0 = the two-launch form)
    const char *lnf_env = getenv("MIT_OCR_LN_FUSED");
    const bool ln_fused = rows_path && !(lnf_env && *lnf_env && atoi(lnf_env) == 0);
    // The FFN's second Linear (K = 2048) with K cut across four waves: for FEW rows only (a page or two, where its 10 us accumulator chain
    // is a sixth of the step) — its sums round differently from the k-sequential chain every other form uses, so a page decoded alone
    // differs from the same page inside a large group in the last bits of the log-probabilities (tests: tokens equal, 1e-5 on the
    // probabilities).  MIT_OCR_FF2_SPLITK=0 keeps the one-chain kernel (bit for bit the tiled form); read per call.
    // norm2 + q projection inside the cross-attention kernel (MIT_OCR_Q2_FUSED=0 = separate launches; bit-identical either way; gemm mode 6 only)
    const char *q2_env = getenv("MIT_OCR_Q2_FUSED");
    const bool q2_fused = ln_fused && gmode == 6 && !(q2_env && *q2_env && atoi(q2_env) == 0);
    const char *sk_env = getenv("MIT_OCR_FF2_SPLITK");
    const int ff2_splitk = (rows_path && R <= FF2_SPLITK_MAX_ROWS && !(sk_env && *sk_env && atoi(sk_env) == 0)) ? 1 : 0;
    auto body = [&](const int step, const int *dyn, hipStream_t st) -> int {
        const int64_t so = dyn ? 0 : (int64_t)step * E;  // host-side step offset; the dyn form adds step * E on the device
        // (w.tgt holds the embedded tokens of this step: the start tokens before the loop, then written by the previous step's beam kernel)
        const int Tk = dyn ? T : step + 1;  // dyn: capacity (grid / LDS); the kernels stop at *dyn + 1
        if (rows_path) {
            // few rows (one page .. a group of pages): every Linear on the one-wave-per-block planar GEMM (pgemm_rows.h) — the LayerNorms,
            // the attention kernels and the ReLU / GELU epilogues hand over bf16 planes, the residual stream and the K / V caches stay
            // fp32.  Same kernels' arithmetic, same plane split, same MFMA order as the tiles of the other form: identical results.
            const int64_t Rp = w.Rp;
            const OcrPlanes nrm_pl{w.nrm_p, Rp, E / 8}, att_pl{w.att_p, Rp, E / 8};
            for (int l = 0; l < 5; ++l) {
                const MitOcrDecoderLayer &ly = dec->layers[l];
                float *qc = w.qkv + (int64_t)(l * 3 + 0) * R * TE;
                float *kc = w.qkv + (int64_t)(l * 3 + 1) * R * TE;
                float *vc = w.qkv + (int64_t)(l * 3 + 2) * R * TE;
                if (ln_fused) {
                    if (pgemm_ln(ly.qkv, w.tgt, E, ly.ln1_w, ly.ln1_b, R, qc + so, TE, MIT_ACT_NONE, nullptr, 0, st, E, (int64_t)R * TE, dyn, E)) return 1;
                } else {
                    if (ocrk_layernorm(w.tgt, E, ly.ln1_w, ly.ln1_b, nullptr, 0, R, E, 1e-5f, st, &nrm_pl)) return 1;
                    if (pgemm(ly.qkv, w.nrm_p, Rp, R, qc + so, TE, MIT_ACT_NONE, nullptr, 0, nullptr, 0, st, E, (int64_t)R * TE, dyn, E)) return 1;
                }
                OcrAttXpos xs{dec->xpos.cos_t, dec->xpos.sin_t, dec->xpos.scale_t, dec->xpos.iscale_t, dec->xpos.pmax, step, 1, E};
                ocrk_attention(qc + so, TE, E, kc, TE, E, vc, TE, E, nullptr, 0, 0, nullptr, R, 1, Tk, 1, st, 4, 80, dyn, &xs, &att_pl);
                if (pgemm(ly.out, w.att_p, Rp, R, w.tgt, E, MIT_ACT_NONE, w.tgt, E, nullptr, 0, st)) return 1;
                const float *mk = a->mem_k + (int64_t)l * N * L * E;
                const float *mv = a->mem_v + (int64_t)l * N * L * E;
                OcrAttXpos xc{dec->xpos.cos_t, dec->xpos.sin_t, dec->xpos.scale_t, dec->xpos.iscale_t, dec->xpos.pmax, step, 0, 0};
                // norm2 + the q projection inside the cross-attention kernel where that form exists (a page or a few; else two launches)
                bool q_inside = false;
                if (q2_fused && ly.q2.bias && ly.q2.N == E) {
                    const OcrAttQProj qp{w.tgt, E, ly.ln2_w, ly.ln2_b, 1e-5f, ly.q2.w_split, ly.q2.ldw, ly.q2.scale, ly.q2.bias};
                    q_inside = ocrk_cross_attention_qproj(qp, mk, (int64_t)L * E, E, mv, (int64_t)L * E, E, a->mem_len, R, L, st, dyn, &xc, &att_pl);
                }
                if (!q_inside) {
                    if (ln_fused) {
                        if (pgemm_ln(ly.q2, w.tgt, E, ly.ln2_w, ly.ln2_b, R, w.q2, E, MIT_ACT_NONE, nullptr, 0, st)) return 1;
                    } else {
                        if (ocrk_layernorm(w.tgt, E, ly.ln2_w, ly.ln2_b, nullptr, 0, R, E, 1e-5f, st, &nrm_pl)) return 1;
                        if (pgemm(ly.q2, w.nrm_p, Rp, R, w.q2, E, MIT_ACT_NONE, nullptr, 0, nullptr, 0, st)) return 1;
                    }
                    ocrk_attention(w.q2, E, E, mk, (int64_t)L * E, E, mv, (int64_t)L * E, E, nullptr, 0, 0, a->mem_len, R, 1, L, 5, st, 4, 80, dyn, &xc, &att_pl);
                }
                if (pgemm(ly.out2, w.att_p, Rp, R, w.tgt, E, MIT_ACT_NONE, w.tgt, E, nullptr, 0, st)) return 1;
                if (ln_fused) {
                    if (pgemm_ln(ly.ff1, w.tgt, E, ly.ln3_w, ly.ln3_b, R, nullptr, 0, MIT_ACT_RELU, w.ffh_p, Rp, st)) return 1;
                } else {
                    if (ocrk_layernorm(w.tgt, E, ly.ln3_w, ly.ln3_b, nullptr, 0, R, E, 1e-5f, st, &nrm_pl)) return 1;
                    if (pgemm(ly.ff1, w.nrm_p, Rp, R, nullptr, 0, MIT_ACT_RELU, nullptr, 0, w.ffh_p, Rp, st)) return 1;
                }
                if (l < 4) {
                    if (pgemm(ly.ff2, w.ffh_p, Rp, R, w.tgt, E, MIT_ACT_NONE, w.tgt, E, nullptr, 0, st, 0, 0, nullptr, 0, ff2_splitk)) return 1;
                } else {  // last layer: the step's output into the activation cache (:570), and as planes for the prediction head
                    if (pgemm(ly.ff2, w.ffh_p, Rp, R, w.decoded + so, TE, MIT_ACT_NONE, w.tgt, E, w.dec_p, Rp, st, 0, 0, dyn, E, ff2_splitk)) return 1;
                }
            }
            if (pgemm(dec->pred1, w.dec_p, Rp, R, nullptr, 0, MIT_ACT_GELU, nullptr, 0, w.p1_p, Rp, st)) return 1;
            if (pgemm(dec->pred, w.p1_p, Rp, R, w.logits, Dp, MIT_ACT_NONE, nullptr, 0, nullptr, 0, st)) return 1;
        } else {
            for (int l = 0; l < 5; ++l) {
                const MitOcrDecoderLayer &ly = dec->layers[l];
                float *qc = w.qkv + (int64_t)(l * 3 + 0) * R * TE;
                float *kc = w.qkv + (int64_t)(l * 3 + 1) * R * TE;
                float *vc = w.qkv + (int64_t)(l * 3 + 2) * R * TE;
                // self attention (:565)
                if (ocrk_layernorm(w.tgt, E, ly.ln1_w, ly.ln1_b, w.nrm, E, R, E, 1e-5f, st)) return 1;
                if (gemm(ly.qkv, w.nrm, E, qc + so, TE, R, MIT_ACT_NONE, nullptr, 0, st, E, (int64_t)R * TE, nullptr, dyn, 0, E)) return 1;
                // (the XPOS rotation of the step's query and of the key history 0 .. step happens inside the attention kernel)
                OcrAttXpos xs{dec->xpos.cos_t, dec->xpos.sin_t, dec->xpos.scale_t, dec->xpos.iscale_t, dec->xpos.pmax, step, 1, E};
                ocrk_attention(qc + so, TE, E, kc, TE, E, vc, TE, E, w.att, E, E, nullptr, R, 1, Tk, 1, st, 4, 80, dyn, &xs);
                if (gemm(ly.out, w.att, E, w.tgt, E, R, MIT_ACT_NONE, w.tgt, E, st)) return 1;
                // cross attention (:567)
                if (ocrk_layernorm(w.tgt, E, ly.ln2_w, ly.ln2_b, w.nrm, E, R, E, 1e-5f, st)) return 1;
                if (gemm(ly.q2, w.nrm, E, w.q2, E, R, MIT_ACT_NONE, nullptr, 0, st)) return 1;
                const float *mk = a->mem_k + (int64_t)l * N * L * E;
                const float *mv = a->mem_v + (int64_t)l * N * L * E;
                OcrAttXpos xc{dec->xpos.cos_t, dec->xpos.sin_t, dec->xpos.scale_t, dec->xpos.iscale_t, dec->xpos.pmax, step, 0, 0};
                ocrk_attention(w.q2, E, E, mk, (int64_t)L * E, E, mv, (int64_t)L * E, E, w.att, E, E, a->mem_len, R, 1, L, 5, st, 4, 80, dyn, &xc);
                if (gemm(ly.out2, w.att, E, w.tgt, E, R, MIT_ACT_NONE, w.tgt, E, st)) return 1;
                // feed forward (:568)
                if (ocrk_layernorm(w.tgt, E, ly.ln3_w, ly.ln3_b, w.nrm, E, R, E, 1e-5f, st)) return 1;
                if (gemm(ly.ff1, w.nrm, E, w.ffh, FF, R, MIT_ACT_RELU, nullptr, 0, st)) return 1;
                if (l < 4) {
                    if (gemm(ly.ff2, w.ffh, FF, w.tgt, E, R, MIT_ACT_NONE, w.tgt, E, st, 0, 0, w.part)) return 1;
                } else {  // last layer writes the step's output straight into the activation cache (:570)
                    if (gemm(ly.ff2, w.ffh, FF, w.decoded + so, TE, R, MIT_ACT_NONE, w.tgt, E, st, 0, 0, dyn ? nullptr : w.part, dyn, 0, E)) return 1;
                }
            }
            if (gemm(dec->pred1, w.decoded + so, TE, w.p1, E, R, MIT_ACT_GELU, nullptr, 0, st, 0, 0, nullptr, dyn, E, 0)) return 1;
            if (gemm(dec->pred, w.p1, E, w.logits, Dp, R, MIT_ACT_NONE, nullptr, 0, st)) return 1;
        }
        if (a->trace_logits)
            MIT_CHECK_HIP(hipMemcpy2DAsync(a->trace_logits + (int64_t)step * R * D, (size_t)D * 4, w.logits, (size_t)Dp * 4,
                                           (size_t)D * 4, R, hipMemcpyDeviceToDevice, st));
        ocrk_logsoftmax_top5(w.logits, Dp, R, D, a->suppress_eos ? a->end_tok : -1, w.vals, w.idx, nullptr, st);
        if (dyn) {
            ocrk_beam_dyn(w.vals, w.idx, hist[0], hist[1], hist_ld, logp[0], logp[1], w.done, a->res_row, a->res_len, a->res_prob, a->res_tok,
                          w.done_count, N, dyn, a->start_tok, a->end_tok, a->max_finished, st, dec->embd, w.tgt, E);
            ocrk_step_advance(w.dstep, st);
        } else if (step == 0) {
            ocrk_beam_init(w.vals, w.idx, hist[cur], hist_ld, logp[cur], N, a->start_tok, st, dec->embd, w.tgt, E);
        } else {
            ocrk_beam_step(w.vals, w.idx, hist[cur], hist[cur ^ 1], hist_ld, logp[cur], logp[cur ^ 1], w.done, a->res_row,
                           a->res_len, a->res_prob, a->res_tok, w.done_count, N, step, a->end_tok, a->max_finished, st, dec->embd, w.tgt, E);
            cur ^= 1;
        }
        if (a->trace_hist)
            MIT_CHECK_HIP(hipMemcpyAsync(a->trace_hist + (int64_t)step * R * hist_ld, hist[cur], (size_t)R * hist_ld * 4,
                                         hipMemcpyDeviceToDevice, st));
        MIT_CHECK_LAUNCH("mit_ocr48_decode");
        return 0;
    };

    // Graph replay is OPT-IN (graph_mode = 1 or MIT_OCR_DECODE_GRAPH=1).  Measured on one page (32 lines, R = 160 rows, 32 steps): 50.2 ms
    // with the graph, 50.6 ms launch by launch — the loop is bound by its kernels' own latency (a 64 x 64 GEMM tile of K = 320 takes 12 us
    // for 20 dependent K-steps whatever launches it; rocprofv3: 46.5 ms of kernel time per call), not by the launches.  Never while
    // tracing (per-step copies at host offsets) or probing (events around every launch).
    static const int graph_env = getenv("MIT_OCR_DECODE_GRAPH") ? atoi(getenv("MIT_OCR_DECODE_GRAPH")) : 0;
    const int gm = a->graph_mode == 1 ? 1 : a->graph_mode == 2 ? 0 : graph_env;   // the argument wins over the environment
    const bool use_graph = gm == 1 && !a->trace_logits && !a->trace_hist && !mit_probe_on() && T >= 4;
    hipGraphExec_t exec = nullptr;
    if (use_graph) {
        reap_graphs();
        static thread_local hipStream_t cap = nullptr;
        if (!cap) MIT_CHECK_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        MIT_CHECK_HIP(hipMemsetAsync(w.dstep, 0, 4, s));
        hipGraph_t graph = nullptr;
        MIT_CHECK_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
        const int rc = body(0, w.dstep, cap);
        const hipError_t ce = hipStreamEndCapture(cap, &graph);
        if (rc || ce != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc ? rc : mit_set_error("mit_ocr48_decode: stream capture of a decode step failed: %s", hipGetErrorString(ce));
        }
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            (void)hipGraphDestroy(graph);
            return mit_set_error("mit_ocr48_decode: hipGraphInstantiate failed: %s", hipGetErrorString(ie));
        }
        g_pending.push_back({exec, graph, nullptr});
    }
    for (int step = 0; step < T; ++step) {
        if (use_graph) {
            MIT_CHECK_HIP(hipGraphLaunch(exec, s));
        } else if (body(step, nullptr, s)) {
            return 1;
        }
        steps = step + 1;
        if (!a->suppress_eos && step >= 1 && (step % 4 == 3) && step + 1 < T) {  // early exit (:765-766) without a per-step sync;
            // with EOS suppressed no hypothesis can finish, so the loop stays fully asynchronous
            int dc = 0;
            MIT_CHECK_HIP(hipMemcpyAsync(&dc, w.done_count, 4, hipMemcpyDeviceToHost, s));
            MIT_CHECK_HIP(hipStreamSynchronize(s));
            if (dc >= N) break;
        }
    }
    if (use_graph) {  // the graph stays alive until its launches have run: an event marks that point, the next call reaps it
        cur = steps <= 1 ? 0 : ((steps - 1) & 1);  // the buffer the last executed step wrote (step 0 and 1 -> hist[0] -> hist[1] ...)
        hipEvent_t ev = nullptr;
        MIT_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        MIT_CHECK_HIP(hipEventRecord(ev, s));
        g_pending.back().done = ev;
    }
    ocrk_beam_finalize(hist[cur], hist_ld, logp[cur], w.done, a->res_row, a->res_len, a->res_prob, a->res_tok, N, steps + 1, s);
    // colour heads over every beam row's activation cache (:789-799); the caller gathers rows res_row[n]
    if (gemm(dec->color1, w.decoded, E, w.cfeat, 64, R * T, MIT_ACT_RELU, nullptr, 0, s)) return 1;
    if (gemm(dec->color_heads, w.cfeat, 64, a->colors, 12, R * T, MIT_ACT_NONE, nullptr, 0, s)) return 1;
    MIT_CHECK_LAUNCH("mit_ocr48_decode");
    a->steps_run = steps;
    return 0;
}
