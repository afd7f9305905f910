// pgemm_rows_ln.hip — LayerNorm fused into the few-row planar GEMM: C = epilogue(LayerNorm(X) @ W) in ONE launch.
//
// The B = 1 decoder step (ocr_decoder.hip, rows path) is a chain of ≈ 60 dependent launches of 5-16 us; three per layer are LayerNorms
// (TransformerDecoderLayer norm1 / norm2 / norm3, manga_translator/ocr/model_48px.py:548-572 via nn.LayerNorm) whose only consumer is
// the Linear that follows.  A 32 x 32 output block needs WHOLE rows of A (K = E = 320), so the workgroup that owns a row block can
// normalise it itself: four waves = four column blocks of one 32-row block; wave w normalises rows 8 w .. 8 w + 7 (eight lanes per row,
// each holding five whole k-cells: 32-byte runs, coalesced), splits its cells into the three bf16 planes and parks them in LDS in the
// MFMA operand layout (65 KB); behind one barrier every wave's K loop is three ds_read_b128 and six MFMAs per step.  The 15 LayerNorm
// launches of a step disappear.
// (Measured on the way, per launch at 160 rows against 5.7 + 5.7 us for the two launches: one wave per block with the row halves in
// registers — 40 strided 16-byte loads per lane, every column block redoing the statistics — 11.7 us; four waves sharing fp32 rows
// in LDS, each splitting its own operand cells in the K loop — 8.9 us; this form: profiles/r11*.)
//
// Bit-exactness: layernorm_kernel<true> (ocr_kernels.hip) gives lane l of its wave the elements d = l, l + 64, ... (summed in that
// order from 0) and reduces the 64 partials by an xor butterfly (32, 16, ..., 1).  Here lane (rr, q) of a wave holds, for row 8 w + rr,
// the partials l = 8 q + j (j < 8): butterfly levels 32, 16, 8 pair q with q ^ 4, q ^ 2, q ^ 1 (the row's eight lanes: ds_swizzle /
// DPP quad permutes), levels 4, 2, 1 pair j with j ^ 4, j ^ 2, j ^ 1 (registers).  Same additions in the same tree => the mean, the variance, the normalised values,
// hence the planes (bf16_split.h, the shared split) and the product are those of layernorm_kernel + pgemm_rows_kernel, bit for bit
// (tests/test_ocr_gpu.py::test_layernorm_inside_the_few_row_gemm_is_bit_identical).
//
// Compiled WITHOUT packed-fp32 code generation (build.py: not in PACKED_FP32_BY_DESIGN) — the statistics are scalar fp32 chains.
#include "conv_gemm_kernels.h"
#include "pgemm_rows.h"
#include "pgemm_rows_epi.h"
#include "ln_rows8.h"
#include "common.h"

using namespace mitcg;

namespace {

constexpr int LN_K = mitln::LN_K, LN_KTS = LN_K / 16;

constexpr int LN_CP = 34;   // cells per (plane, k-cell) slab in LDS: 32 rows + 2 of padding (the 16 lanes of a write pass fall on 16 distinct 16-byte slots)

template <int NPROD, int D>
// (leading scalar arguments: what the first requests — W cells, the rows — need; preloaded into SGPRs with the wave, see build.py)
__global__ __launch_bounds__(256) void pgemm_rows_ln_kernel(const uint16_t *pw, const float *lnx, const unsigned int ldw_u, const int ldx_i, const int Mrows,
                                                            const int MT, const int NT4, const int NT, const float *lnw, const float *lnb,
                                                            const MitPGemm p, const PgRowsExt x, const float eps) {
    constexpr int K8 = LN_K >> 3;
    __shared__ __attribute__((aligned(16))) u32x4 apl[3 * K8 * LN_CP];   // the block's normalised rows as planes: cell (pl, k8, row) at (pl K8 + k8) LN_CP + row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // block -> (group of four column blocks, row block): the row blocks of a column group (same W cells) on one XCD
    const int total = MT * NT4, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int nt4 = t / MT, mt = t - nt4 * MT;
    const int nt = nt4 * 4 + wave;
    const bool has_tile = nt < NT;
    const int m0 = mt * 32, n0 = nt * 32;
    const unsigned int w_step = ldw_u * 32u;                               // bytes per k step (two k cells)
    const unsigned int w_plane = (unsigned int)K8 * ldw_u * 16u;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(pw), 0, 3 * w_plane, 0x00020000);
    const unsigned int w_off = ((unsigned int)lh * ldw_u + (unsigned int)(n0 + li)) * 16u;   // (past the planes: the descriptor answers 0)

    u32x4 fw[D][3];
    auto issue = [&](const int d, const int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fw[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off, pl * w_plane + (unsigned int)ks * w_step, 0);
    };
#pragma unroll
    for (int d = 0; d < D && d < LN_KTS; ++d) issue(d, d);   // the first W cells travel while the rows are normalised
    __builtin_amdgcn_sched_barrier(0);

    {   // ---- rows 8 wave .. + 7 of the block: lane (rr, q) holds the cells k8 = 8 b + q, i.e. d = 64 b + 8 q + j.  Rows past M repeat
        // row M - 1 (normalised, never stored).
        const int rr = lane >> 3, q = lane & 7;
        const int row = min(m0 + 8 * wave + rr, Mrows - 1);
        f32x4 v[5][2];
        mitln::ln_row_cells(lnx + (int64_t)row * ldx_i + 8 * q, lnw + 8 * q, lnb + 8 * q, eps, v);   // (ln_rows8.h: layernorm_kernel's bits)
        u32x4 *dst = apl + q * LN_CP + 8 * wave + rr;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            u32x4 h, m, l;
            split8(v[b][0], v[b][1], h, m, l);
            dst[(8 * b) * LN_CP] = h;
            dst[(K8 + 8 * b) * LN_CP] = m;
            dst[(2 * K8 + 8 * b) * LN_CP] = l;
        }
    }
    __syncthreads();
    if (!has_tile) return;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const u32x4 *ya = apl + lh * LN_CP + li;   // this lane's operand cells of step ks: (pl, 2 ks + lh, li)
    // the A cells of step ks + 1 leave LDS while the six MFMAs of step ks run; the W cells stay D steps ahead (the fences keep both where
    // they are written: left alone the scheduler sinks every load to its use, which empties the ring — 14.5 us per launch instead of 7)
    u32x4 fa[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fa[0][pl] = ya[(pl * K8) * LN_CP];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < LN_KTS; ++ks) {
        const int d = ks % D, cur = ks & 1;
        if (ks + 1 < LN_KTS) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[cur ^ 1][pl] = ya[(pl * K8 + 2 * (ks + 1)) * LN_CP];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 9 - NPROD; pr < 9; ++pr)   // transposed result (rows = output columns), as pgemm_rows_kernel
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[d][kSplitPB[pr]]), __builtin_bit_cast(bf16x8, fa[cur][kSplitPA[pr]]), acc, 0, 0, 0);
        if (ks + D < LN_KTS) issue(d, ks + D);
        __builtin_amdgcn_sched_barrier(0);
    }
    pg_rows_epilogue(p, x, acc, m0, n0, 0, lane);
}

template <int NPROD>
void launch(const MitPGemm &p, const PgRowsExt &x, const PgRowsLn &ln, hipStream_t s) {
    const int MT = (p.M + 31) / 32, NT = (p.N + 31) / 32, NT4 = (NT + 3) / 4;
    const int total = MT * NT4, per = (total + 7) / 8;
    hipLaunchKernelGGL((pgemm_rows_ln_kernel<NPROD, PG_ROWS_DEPTH>), dim3(per * 8), dim3(256), 0, s, p.w_planes, ln.x, (unsigned int)p.ldw, (int)ln.ldx, p.M, MT, NT4, NT, ln.w, ln.b,
                       p, x, ln.eps);
}

}  // namespace

int mit_pgemm_rows_ln(const MitPGemm &d, const PgRowsExt &x, const PgRowsLn &ln, hipStream_t s) {
    MitPGemm p = d;
    if (!p.w_planes || !ln.x || !ln.w || !ln.b || p.M <= 0 || p.N <= 0) return mit_set_error("mit_pgemm_rows_ln: bad operands (M=%d N=%d)", p.M, p.N);
    if (p.K != LN_K) return mit_set_error("mit_pgemm_rows_ln: K must be %d (got %d)", LN_K, p.K);
    if (p.Z > 1 || p.pre) return mit_set_error("mit_pgemm_rows_ln: no batch, no pre operand");
    uint16_t *planes = p.c_planes ? p.c_planes : x.also_planes;
    if (!p.c && !planes) return mit_set_error("mit_pgemm_rows_ln: no output");
    if (p.c_planes && x.also_planes) return mit_set_error("mit_pgemm_rows_ln: two planar outputs");
    if ((p.N & 3) || (planes && (p.N & 7)) || (p.ldc & 3) || (p.ld_post & 3) || (ln.ldx & 3) || (x.nsplit & 7) || (x.nhi & 3) || (x.c_dyn & 3))
        return mit_set_error("mit_pgemm_rows_ln: N / strides must keep 16-byte cells whole");
    if ((uint64_t)3 * (LN_K / 8) * (uint64_t)p.ldw * 16u >= (1ull << 32)) return mit_set_error("mit_pgemm_rows_ln: W planes exceed 4 GB");
    auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    if (!al16(p.w_planes) || !al16(ln.x) || !al16(ln.w) || !al16(ln.b) || !al16(p.c) || !al16(planes) || !al16(p.scale) || !al16(p.bias) || !al16(p.post))
        return mit_set_error("mit_pgemm_rows_ln: operands, outputs, LayerNorm weights, scale / bias and post must be 16-byte aligned");
    const int a = p.act & 0xff;
    if (a != MIT_ACT_NONE && a != MIT_ACT_RELU && a != MIT_ACT_GELU) return mit_set_error("mit_pgemm_rows_ln: activation %d", p.act);
    if (p.nprod == 0) p.nprod = mit_gemm_mode_get();
    p.Z = 1;
    const double bytes = 4.0 * p.M * LN_K + 6.0 * (double)LN_K * p.N + (p.c ? 4.0 : 0.0) * p.M * p.N + (planes ? 6.0 : 0.0) * p.M * p.N;
    MitProbeScope probe("pgemm_rows_ln_kernel", s, bytes, 2.0 * p.M * (double)p.N * LN_K);
    if (p.nprod == 6) launch<6>(p, x, ln, s);
    else if (p.nprod == 9) launch<9>(p, x, ln, s);
    else return mit_set_error("mit_pgemm_rows_ln: nprod must be 6 or 9 (got %d)", p.nprod);
    MIT_CHECK_LAUNCH("mit_pgemm_rows_ln");
    return 0;
}
