// pgemm_rows_ln.hip — LayerNorm fused into the few-row planar GEMM: C = epilogue(LayerNorm(X) @ W) in ONE launch.
//
// The B = 1 decoder step (ocr_decoder.hip, rows path) is a chain of ≈ 60 dependent launches of 5-16 us; three per layer are LayerNorms
// (TransformerDecoderLayer norm1 / norm2 / norm3, manga_translator/ocr/model_48px.py:548-572 via nn.LayerNorm) whose only consumer is
// the Linear that follows.  A wave of pgemm_rows_kernel owns a 32 x 32 block of the output and needs WHOLE rows of A (K = E = 320), so
// it can normalise its 32 rows itself: lane (li, lh) loads the halves k = 16 ks + 8 lh .. + 7 of row m0 + li as fp32 (160 registers),
// the two lanes of a row exchange eight partial sums per moment, and every k step splits its freshly normalised cell into the three
// bf16 planes right before the MFMAs.  Every column block redoes the (tiny) statistics of its rows; the 15 LayerNorm launches of a step
// disappear.
//
// Bit-exactness: layernorm_kernel<true> (ocr_kernels.hip) gives lane l of its wave the elements d = l, l + 64, ... (summed in that
// order from 0) and reduces the 64 partials by an xor butterfly (32, 16, ..., 1).  Element d lives here on lane (li, (d >> 3) & 1), so
// the partials l = 16 a + 8 lh + j (a < 4, j < 8) are all in this lane; butterfly levels 32 and 16 pair a with a ^ 2 and a ^ 1 (same
// lane), level 8 pairs the two lanes of the row (one cross-lane exchange per j), levels 4, 2, 1 pair j with j ^ 4, j ^ 2, j ^ 1 (same
// lane).  Same additions in the same tree => the mean, the variance, the normalised values, hence the planes (bf16_split.h, the shared
// split) and the product are those of layernorm_kernel + pgemm_rows_kernel, bit for bit.  tests/test_pgemm_gpu.py checks exactly that.
//
// Compiled WITHOUT packed-fp32 code generation (build.py: not in PACKED_FP32_BY_DESIGN) — the statistics are scalar fp32 chains.
#include "conv_gemm_kernels.h"
#include "pgemm_rows.h"
#include "pgemm_rows_epi.h"
#include "common.h"

using namespace mitcg;

namespace {

constexpr int LN_K = 320, LN_KTS = LN_K / 16;

template <int NPROD, int D>
__global__ __launch_bounds__(64) void pgemm_rows_ln_kernel(const MitPGemm p, const PgRowsExt x, const PgRowsLn ln, const int MT, const int NT) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    // block -> (column block, row block) as pgemm_rows_kernel: the row blocks of a column block (same W cells) on one XCD
    const int total = MT * NT, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int nt = t / MT, mt = t - nt * MT;
    const int m0 = mt * 32, n0 = nt * 32;
    constexpr int K8 = LN_K >> 3;
    const unsigned int w_step = (unsigned int)p.ldw * 32u;                               // bytes per k step (two k cells)
    const unsigned int w_plane = (unsigned int)K8 * (unsigned int)p.ldw * 16u;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.w_planes), 0, 3 * w_plane, 0x00020000);
    const unsigned int w_off = ((unsigned int)lh * (unsigned int)p.ldw + (unsigned int)(n0 + li)) * 16u;

    u32x4 fw[D][3];
    auto issue = [&](const int d, const int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fw[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off, pl * w_plane + (unsigned int)ks * w_step, 0);
    };
#pragma unroll
    for (int d = 0; d < D && d < LN_KTS; ++d) issue(d, d);   // the first W cells travel while the rows are normalised

    // ---- this lane's half rows: k = 16 ks + 8 lh + j.  Rows past M repeat row M - 1 (loaded, normalised, never stored).
    const int row = min(m0 + li, p.M - 1);
    const float *xr = ln.x + (int64_t)row * ln.ldx + 8 * lh;
    f32x4 v[LN_KTS][2];
#pragma unroll
    for (int ks = 0; ks < LN_KTS; ++ks) {
        v[ks][0] = *reinterpret_cast<const f32x4 *>(xr + 16 * ks);
        v[ks][1] = *reinterpret_cast<const f32x4 *>(xr + 16 * ks + 4);
    }
    // butterfly of layernorm_kernel over the 64 partials, of which this lane holds l = 16 a + 8 lh + j
    auto reduce = [&](float (&pt)[4][8]) __attribute__((always_inline)) -> float {
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            q[j] = (pt[0][j] + pt[2][j]) + (pt[1][j] + pt[3][j]);   // levels 32 (a ^ 2) and 16 (a ^ 1)
            q[j] += __shfl_xor(q[j], 32);                           // level 8: the row's other lane
        }
        const float r0 = q[0] + q[4], r1 = q[1] + q[5], r2 = q[2] + q[6], r3 = q[3] + q[7];   // level 4
        return (r0 + r2) + (r1 + r3);                                                         // levels 2 and 1
    };
    float pt[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < LN_KTS / 4; ++b) s += v[a + 4 * b][j >> 2][j & 3];
            pt[a][j] = s;
        }
    const float mean = reduce(pt) / (float)LN_K;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < LN_KTS / 4; ++b) {
                const float tt = v[a + 4 * b][j >> 2][j & 3] - mean;
                s += tt * tt;
            }
            pt[a][j] = s;
        }
    const float rstd = 1.0f / sqrtf(reduce(pt) / (float)LN_K + ln.eps);
    const float *gw = ln.w + 8 * lh, *gb = ln.b + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < LN_KTS; ++ks) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(gw + 16 * ks + 4 * hf), bv = *reinterpret_cast<const f32x4 *>(gb + 16 * ks + 4 * hf);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[ks][hf][c] = (v[ks][hf][c] - mean) * rstd * wv[c] + bv[c];
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < LN_KTS; ++ks) {
        u32x4 fa[3];
        split8(v[ks][0], v[ks][1], fa[0], fa[1], fa[2]);
        const int d = ks % D;
#pragma unroll
        for (int pr = 9 - NPROD; pr < 9; ++pr)   // transposed result (rows = output columns), as pgemm_rows_kernel
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[d][kSplitPB[pr]]), __builtin_bit_cast(bf16x8, fa[kSplitPA[pr]]), acc, 0, 0, 0);
        if (ks + D < LN_KTS) issue(d, ks + D);
        __builtin_amdgcn_sched_barrier(0);
    }
    pg_rows_epilogue(p, x, acc, m0, n0, 0, lane);
}

template <int NPROD>
void launch(const MitPGemm &p, const PgRowsExt &x, const PgRowsLn &ln, hipStream_t s) {
    const int MT = (p.M + 31) / 32, NT = (p.N + 31) / 32;
    const int total = MT * NT, per = (total + 7) / 8;
    hipLaunchKernelGGL((pgemm_rows_ln_kernel<NPROD, 6>), dim3(per * 8), dim3(64), 0, s, p, x, ln, MT, NT);
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
