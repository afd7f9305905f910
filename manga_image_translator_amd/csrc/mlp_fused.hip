// mlp_fused.hip — the ConvNeXt block's pointwise pair as ONE kernel:  y = x_res + gamma * (W2 · gelu(W1 · t + b1) + b2)
// (ConvNeXtBlock.forward, manga_translator/ocr/model_48px.py:203-214: pwconv1 -> GELU -> pwconv2 -> gamma -> + input), split-bf16 p6
// arithmetic (conv_gemm_split.h) on v_mfma_f32_32x32x16_bf16.  The 4C-wide hidden activations never leave the registers.
//
// Why a kernel of its own: as two launches the C = 80 stage writes and re-reads [M, 320] fp32 (3.7 GB + 3.7 GB per 16-page group) for
// 2 x 148 GFLOP — pwconv1 runs at 0.18 of the bf16 MFMA peak (a 5-step K loop behind a GELU epilogue and a 64 KB tile store), pwconv2
// at 0.24-0.27.  Here a wave owns 32 pixels and walks the hidden dimension in blocks of 32:
//
//   GEMM 1 (transposed):  Ht[32 hidden, 32 pixels] = W1t[32 hidden, C] · Xt[C, 32 pixels]
//       A operand = W1 planes (the cells mit_gemm_split_pack makes for pwconv1: [plane][k / 8][n = hidden][8 k]), B operand = the wave's
//       X rows, loaded once (lane = pixel, k-group = lane >> 5: 8 consecutive channels = 32 contiguous bytes), split once, kept in
//       registers for all hidden blocks.  The MFMA's C layout then has lane = PIXEL and registers = hidden index
//       (r -> (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) — which is an A-operand layout for the second GEMM: a lane's eight values
//       r = 8 s .. 8 s + 7 are "8 consecutive k" of MFMA step s, if the B operand lists its rows in the same order.
//   bias + GELU + split into three bf16 planes: in registers (same gelu_fast, same split as the tiles').
//   GEMM 2:  Y[32 pixels, C] += H[32 pixels, 32 hidden] · W2p[32 hidden, C], W2p = pwconv2's weight with its ROWS PERMUTED to that
//       register order (k' = 32 hb + 16 s + 8 lh + j  <->  hidden 32 hb + (j & 3) + 8 (2 s + (j >> 2)) + 4 lh), packed into planes once.
//
// Weights of a hidden block (W1: 3 x C/8 x 32 cells, W2p: 3 x 4 x NPAD cells, 33 KB at C = 80) go global -> registers -> LDS, double
// buffered, one barrier per hidden block (66 MFMAs per wave between barriers; the tiles have 24).  Plane pairs and k order are the
// tiles': GEMM 1 adds exactly the products pwconv1's tile adds, in the same order (the operands' roles are swapped, a product is
// commutative) — H is bit-identical to the two-launch form; GEMM 2 sees the same 16 hidden values per MFMA step in other k slots, so
// its fp32 sums may differ in the last bits from the two-launch form (same error bound; parity tolerances unchanged, and the result
// does not depend on batch or grid: this kernel is the only form the layer takes in GEMM mode 6).
#include "conv_gemm_kernels.h"

namespace {
using namespace mitcg;

// ABL (timing ablations, WRONG results, MIT_MLP_ABLATE=<n> in the environment of a MIT_CONV_EXPERIMENTS build only): 1 = no GELU (bias add
// only), 2 = no weight staging after the first block, 4 = no second contraction, 8 = no first contraction
template <int C, int ABL = 0>
__global__ __launch_bounds__(256, 2) void convnext_mlp_kernel(const float *__restrict__ x, const int64_t ldx, const u32x4 *__restrict__ w1p,
                                                              const float *__restrict__ b1, const u32x4 *__restrict__ w2p, const int ldn2,
                                                              const float *__restrict__ scale2, const float *__restrict__ bias2,
                                                              const float *post, const int64_t ldp, float *out, const int64_t ldo, const int M) {
    static_assert(C % 16 == 0 && C <= 96, "stage width");
    constexpr int KS1 = C / 16;        // MFMA steps of GEMM 1
    constexpr int K81 = C / 8;         // k cells of W1 per plane
    constexpr int HID = 4 * C;
    constexpr int NHB = HID / 32;      // hidden blocks
    constexpr int NB = (C + 31) / 32;  // 32-column blocks of the output
    constexpr int NPAD = NB * 32;
    constexpr int W1_CELLS = 3 * K81 * 32, W2_CELLS = 3 * 4 * NPAD, BUF_CELLS = W1_CELLS + W2_CELLS;
    constexpr int W2_LOAD = 3 * 4 * C;  // the W2p cells that exist (columns n < C); the padded columns of the LDS block stay zero
    constexpr int W1_IT = (W1_CELLS + 255) / 256, W2_IT = (W2_LOAD + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4 *wbuf = reinterpret_cast<u32x4 *>(smem);                 // [2][BUF_CELLS]: W1 block [3][K81][32], then W2p block [3][4][NPAD]
    float *b1s = reinterpret_cast<float *>(wbuf + 2 * BUF_CELLS);  // [HID]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int row0 = (blockIdx.x * 4 + wave) * 32;  // the wave's 32 pixels

    for (int i = tid; i < HID; i += 256) b1s[i] = b1[i];

    u32x4 st1[W1_IT], st2[W2_IT];
    auto load_w1 = [&](const int hb) {
#pragma unroll
        for (int it = 0; it < W1_IT; ++it) {
            const int i = tid + it * 256;
            const int pl = i / (K81 * 32), rem = i - pl * (K81 * 32), k8 = rem >> 5, n = rem & 31;
            st1[it] = w1p[i < W1_CELLS ? (int64_t)(pl * K81 + k8) * HID + 32 * hb + n : 0];
        }
    };
    auto store_w1 = [&](const int buf) {
#pragma unroll
        for (int it = 0; it < W1_IT; ++it) {
            const int i = tid + it * 256;
            if (i < W1_CELLS) wbuf[buf * BUF_CELLS + i] = st1[it];
        }
    };
    auto load_w2 = [&](const int hb) {
#pragma unroll
        for (int it = 0; it < W2_IT; ++it) {
            const int i = tid + it * 256;
            const int pk = i / C, n = i - pk * C;  // pk = plane * 4 + k cell of the block
            st2[it] = w2p[i < W2_LOAD ? (int64_t)((pk >> 2) * (HID / 8) + 4 * hb + (pk & 3)) * ldn2 + n : 0];
        }
    };
    auto store_w2 = [&](const int buf) {
#pragma unroll
        for (int it = 0; it < W2_IT; ++it) {
            const int i = tid + it * 256;
            const int pk = i / C, n = i - pk * C;
            if (i < W2_LOAD) wbuf[buf * BUF_CELLS + W1_CELLS + pk * NPAD + n] = st2[it];
        }
    };

    f32x16 acc2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;

    load_w1(0);  // the first block's weights travel while the X tile is brought in
    load_w2(0);
    // This workgroup's 128 X rows come in as whole 320-byte rows (coalesced float4 runs), pass through LDS — the area the weight blocks
    // use afterwards — and leave as each wave's B-operand fragments: lane = pixel, k-group = lane >> 5, 8 consecutive channels per MFMA
    // step, split into three planes once and kept in registers for all hidden blocks.
    u32x4 xh[KS1], xm[KS1], xl[KS1];
    {
        constexpr int XP = C + 4;  // row pitch in floats
        static_assert(128 * XP * 4 <= 2 * BUF_CELLS * 16, "X tile fits the weight buffers");
        float *xs = smem;
        constexpr int XQ = 128 * (C / 4), XIT = (XQ + 255) / 256;
        f32x4 xv[XIT];
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int q = tid + 256 * it, row = q / (C / 4), c4 = q - row * (C / 4);
            const int grow = blockIdx.x * 128 + row < M ? blockIdx.x * 128 + row : M - 1;
            xv[it] = *reinterpret_cast<const f32x4 *>(x + (int64_t)grow * ldx + (q < XQ ? c4 : 0) * 4);
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int q = tid + 256 * it, row = q / (C / 4), c4 = q - row * (C / 4);
            if (q < XQ) *reinterpret_cast<f32x4 *>(xs + row * XP + c4 * 4) = xv[it];
        }
        __syncthreads();
        const float *xr = xs + (wave * 32 + li) * XP + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(xr + 16 * ks), v1 = *reinterpret_cast<const f32x4 *>(xr + 16 * ks + 4);
            split8(v0, v1, xh[ks], xm[ks], xl[ks]);
        }
        __syncthreads();  // every wave has its fragments: the area becomes the weight buffers
    }
    if (NPAD > C)  // padded output columns multiply zeros: written once, in both buffers
        for (int i = tid; i < 2 * 12 * (NPAD - C); i += 256) {
            const int buf = i / (12 * (NPAD - C)), rem = i - buf * (12 * (NPAD - C)), pk = rem / (NPAD - C), n = C + rem - pk * (NPAD - C);
            wbuf[buf * BUF_CELLS + W1_CELLS + pk * NPAD + n] = u32x4{0u, 0u, 0u, 0u};
        }
    store_w1(0);
    store_w2(0);
    __syncthreads();

#pragma unroll 1
    for (int hb = 0; hb < NHB; ++hb) {
        const int cur = hb & 1;
        const u32x4 *w1s = wbuf + cur * BUF_CELLS;
        const u32x4 *w2s = w1s + W1_CELLS;
        const bool more = hb + 1 < NHB && !(ABL & 2);
        if (more) {  // the next block's weights are requested now and parked in LDS at the end of this block: a whole block of latency cover
            load_w1(hb + 1);
            load_w2(hb + 1);
        }
        // ---- GEMM 1: Ht block = W1t · Xt ------------------------------------------------------------------------------
        // (fragment reads run one MFMA step ahead of their use: with two waves per SIMD nothing else hides the LDS latency)
        float bb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bb[r] = b1s[32 * hb + (r & 3) + 8 * (r >> 2) + 4 * lh];
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        u32x4 a[3], an[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl] = w1s[(pl * K81 + lh) * 32 + li];
#pragma unroll
        for (int ks = 0; ks < ((ABL & 8) ? 1 : KS1); ++ks) {
            if (ks + 1 < KS1) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) an[pl] = w1s[(pl * K81 + 2 * (ks + 1) + lh) * 32 + li];
            }
            const u32x4 xb[3] = {xh[ks], xm[ks], xl[ks]};
#pragma unroll
            for (int pr = 3; pr < 9; ++pr)  // the tiles' pair order; their A plane (activations) is this MFMA's B operand
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kSplitPB[pr]]), __builtin_bit_cast(bf16x8, xb[kSplitPA[pr]]), acc1, 0, 0, 0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = an[pl];
        }
        // the second contraction's first fragments travel while the VALU works on the activations
        u32x4 b[3], bn[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[pl] = w2s[(pl * 4 + lh) * NPAD + li];
        // ---- bias + GELU + split: the block's 16 hidden values of this lane's pixel become two A-operand fragments ------
        u32x4 hh[2], hm[2], hl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = acc1[8 * s + j] * 1.f + bb[8 * s + j];  // the tile's epilogue: acc * scale (none) + bias
                g[j] = (ABL & 1) ? v : gelu_fast(v);
            }
            split8(f32x4{g[0], g[1], g[2], g[3]}, f32x4{g[4], g[5], g[6], g[7]}, hh[s], hm[s], hl[s]);
        }
        // ---- GEMM 2: Y += H block · W2p block ------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < ((ABL & 4) ? 1 : 2 * NB); ++i) {
            const int s = i / NB, nb = i - s * NB;
            if (i + 1 < 2 * NB) {
                const int s1 = (i + 1) / NB, nb1 = (i + 1) - s1 * NB;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bn[pl] = w2s[(pl * 4 + 2 * s1 + lh) * NPAD + nb1 * 32 + li];
            }
            const u32x4 ha[3] = {hh[s], hm[s], hl[s]};
#pragma unroll
            for (int pr = 3; pr < 9; ++pr)
                acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ha[kSplitPA[pr]]), __builtin_bit_cast(bf16x8, b[kSplitPB[pr]]), acc2[nb], 0, 0, 0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[pl] = bn[pl];
        }
        if (more) {
            store_w1(cur ^ 1);
            store_w2(cur ^ 1);
        }
        __syncthreads();
    }

    // ---- epilogue: gamma (scale2), bias, residual ---------------------------------------------------------------------------
    // The accumulators (lane = column, registers = rows) pass through the wave's own slice of the now idle weight buffers and come
    // back as float4 runs of one row: 10 dwordx4 residual loads and 10 dwordx4 stores per lane, 320 contiguous bytes per row, instead
    // of 48 + 48 dword accesses.  All residual reads first, then all stores: `post` may be `out` itself, so the compiler keeps every
    // load behind the stores that precede it in program order — interleaved they become memory round trips in a row (33 us of a
    // 59 us workgroup in the first version of this kernel).  Same per-element arithmetic as the tiles' epilogue: acc * scale + bias, + post.
    constexpr int PITCH = C + 4;
    static_assert(32 * PITCH * 4 * 4 <= 2 * BUF_CELLS * 16, "transpose slices fit the weight buffers");
    float *tb = smem + wave * (32 * PITCH);
    // (the loop's last barrier is behind every wave's last weight read)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int c = nb * 32 + li;
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + c] = acc2[nb][r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-synchronous exchange: LDS serves a wave's accesses in order
    constexpr int Q = 32 * (C / 4), QIT = (Q + 63) / 64;
    f32x4 tv[QIT], pv[QIT];
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
        const int q = lane + 64 * it, row = q / (C / 4), c4 = q - row * (C / 4);
        const bool ok = q < Q && row0 + row < M;
        tv[it] = *reinterpret_cast<const f32x4 *>(tb + (q < Q ? row : 0) * PITCH + (q < Q ? c4 : 0) * 4);
        pv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok && post) pv[it] = *reinterpret_cast<const f32x4 *>(post + (int64_t)(row0 + row) * ldp + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
        const int q = lane + 64 * it, row = q / (C / 4), c4 = q - row * (C / 4);
        if (!(q < Q && row0 + row < M)) continue;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (scale2) sc = *reinterpret_cast<const f32x4 *>(scale2 + c4 * 4);
        if (bias2) bi = *reinterpret_cast<const f32x4 *>(bias2 + c4 * 4);
        f32x4 v = tv[it] * sc + bi;
        if (post) v += pv[it];
        *reinterpret_cast<f32x4 *>(out + (int64_t)(row0 + row) * ldo + c4 * 4) = v;
    }
}

template <int C, int ABL = 0>
int launch_mlp(const float *x, int64_t ldx, const uint16_t *w1p, const float *b1, const uint16_t *w2p, int ldn2, const float *scale2,
               const float *bias2, const float *post, int64_t ldp, float *out, int64_t ldo, int M, hipStream_t s) {
    constexpr int NPAD = (C + 31) / 32 * 32;
    const size_t smem = (size_t)2 * (3 * (C / 8) * 32 + 3 * 4 * NPAD) * 16 + (size_t)4 * C * sizeof(float);
    auto kern = convnext_mlp_kernel<C, ABL>;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(kern), smem);
    hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), smem, s, x, ldx, reinterpret_cast<const u32x4 *>(w1p), b1,
                       reinterpret_cast<const u32x4 *>(w2p), ldn2, scale2, bias2, post, ldp, out, ldo, M);
    return 0;
}

}  // namespace

extern "C" int mit_convnext_mlp_supported(int C) { return C == 80 ? 1 : 0; }

extern "C" int mit_convnext_mlp(const float *x_dev, int64_t ldx, int M, int C, const uint16_t *w1_planes_dev, const float *b1_dev,
                                const uint16_t *w2perm_planes_dev, int64_t ldn2, const float *scale2_dev, const float *bias2_dev,
                                const float *post_dev, int64_t ldp, float *out_dev, int64_t ldo, void *stream) {
    if (!x_dev || !w1_planes_dev || !b1_dev || !w2perm_planes_dev || !out_dev) return mit_set_error("mit_convnext_mlp: null pointer");
    if (!mit_convnext_mlp_supported(C)) return mit_set_error("mit_convnext_mlp: C = %d is not instantiated (80)", C);
    if (M <= 0) return 0;
    if (M > 0x7fffff00) return mit_set_error("mit_convnext_mlp: M too large");
    if ((ldx & 3) || (ldo & 3) || (ldp & 3) || ldx < C || ldo < C || (post_dev && ldp < C) || ldn2 < C || ldn2 > 0x7fffffff)
        return mit_set_error("mit_convnext_mlp: row strides must cover C and be multiples of 4 floats");
    auto unaligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    if (unaligned(x_dev) || unaligned(w1_planes_dev) || unaligned(w2perm_planes_dev) || unaligned(out_dev) || unaligned(post_dev) ||
        unaligned(scale2_dev) || unaligned(bias2_dev))
        return mit_set_error("mit_convnext_mlp: x, out, post, scale / bias and the plane tables must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // algorithmic bytes: the block input and the residual read once, the output written once (the hidden activations stay on chip);
    // FLOPs: both contractions
    MitProbeScope probe("convnext_mlp_kernel<80>", s, (double)M * C * 4.0 * 3.0, 2.0 * 2.0 * (double)M * C * (4.0 * C));
#ifdef MIT_CONV_EXPERIMENTS
    static const int abl = getenv("MIT_MLP_ABLATE") ? atoi(getenv("MIT_MLP_ABLATE")) : 0;
#define MIT_MLP_ABL(n) if (abl == n) { launch_mlp<80, n>(x_dev, ldx, w1_planes_dev, b1_dev, w2perm_planes_dev, (int)ldn2, scale2_dev, bias2_dev, post_dev, ldp, out_dev, ldo, M, s); MIT_CHECK_LAUNCH("mit_convnext_mlp"); return 0; }
    MIT_MLP_ABL(1) MIT_MLP_ABL(2) MIT_MLP_ABL(4) MIT_MLP_ABL(8) MIT_MLP_ABL(3) MIT_MLP_ABL(12) MIT_MLP_ABL(15)
#undef MIT_MLP_ABL
#endif
    switch (C) {
        case 80: launch_mlp<80>(x_dev, ldx, w1_planes_dev, b1_dev, w2perm_planes_dev, (int)ldn2, scale2_dev, bias2_dev, post_dev, ldp, out_dev, ldo, M, s); break;
        default: return mit_set_error("mit_convnext_mlp: C = %d", C);
    }
    MIT_CHECK_LAUNCH("mit_convnext_mlp");
    return 0;
}
