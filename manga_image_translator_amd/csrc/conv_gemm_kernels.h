// conv_gemm_kernels.h — implicit-GEMM convolution / batched GEMM on gfx950 fp32 MFMA.
//
// One kernel family serves every dense contraction of the hot path:
// Conv2d (any k, stride, zero/reflect padding), the stride-parity sub-convolutions of
// ConvTranspose2d, nn.Linear, and the DFT-as-GEMM stages of LaMa's FourierUnit.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32 — exact fp32, 64 cycles/instruction/SIMD):
//   workgroup = 256 threads = 4 waves arranged WAVES_M x WAVES_N, block tile BM x BN,
//   K-tile BK.  A (gathered NHWC activations, k contiguous in HBM) is staged
//   global -> VGPR (float4) -> LDS transposed to [k][m]; W ([K][N], n contiguous) is staged
//   global -> VGPR -> LDS [k][n].  Both LDS images are read with conflict-free ds_read_b32
//   (lanes 0-31 read 32 consecutive dwords of row k, lanes 32-63 row k+1), which is exactly
//   the 32x32x2 fragment layout (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]).  fp32 MFMA is slow
//   enough (64 cyc) that LDS bandwidth is <5 % utilised; the loop is MFMA-issue bound when
//   the next tile's global loads (issued before the MFMA block, written to the other LDS
//   buffer after it) land in time.  One barrier per K-tile.
//
// Summation order is k-sequential (tap-major, channel-minor) inside one accumulator, so
// results do not depend on the tile configuration or grid.
//
// The kernel templates live in this header so that their instantiations can be spread over several translation units
// (conv_gemm_inst*.hip, selected by the group column of conv_gemm_cfgs.inc) and compile in parallel; conv_gemm.hip holds the
// configuration table, the tile choice, the probe and the C entry points.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <vector>
#include "../../include/mit_hip.h"
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
namespace mitcg {

struct RowOff {
    int64_t c, pre, post;
};
struct LutOff {
    int32_t t1, t2;  // float offsets of a row's two lookup-table rows (MitConvGemm.lut_rows); kept in their own LDS array, only by launches that use them
};

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 in exact arithmetic, <= 6e-7 in f32 near 0 where GELU multiplies it by x/2):
// GELU through it is within 2.6e-7 absolute of the erff form over the whole range, at a third of the instructions.
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = v * 0.70710678118654752440f;
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.f));
    float q = 1.061405429f;
    q = __builtin_fmaf(q, t, -1.453152027f);
    q = __builtin_fmaf(q, t, 1.421413741f);
    q = __builtin_fmaf(q, t, -0.284496736f);
    q = __builtin_fmaf(q, t, 0.254829592f);
    q *= t;
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
    const float erf_abs = 1.f - q * e;
    const float hx = 0.5f * v;
    return hx + hx * copysignf(erf_abs, x);
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v, float alpha) {
    if (ACT == MIT_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == MIT_ACT_LEAKY) return v > 0.f ? v : v * alpha;
    if (ACT == MIT_ACT_SILU) return v / (1.f + expf(-v));
    if (ACT == MIT_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    if (ACT == MIT_ACT_GELU) return gelu_fast(v);
    return v;
}

// ---- epilogue shared by both kernels: row offsets computed once per row, shared through LDS ----
// The activation (and whether a residual joins) is a compile-time parameter of the store loop and dispatched once per
// wave: a per-element switch costs more than the stores on the small-K layers.
template <int TM, int TN, int ACT, bool HAS_POST, int XE>
__device__ __forceinline__ void epilogue_store(const MitConvGemm &p, f32x16 (&acc)[TM][TN], const RowOff *rowoff, const int n0,
                                               const int wm0, const int wn0, const int tid) {
    const int lane = tid & 63;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const bool has_pre = p.pre.base != nullptr;
    const bool post_first = (p.act & MIT_ACT_POST_FIRST) != 0;  // residual joins before the activation (ResNet BasicBlock)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn0 + ni * 32 + li;
        if (n >= p.N) continue;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float bi = p.bias ? p.bias[n] : 0.f;
        int64_t ncol_c = n, ncol_pre = n, ncol_post = n;
        if (p.c.nsplit) ncol_c = (int64_t)(n / p.c.nsplit) * p.c.nhi + (n % p.c.nsplit);
        if (has_pre && p.pre.nsplit) ncol_pre = (int64_t)(n / p.pre.nsplit) * p.pre.nhi + (n % p.pre.nsplit);
        if (HAS_POST && p.post.nsplit) ncol_post = (int64_t)(n / p.post.nsplit) * p.post.nhi + (n % p.post.nsplit);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const RowOff ro = rowoff[row];
                if (ro.c < 0) continue;
                float v = acc[mi][ni][r];
                if (has_pre) v += p.pre.base[ro.pre + ncol_pre];
                v = v * sc + bi;
                if (HAS_POST && post_first) v += p.post.base[ro.post + ncol_post];
                if (!(XE & 2)) v = apply_act<ACT>(v, p.act_alpha);
                if (HAS_POST && !post_first) v += p.post.base[ro.post + ncol_post];
                if (XE & 1) {  // timing ablation: results computed but (practically) never stored
                    if (v == 12345.678f) p.c.base[ro.c + ncol_c] = v;
                } else {
                    p.c.base[ro.c + ncol_c] = v;
                }
            }
        }
    }
}

// Vector form of the store loop: each 32x32 accumulator block goes through a per-wave LDS buffer so that a lane ends up with
// four consecutive columns of one row — 4 dwordx4 stores (and residual loads) per block instead of 16 dword ones.  Same
// per-element arithmetic in the same order, so results are bit-identical to the scalar loop.  Needs N % 4 == 0, plain (unsplit)
// column maps and 16-byte aligned bases / strides (MIT_ACT_VEC_OK, set by the launcher).
constexpr int EPI_PITCH = 36;
#define MIT_ACT_VEC_OK 0x200

template <int TM, int TN, int ACT, bool HAS_POST, int XE, bool HAS_LUT = false>
__device__ __forceinline__ void epilogue_store_vec(const MitConvGemm &p, f32x16 (&acc)[TM][TN], const RowOff *rowoff, float *tbuf,
                                                   const int n0, const int wm0, const int wn0, const int tid, const LutOff *lutoff = nullptr) {
    const int lane = tid & 63;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int vr = lane >> 3, vc = (lane & 7) * 4;
    const bool has_pre = p.pre.base != nullptr;
    const bool post_first = (p.act & MIT_ACT_POST_FIRST) != 0;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn0 + ni * 32 + vc;
        const bool n_ok = n < p.N;
        // column-split output map (planar spectra, q | k | v slabs): a float4 group never straddles a split (nsplit % 4 == 0, checked by
        // the launcher), so only its first column is mapped
        const int64_t nc = p.c.nsplit ? (int64_t)(n / p.c.nsplit) * p.c.nhi + (n % p.c.nsplit) : (int64_t)n;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (n_ok && p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + n);
        if (n_ok && p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tbuf[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_PITCH + li] = acc[mi][ni][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-synchronous exchange: LDS serves a wave's accesses in order
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rl = vr + 8 * j;
                f32x4 v = *reinterpret_cast<const f32x4 *>(tbuf + rl * EPI_PITCH + vc);
                const RowOff ro = rowoff[wm0 + mi * 32 + rl];
                if (ro.c < 0 || !n_ok) continue;
                if (has_pre) v += *reinterpret_cast<const f32x4 *>(p.pre.base + ro.pre + n);
                v = v * sc + bi;
                f32x4 pv = {0.f, 0.f, 0.f, 0.f};
                if (HAS_POST) pv = *reinterpret_cast<const f32x4 *>(p.post.base + ro.post + n);
                if (HAS_POST && post_first) v += pv;
                if (!(XE & 2)) {
                    v.x = apply_act<ACT>(v.x, p.act_alpha);
                    v.y = apply_act<ACT>(v.y, p.act_alpha);
                    v.z = apply_act<ACT>(v.z, p.act_alpha);
                    v.w = apply_act<ACT>(v.w, p.act_alpha);
                }
                if (HAS_POST && !post_first) v += pv;
                if (HAS_LUT) {  // the row-lookup form (its own instantiation): two table rows joined after everything else, in this order
                    const LutOff lo = lutoff[wm0 + mi * 32 + rl];
                    v += *reinterpret_cast<const f32x4 *>(p.lut1 + lo.t1 + n);
                    v += *reinterpret_cast<const f32x4 *>(p.lut2 + lo.t2 + n);
                }
                *reinterpret_cast<f32x4 *>(p.c.base + ro.c + nc) = v;
            }
            asm volatile("" ::: "memory");
        }
    }
}

template <int BM, int TM, int TN, int XE = 0, int SMEM_FLOATS = 0, int NTHR = 256>
__device__ __forceinline__ void epilogue(const MitConvGemm &p, f32x16 (&acc)[TM][TN], float *smem, const int M, const int m0,
                                         const int n0, const int wm0, const int wn0, const int z1, const int z0,
                                         const int HoWo, const int tid) {
    // tid: the caller's thread index — threadIdx.x, or the same number rebuilt after the K loop (wave index kept in an SGPR, lane from
    // mbcnt) so that no thread-index-derived register has to stay live, or be spilled to scratch, across the loop
    RowOff *rowoff = reinterpret_cast<RowOff *>(smem);  // BM entries (<= A/B staging area)
    const int64_t c_dyn = p.dyn ? (int64_t)(*p.dyn) * p.c_dyn : 0;  // device-side step offset (hipGraph-replayed sequences), else 0
    for (int r = tid; r < BM; r += NTHR) {
        const int m = m0 + r;
        RowOff ro = {-1, 0, 0};
        if (m < M) {
            const int nb = m / HoWo;
            const int rem = m - nb * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            ro.c = c_dyn + z1 * p.c.zs1 + z0 * p.c.zs0 + (int64_t)nb * p.c.bs + (int64_t)oy * p.c.ys + (int64_t)ox * p.c.xs;
            ro.pre = z1 * p.pre.zs1 + z0 * p.pre.zs0 + (int64_t)nb * p.pre.bs + (int64_t)oy * p.pre.ys +
                     (int64_t)ox * p.pre.xs;
            ro.post = z1 * p.post.zs1 + z0 * p.post.zs0 + (int64_t)nb * p.post.bs + (int64_t)oy * p.post.ys +
                      (int64_t)ox * p.post.xs;
        }
        rowoff[r] = ro;
    }
    constexpr int ROWOFF_FLOATS = (BM * (int)sizeof(RowOff) + 15) / 16 * 4;
    constexpr int LUT_FLOATS = BM * (int)sizeof(LutOff) / 4;   // the row-lookup offsets sit behind the transpose buffers
    constexpr int NWAVES = NTHR / 64;  // one transpose buffer per wave
    constexpr bool VEC_FITS = SMEM_FLOATS >= ROWOFF_FLOATS + NWAVES * 32 * EPI_PITCH + LUT_FLOATS && !(XE & 1);
    LutOff *lutoff = reinterpret_cast<LutOff *>(smem + ROWOFF_FLOATS + NWAVES * 32 * EPI_PITCH);
    const bool lut = VEC_FITS && p.lut_rows != nullptr;   // (the launcher admits lut_rows only on kernels and operands that take the vector path)
    if (VEC_FITS && lut) {
        for (int r = tid; r < BM; r += NTHR) {
            const int m = m0 + r;
            const unsigned int packed = m < M ? (unsigned int)p.lut_rows[m] : 0u;
            lutoff[r] = LutOff{(int32_t)((packed & 0xffffu) * (unsigned int)p.lut_ld), (int32_t)((packed >> 16) * (unsigned int)p.lut_ld)};
        }
    }
    __syncthreads();

    const bool has_post = p.post.base != nullptr;
    float *tbuf = smem + ROWOFF_FLOATS + (tid >> 6) * (32 * EPI_PITCH);
    const bool vec = VEC_FITS && (p.act & MIT_ACT_VEC_OK);
    if (VEC_FITS && vec && lut) {  // act in {none, relu}, no post residual (mit_conv_gemm_cfg checks)
        if ((p.act & 0xff) == MIT_ACT_RELU) epilogue_store_vec<TM, TN, MIT_ACT_RELU, false, XE, true>(p, acc, rowoff, tbuf, n0, wm0, wn0, tid, lutoff);
        else epilogue_store_vec<TM, TN, MIT_ACT_NONE, false, XE, true>(p, acc, rowoff, tbuf, n0, wm0, wn0, tid, lutoff);
        return;
    }
#define MIT_EPI(A)                                                                                   \
    if (VEC_FITS && vec) {                                                                           \
        if (has_post) epilogue_store_vec<TM, TN, A, true, XE>(p, acc, rowoff, tbuf, n0, wm0, wn0, tid); \
        else epilogue_store_vec<TM, TN, A, false, XE>(p, acc, rowoff, tbuf, n0, wm0, wn0, tid);         \
    } else if (has_post) epilogue_store<TM, TN, A, true, XE>(p, acc, rowoff, n0, wm0, wn0, tid);      \
    else epilogue_store<TM, TN, A, false, XE>(p, acc, rowoff, n0, wm0, wn0, tid)
    switch (p.act & 0xff) {
        case MIT_ACT_RELU: MIT_EPI(MIT_ACT_RELU); break;
        case MIT_ACT_LEAKY: MIT_EPI(MIT_ACT_LEAKY); break;
        case MIT_ACT_SILU: MIT_EPI(MIT_ACT_SILU); break;
        case MIT_ACT_SIGMOID: MIT_EPI(MIT_ACT_SIGMOID); break;
        case MIT_ACT_GELU: MIT_EPI(MIT_ACT_GELU); break;
        default: MIT_EPI(MIT_ACT_NONE); break;
    }
#undef MIT_EPI
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const MitConvGemm p, const int M, const int MT,
                                                          const int NT, const int KT) {
    constexpr int WM = BM / WAVES_M;  // wave tile rows
    constexpr int WN = BN / WAVES_N;
    constexpr int TM = WM / 32;  // 32x32 blocks per wave
    constexpr int TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int KQ = BK / 4;                    // float4 chunks along k
    constexpr int A_ITERS = BM * KQ / 256;        // float4 loads of A per thread per K-tile
    constexpr int A_MSTEP = 256 / KQ;             // rows covered per iteration
    constexpr int NQ = BN / 4;                    // float4 chunks along n
    constexpr int B_ITERS = (BK * NQ + 255) / 256;
    constexpr int B_KSTEP = 256 / NQ;
    constexpr bool B_PARTIAL = (BK * NQ) < 256;  // fewer float4 chunks than threads
    static_assert(A_ITERS >= 1 && (BM * KQ) % 256 == 0, "A tile must fill the workgroup");
    constexpr int LDA = BM + (BK == 16 ? 2 : 1);  // ds_write_b32 conflict-free transposed store
    constexpr int LDB = BN + 4;
    constexpr int A_TILE = BK * LDA;
    constexpr int B_TILE = BK * LDB;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;               // [2][BK][LDA]
    float *Bs = smem + 2 * A_TILE;  // [2][BK][LDB]
    int *tapinfo = reinterpret_cast<int *>(smem + 2 * A_TILE + 2 * B_TILE);  // [ntaps][2]: packed dy/dx, off

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;

    // ---- block -> tile mapping (XCD-aware: block b runs on XCD b % 8; give each XCD a
    // contiguous run of tiles, n fastest, so tiles sharing an A panel share an L2) ----
    const int nwg = MT * NT;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / NT, nt = bid - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y;
    const int z1 = z / p.zdiv, z0 = z - z1 * p.zdiv;

    const float *__restrict__ a_base = p.a + z1 * p.a_zs1 + z0 * p.a_zs0 + (p.dyn ? (int64_t)(*p.dyn) * p.a_dyn : 0);
    const float *__restrict__ w_base = p.w + z1 * p.w_zs1 + z0 * p.w_zs0;

    for (int t = tid; t < p.ntaps; t += 256) {
        tapinfo[2 * t] = (int(p.tap_dy[t]) & 0xffff) | (int(p.tap_dx[t]) << 16);
        tapinfo[2 * t + 1] = p.tap_off[t];
    }

    // ---- per-thread A rows (fixed across the K loop) ----
    const int aq = tid % KQ;  // float4 chunk along k
    const int am = tid / KQ;  // first row
    int64_t a_rowbase[A_ITERS];
    int a_iy0[A_ITERS], a_ix0[A_ITERS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int m = m0 + am + i * A_MSTEP;
        if (m < M) {
            const int nb = m / HoWo;
            const int rem = m - nb * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_rowbase[i] = (int64_t)nb * p.a_bs;
            a_iy0[i] = oy * p.sy;
            a_ix0[i] = ox * p.sx;
        } else {
            a_rowbase[i] = 0;
            a_iy0[i] = -(1 << 28);  // forces out-of-range -> zero (also under reflect, see below)
            a_ix0[i] = 0;
        }
    }
    const int bn4 = tid % NQ;
    const int bk = tid / NQ;
    const bool b_ncol_ok = (n0 + bn4 * 4) < p.Nw;

    const int Ktot = p.ntaps * p.Cin;

    f32x4 a_reg[A_ITERS];
    f32x4 b_reg[B_ITERS];

    __syncthreads();  // tapinfo visible

    auto load_tile = [&](int kt) {
        // A: gather
        const int kk = kt * BK + aq * 4;
        const bool kvalid = kk < Ktot;
        int tap = 0, ci = 0, dy = 0, dx = 0, toff = 0;
        if (kvalid) {
            tap = kk / p.Cin;
            ci = kk - tap * p.Cin;
            const int packed = tapinfo[2 * tap];
            dy = (int)(short)(packed & 0xffff);
            dx = packed >> 16;
            toff = tapinfo[2 * tap + 1];
        }
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            int iy = a_iy0[i] + dy;
            int ix = a_ix0[i] + dx;
            bool ok = kvalid && (a_iy0[i] >= 0);
            if (p.pad_mode == MIT_PAD_REFLECT) {
                iy = iy < 0 ? -iy : (iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix);
            } else {
                ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float *ptr = a_base + a_rowbase[i] + (int64_t)iy * p.a_ys + (int64_t)ix * p.a_xs + toff + ci;
                v = *reinterpret_cast<const f32x4 *>(ptr);
            }
            a_reg[i] = v;
        }
        // W
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int k = kt * BK + bk + i * B_KSTEP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ncol_ok && k < p.Kw && (!B_PARTIAL || bk < BK)) {
                const float *ptr = w_base + (int64_t)k * p.ldw + n0 + bn4 * 4;
                v = *reinterpret_cast<const f32x4 *>(ptr);
            }
            b_reg[i] = v;
        }
    };

    auto store_tile = [&](int buf) {
        float *as = As + buf * A_TILE;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int ml = am + i * A_MSTEP;
#pragma unroll
            for (int j = 0; j < 4; ++j) as[(aq * 4 + j) * LDA + ml] = a_reg[i][j];
        }
        float *bs = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int kl = bk + i * B_KSTEP;
            if (!B_PARTIAL || bk < BK) *reinterpret_cast<f32x4 *>(bs + kl * LDB + bn4 * 4) = b_reg[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_tile(kt + 1);
        const float *as = As + cur * A_TILE + lh * LDA + wm0 + li;
        const float *bs = Bs + cur * B_TILE + lh * LDB + wn0 + li;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float af[TM], bf[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[mi] = as[(2 * ks) * LDA + mi * 32];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[ni] = bs[(2 * ks) * LDB + ni * 32];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < KT) store_tile(cur ^ 1);
        __syncthreads();
    }

    epilogue<BM, TM, TN, 0, 2 * A_TILE + 2 * B_TILE>(p, acc, smem, M, m0, n0, wm0, wn0, z1, z0, HoWo, (int)threadIdx.x);
}

// ---- fast path: Cin % BK == 0, so every K-tile lies inside ONE tap ----------------------------
// The gather geometry (reflect / zero padding, strides, tap offsets) is evaluated once per block
// into an LDS table rowtab[tap][row] of 32-bit element offsets (-1 = contributes zeros); the K
// loop then costs one ds_read_b32 + one 64-bit add per 16-byte load, the tap index and channel
// offset are wave-uniform scalars, and the next k-step's fragments are read from LDS while the
// current step's MFMAs issue.  Same tiling, LDS images, accumulation order and epilogue as the
// generic kernel, so results are bitwise identical to it.
constexpr int FAST_MAX_TAPS = 16;

// VAR bits (scheduling variants, identical arithmetic): 1 = write-after-barrier rotation, 2 = fragment reads of k-step s+1 pinned
// ahead of the MFMAs of k-step s (sched_barrier), 4 = gather offsets kept in registers while the tap does not change +
// incremental weight pointer, 8 = raised wave priority while MFMAs issue.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW, int VAR = 0>
__global__ __launch_bounds__(256, MINW) void conv_gemm_fast_kernel(const MitConvGemm p, const int M, const int MT,
                                                               const int NT, const int KT) {
    constexpr int WM = BM / WAVES_M;
    constexpr int WN = BN / WAVES_N;
    constexpr int TM = WM / 32;
    constexpr int TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr bool ROT = (VAR & 1) != 0;
    constexpr bool PIPE = (VAR & 2) != 0;
    constexpr bool CACHE = (VAR & 4) != 0;
    constexpr bool PRIO = (VAR & 8) != 0;
    constexpr bool MID = (VAR & 16) != 0;
    // 16384: A fetched in full 128-byte lines — 8 lanes x 16 B per row, i.e. the 16-channel slices of TWO consecutive K-tiles in one
    // load (8 rows x 128 B per wave instruction instead of 16 rows x 64 B); needs Cin % 32 == 0.  Each half goes to LDS in its own K-tile.
    constexpr bool A2 = (VAR & 16384) != 0;
    constexpr int A2_ITERS = BM / 32;
    // timing ablations (WRONG results; scripts/bench_conv.py only): skip the in-loop global loads / LDS stores / barrier / fragment reads
    constexpr bool X_NOA = (VAR & 512) != 0, X_NOB = (VAR & 1024) != 0, X_HOT = (VAR & 2048) != 0;  // skip A / B loads; A rows folded into 64 KB
    constexpr bool X_NOLOAD = (VAR & 32) != 0, X_NOSTORE = (VAR & 64) != 0, X_NOBAR = (VAR & 128) != 0, X_NOFRAG = (VAR & 256) != 0;  // next tile's LDS stores issued between the MFMAs of k-steps 4..6, not after the last one
    constexpr int KQ = BK / 4;
    constexpr int A_ITERS = BM * KQ / 256;
    constexpr int A_MSTEP = 256 / KQ;
    constexpr int NQ = BN / 4;
    constexpr int B_ITERS = (BK * NQ + 255) / 256;
    constexpr int B_KSTEP = 256 / NQ;
    constexpr bool B_PARTIAL = (BK * NQ) < 256;  // BN = 32: 128 float4 chunks per K-tile, the upper half of the workgroup loads nothing
    static_assert(A_ITERS >= 1 && (BM * KQ) % 256 == 0, "A tile must fill the workgroup");
    static_assert(B_PARTIAL || (BK * NQ) % 256 == 0, "B tile must fill the workgroup or fit in one pass");
    constexpr int LDA = BM + (BK == 16 ? 2 : 1);
    constexpr int LDB = BN + 4;
    constexpr int A_TILE = BK * LDA;
    constexpr int B_TILE = BK * LDB;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * A_TILE;
    int *rowtab = reinterpret_cast<int *>(smem + 2 * A_TILE + 2 * B_TILE);  // [ntaps][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;

    const int nwg = MT * NT;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / NT, nt = bid - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y;
    const int z1 = z / p.zdiv, z0 = z - z1 * p.zdiv;
    const int HoWo = p.Ho * p.Wo;

    const float *__restrict__ a_base = p.a + z1 * p.a_zs1 + z0 * p.a_zs0 + (p.dyn ? (int64_t)(*p.dyn) * p.a_dyn : 0);
    const float *__restrict__ w_base = p.w + z1 * p.w_zs1 + z0 * p.w_zs0;

    // ---- gather table ----
    for (int idx = tid; idx < p.ntaps * BM; idx += 256) {
        const int t = idx / BM, r = idx - t * BM;
        const int m = m0 + r;
        int off = -1;
        if (m < M) {
            const int nb = m / HoWo;
            const int rem = m - nb * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            int iy = oy * p.sy + p.tap_dy[t];
            int ix = ox * p.sx + p.tap_dx[t];
            bool ok = true;
            if (p.pad_mode == MIT_PAD_REFLECT) {
                iy = iy < 0 ? -iy : (iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix);
            } else {
                ok = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            if (ok) off = (int)((int64_t)nb * p.a_bs + (int64_t)iy * p.a_ys + (int64_t)ix * p.a_xs + p.tap_off[t]);
        }
        rowtab[idx] = off;
    }

    const int aq = tid % KQ;
    const int am = tid / KQ;
    const int bn4 = tid % NQ;
    const int bk = tid / NQ;
    const bool b_ncol_ok = (n0 + bn4 * 4) < p.Nw && (!B_PARTIAL || bk < BK);
    const float *__restrict__ a_thr = a_base + aq * 4;
    const float *__restrict__ w_thr = w_base + n0 + bn4 * 4;

    f32x4 a_reg[A_ITERS];
    f32x4 b_reg[B_ITERS];

    __syncthreads();  // rowtab visible

    // (tap, ci0) of the tile being loaded: wave-uniform, advanced incrementally
    int ld_tap = 0, ld_ci0 = 0;
    int a_off[A_ITERS];
    const int aq8 = tid & 7, am8 = tid >> 3;
    f32x4 a2_reg[A2 ? A2_ITERS : 1];
    int a2_off[A2 ? A2_ITERS : 1];
    auto load_a2 = [&]() {  // K-tiles (2j, 2j + 1): channels ld_ci0 .. ld_ci0 + 31 of tap ld_tap
        if (ld_ci0 == 0) {
            const int *rt = rowtab + ld_tap * BM + am8;
#pragma unroll
            for (int i = 0; i < A2_ITERS; ++i) a2_off[i] = rt[i * 32];
        }
        const float *ak = a_base + ld_ci0 + aq8 * 4;
#pragma unroll
        for (int i = 0; i < A2_ITERS; ++i) {
            const int off = a2_off[i];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (off >= 0) v = *reinterpret_cast<const f32x4 *>(ak + off);
            a2_reg[i] = v;
        }
        ld_ci0 += 2 * BK;
        if (ld_ci0 >= p.Cin) {
            ld_ci0 = 0;
            ++ld_tap;
        }
    };
    auto store_a2 = [&](int buf, int par) {  // the lanes holding K-tile parity `par` of the pair
        if ((aq8 >> 2) == par) {
            float *as = As + buf * A_TILE + ((aq8 & 3) * 4) * LDA + am8;
#pragma unroll
            for (int i = 0; i < A2_ITERS; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) as[j * LDA + i * 32] = a2_reg[i][j];
        }
    };
    const float *__restrict__ w_row = w_thr + (int64_t)bk * p.ldw;  // CACHE: row (kt*BK + bk) of this thread's weight column
    const int64_t w_kstep = (int64_t)B_KSTEP * p.ldw, w_tstep = (int64_t)BK * p.ldw;
    auto load_tile = [&](int kt) {
        const int *rt = rowtab + ld_tap * BM + am;
        const float *ak = a_thr + ld_ci0;
        if (!A2 && (!CACHE || ld_ci0 == 0)) {  // wave-uniform: the row offsets only change with the tap
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) a_off[i] = rt[i * A_MSTEP];
        }
#pragma unroll
        for (int i = 0; i < (A2 ? 0 : A_ITERS); ++i) {
            const int off = X_HOT ? (a_off[i] & 0x3ffc) : a_off[i];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!X_NOA) {
                if (off >= 0) v = *reinterpret_cast<const f32x4 *>(ak + off);
                a_reg[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int k = kt * BK + bk + i * B_KSTEP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (CACHE) {
                if (b_ncol_ok && k < p.Kw) v = *reinterpret_cast<const f32x4 *>(w_row + i * w_kstep);
            } else {
                if (b_ncol_ok && k < p.Kw) v = *reinterpret_cast<const f32x4 *>(w_thr + (int64_t)k * p.ldw);
            }
            if (!X_NOB) b_reg[i] = v;
        }
        if (CACHE) w_row += w_tstep;
        if (!A2) {
            ld_ci0 += BK;
            if (ld_ci0 >= p.Cin) {
                ld_ci0 = 0;
                ++ld_tap;
            }
        }
    };

    auto store_a = [&](int buf) {
        float *as = As + buf * A_TILE;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int ml = am + i * A_MSTEP;
#pragma unroll
            for (int j = 0; j < 4; ++j) as[(aq * 4 + j) * LDA + ml] = a_reg[i][j];
        }
    };
    auto store_b = [&](int buf) {
        float *bs = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int kl = bk + i * B_KSTEP;
            if (!B_PARTIAL || bk < BK) *reinterpret_cast<f32x4 *>(bs + kl * LDB + bn4 * 4) = b_reg[i];
        }
    };
    auto store_tile = [&](int buf) {
        if (!A2) store_a(buf);
        store_b(buf);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;

    if (A2) load_a2();
    load_tile(0);
    if (A2) store_a2(0, 0);
    store_tile(0);
    if (ROT && KT > 1) load_tile(1);  // stays in registers across the barrier
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        const float *as = As + cur * A_TILE + lh * LDA + wm0 + li;
        const float *bs = Bs + cur * B_TILE + lh * LDB + wn0 + li;
        float af[2][TM], bf[2][TN];
        if (MID) {  // first fragments on their way while the global loads are being issued
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[0][mi] = as[mi * 32];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[0][ni] = bs[ni * 32];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ROT) {  // tile kt+1 (loaded during the previous iteration) -> LDS right after the barrier, then fetch kt+2
            if (kt + 1 < KT) store_tile(cur ^ 1);
            if (kt + 2 < KT) load_tile(kt + 2);
        } else if (kt + 1 < KT && !X_NOLOAD) {
            if (A2 && (kt & 1)) load_a2();  // both halves of the previous pair are in LDS by now
            load_tile(kt + 1);
        }
        if (!MID) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[0][mi] = as[mi * 32];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[0][ni] = bs[ni * 32];
        }
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int c = ks & 1;
            if (ks + 1 < BK / 2 && !X_NOFRAG) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[c ^ 1][mi] = as[(2 * ks + 2) * LDA + mi * 32];
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bf[c ^ 1][ni] = bs[(2 * ks + 2) * LDB + ni * 32];
            } else if (X_NOFRAG) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[c ^ 1][mi] = af[c][mi] + 1.f;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bf[c ^ 1][ni] = bf[c][ni];
            }
            if (PIPE || MID) __builtin_amdgcn_sched_barrier(0);  // the reads above stay ahead of this step's MFMAs
            if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][mi], bf[c][ni], acc[mi][ni], 0, 0, 0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (PIPE || MID) __builtin_amdgcn_sched_barrier(0);
            if (MID && !ROT && kt + 1 < KT) {
                if (ks == BK / 2 - 4) store_a(cur ^ 1);
                if (ks == BK / 2 - 3) store_b(cur ^ 1);
                if (ks == BK / 2 - 4 || ks == BK / 2 - 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!ROT && !MID && kt + 1 < KT && !X_NOSTORE) {
            if (A2) store_a2(cur ^ 1, (kt + 1) & 1);
            store_tile(cur ^ 1);
        }
        if (!X_NOBAR) __syncthreads();
    }
    if (X_NOBAR) __syncthreads();

    // the launcher sizes the dynamic LDS for the larger of the staging area and the epilogue's row table + per-wave transpose
    // buffers, so small tiles (64 x 64) get the dwordx4 store path too
    constexpr int EPI_FLOATS = (BM * (int)sizeof(RowOff) + 15) / 16 * 4 + 4 * 32 * EPI_PITCH + BM * (int)sizeof(LutOff) / 4;
    constexpr int SMEM_F = (2 * A_TILE + 2 * B_TILE) > EPI_FLOATS ? (2 * A_TILE + 2 * B_TILE) : EPI_FLOATS;
    epilogue<BM, TM, TN, (VAR >> 12) & 3, SMEM_F>(p, acc, smem, M, m0, n0, wm0, wn0, z1, z0, HoWo, (int)threadIdx.x);
}
// ---- N <= 4: one output column group per row — a dot product, not a tile ------------------------------------------------
// An MFMA tile would idle >= 28 of its 32 columns (the ctd heads' last ConvTranspose2d 64 -> 1 and 16 -> 1 ran at 2 TFLOP/s on
// the 128 x 32 tile).  Here LPR lanes share one output row: each tap's Cin contiguous floats are read as float4 by those lanes
// together (64 / LPR rows x Cin * 4 bytes per wave instruction, fully coalesced), weights come transposed from LDS, partial sums
// meet in a shuffle tree.  L1/L2-bound on the A gather.  The summation order is lane-major (not the k-sequential chain of the
// MFMA tiles), so results differ from them by fp32 rounding (<= 1e-6 relative on these layers).
template <int RPI, int NMAX, int KV, int LPR>
__global__ __launch_bounds__(256) void conv_gemv_kernel(const MitConvGemm p, const int M, const int MT, const int NT, const int KT) {
    static_assert((NMAX == 1 || NMAX == 4) && KV == 4 && (LPR == 4 || LPR == 16), "gemv shape");
    constexpr int RPW = 64 / LPR;  // output pixels per wave per iteration
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ktot = p.ntaps * p.Cin;
    float *wt = smem;  // [NMAX][Ktot]
    for (int i = threadIdx.x; i < Ktot * NMAX; i += 256) {
        const int n = i / Ktot, k = i - n * Ktot;
        wt[i] = (n < p.Nw) ? p.w[(int64_t)k * p.ldw + n] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rslot = lane / LPR;
    // one output row (nb, oy) per blockIdx.y: the row decode and every tap's input row are wave-uniform
    const int nb = blockIdx.y / p.Ho, oy = blockIdx.y - nb * p.Ho;
    const float *arow = p.a + (int64_t)nb * p.a_bs;
    const int cq = p.Cin >> 2;  // float4 chunks per tap
    const int64_t oc0 = (int64_t)nb * p.c.bs + (int64_t)oy * p.c.ys;
    const int64_t opre0 = (int64_t)nb * p.pre.bs + (int64_t)oy * p.pre.ys;
    const int64_t opost0 = (int64_t)nb * p.post.bs + (int64_t)oy * p.post.ys;
    const bool post_first = (p.act & MIT_ACT_POST_FIRST) != 0;
    const bool reflect = p.pad_mode == MIT_PAD_REFLECT;
#pragma unroll 1
    for (int it = 0; it < RPI; ++it) {
        const int ox = ((blockIdx.x * 4 + wave) * RPI + it) * RPW + rslot;
        const bool mok = ox < p.Wo;
        float acc[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
        for (int t = 0; t < p.ntaps; ++t) {
            int iy = oy * p.sy + p.tap_dy[t], ix = ox * p.sx + p.tap_dx[t];
            bool ok = mok;
            if (reflect) {
                iy = iy < 0 ? -iy : (iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix);
            } else {
                ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            const float *ptr = arow + (int64_t)iy * p.a_ys + (int64_t)ix * p.a_xs + p.tap_off[t];
            const float *wrow = wt + t * p.Cin;
            for (int q = sub; q < cq; q += LPR) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4 *>(ptr + q * 4);
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wrow + n * Ktot + q * 4);
                    acc[n] = __builtin_fmaf(v.x, w4.x, acc[n]);
                    acc[n] = __builtin_fmaf(v.y, w4.y, acc[n]);
                    acc[n] = __builtin_fmaf(v.z, w4.z, acc[n]);
                    acc[n] = __builtin_fmaf(v.w, w4.w, acc[n]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o);
        if (mok && sub == 0) {
            const int64_t oc = oc0 + (int64_t)ox * p.c.xs, opre = opre0 + (int64_t)ox * p.pre.xs, opost = opost0 + (int64_t)ox * p.post.xs;
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                if (n >= p.N) break;
                float v = acc[n];
                if (p.pre.base) v += p.pre.base[opre + n];
                v = v * (p.scale ? p.scale[n] : 1.f) + (p.bias ? p.bias[n] : 0.f);
                const float pv = p.post.base ? p.post.base[opost + n] : 0.f;
                if (post_first) v += pv;
                switch (p.act & 0xff) {
                    case MIT_ACT_RELU: v = apply_act<MIT_ACT_RELU>(v, p.act_alpha); break;
                    case MIT_ACT_LEAKY: v = apply_act<MIT_ACT_LEAKY>(v, p.act_alpha); break;
                    case MIT_ACT_SILU: v = apply_act<MIT_ACT_SILU>(v, p.act_alpha); break;
                    case MIT_ACT_SIGMOID: v = apply_act<MIT_ACT_SIGMOID>(v, p.act_alpha); break;
                    case MIT_ACT_GELU: v = apply_act<MIT_ACT_GELU>(v, p.act_alpha); break;
                    default: break;
                }
                if (!post_first) v += pv;
                p.c.base[oc + n] = v;
            }
        }
    }
}

struct CfgEntry {
    const char *name;
    int BM, BN, BK;
    void (*launch)(const MitConvGemm &, int M, int MT, int NT, int KT, hipStream_t);
    int fast;  // 1: conv_gemm_fast_kernel (needs fast_eligible()); 2: and Cin % 32 == 0; 3: conv_gemv_kernel (needs gemv_eligible()); 4: conv_gemm_split_kernel (needs split_eligible())
    const char *kernel;  // the kernel's template-id as profilers print it, e.g. "conv_gemm_fast_kernel<128, 128, 16, 1, 4, 4, 4>"
};

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
size_t smem_bytes() {
    constexpr int LDA = BM + (BK == 16 ? 2 : 1);
    constexpr int LDB = BN + 4;
    size_t staging = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float) + MIT_MAX_TAPS * 2 * sizeof(int);
    size_t rows = (size_t)BM * sizeof(RowOff);
    return staging > rows ? staging : rows;
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
void launch_cfg(const MitConvGemm &p, int M, int MT, int NT, int KT, hipStream_t s) {
    dim3 grid(MT * NT, p.Z, 1);
    size_t smem = smem_bytes<BM, BN, BK, WAVES_M, WAVES_N>();
    auto kern = conv_gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N>;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(kern), smem);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p, M, MT, NT, KT);
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW, int VAR = 0>
void launch_fast(const MitConvGemm &p, int M, int MT, int NT, int KT, hipStream_t s) {
    constexpr int LDA = BM + (BK == 16 ? 2 : 1);
    constexpr int LDB = BN + 4;
    size_t staging = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float) + (size_t)p.ntaps * BM * sizeof(int);
    size_t rows = ((size_t)BM * sizeof(RowOff) + 15) / 16 * 16 + (size_t)4 * 32 * EPI_PITCH * sizeof(float) + (size_t)BM * sizeof(LutOff);  // row table + transpose buffers + lookup offsets
    size_t smem = staging > rows ? staging : rows;
    auto kern = conv_gemm_fast_kernel<BM, BN, BK, WAVES_M, WAVES_N, MINW, VAR>;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(kern), smem);
    dim3 grid(MT * NT, p.Z, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p, M, MT, NT, KT);
}
template <int RPI, int NMAX, int KV, int LPR>
void launch_gemv(const MitConvGemm &p, int M, int MT, int NT, int KT, hipStream_t s) {
    constexpr int X_PER_BLOCK = 4 * RPI * (64 / LPR);
    const size_t smem = (size_t)p.ntaps * p.Cin * NMAX * sizeof(float);
    dim3 grid((p.Wo + X_PER_BLOCK - 1) / X_PER_BLOCK, p.NB * p.Ho, 1);
    hipLaunchKernelGGL((conv_gemv_kernel<RPI, NMAX, KV, LPR>), grid, dim3(256), smem, s, p, M, MT, NT, KT);
}
}  // namespace mitcg

#include "conv_gemm_split.h"  // conv_gemm_split_kernel, gemm_split_pack_kernel, launch_split (the split-bf16 tiles: GEMM mode 6 | 9)
#include "conv_gemm_split_pp.h"  // conv_gemm_split_pp_kernel, launch_split_pp: the eight-wave ping-pong form of the split tile
