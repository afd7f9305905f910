// conv_gemm_split_pp.h — the ping-pong form of the split-bf16 tile (round 6): 512 threads = two waves per SIMD (waves w and w + 4 share
// one), the second half of the workgroup running ONE barrier behind the first.  Between two barriers one wave of a SIMD is in its
// COMPUTE segment — the 6 * TM * TN MFMAs of a K-tile and nothing else, s_setprio 1 — while its partner is in its LOAD segment: fragment
// reads of its own next K-tile, the split (VALU) and the LDS writes of the K-tile after that, the global loads of the one after that.
// The matrix pipe of a SIMD sees one MFMA stream at a time, and the staging work that the four-wave tile (conv_gemm_split.h) threads
// between its MFMAs runs on the other issue ports of the same SIMD meanwhile (MI355X_MICROARCH.md, "Two waves per SIMD").
//
// Same arithmetic as every p6 tile: per accumulator and 16-deep k step the plane pairs kSplitPA / kSplitPB[3 .. 8] in that order, fp32
// accumulation in the MFMA, the shared epilogue — bit-identical results (tests/test_conv_gemm_gpu.py, scripts/split_check).
//
// Timeline (g = barrier count; first half: L(t) in [2t, 2t+1], C(t) in [2t+1, 2t+2]; second half one barrier later):
//   L(t):  ds_read fragments of K-tile t (LDS stage t & 1)  ->  split + ds_write the registers holding K-tile t+1 into stage (t+1) & 1
//          ->  global loads of K-tile t+3 into the same registers (two register sets)  ->  lgkmcnt(0), barrier
//   C(t):  MFMAs of K-tile t, barrier
// Hazards: stage (t+1) & 1 held K-tile t-1, whose fragments both halves have read (and waited for) before barrier 2t, and L(t) starts
// at barrier 2t or later; K-tile t+1 is complete in LDS before barrier 2t+2 (both halves' L(t) end at or before it), and is first read
// in the first half's L(t+1), which starts there.  Two stages suffice.
#pragma once

namespace mitcg {

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW, int NPROD, int VAR = 0>
__global__ __launch_bounds__(512, MINW) void conv_gemm_split_pp_kernel(const MitConvGemm p, const int M, const int MT, const int NT,
                                                                     const int KT) {
    constexpr int NTHR = 512;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 8 && TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile");
    static_assert(BK == 16 && NPROD == 6, "one MFMA k step per K-tile, six plane pairs");
    constexpr int KH = 2, KQ = 4;
    constexpr int A_ITERS = BM * KQ / NTHR, A_MSTEP = NTHR / KQ;
    static_assert(A_ITERS >= 1 && (BM * KQ) % NTHR == 0, "A tile must fill the workgroup");
    constexpr int SA = BM, SB = BN;
    constexpr int A_TILE = 3 * KH * SA, B_TILE = 3 * KH * SB;
    constexpr int B_CPP = KH * BN, B_CELLS = 3 * B_CPP, B_ITERS = (B_CELLS + NTHR - 1) / NTHR;
    constexpr bool X_NOSPLIT = (VAR & 4) != 0, X_NOMFMA = (VAR & 8) != 0;  // timing ablations (WRONG results; MIT_CONV_EXPERIMENTS builds)
    constexpr bool PRIO = (VAR & 2) == 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4 *As = reinterpret_cast<u32x4 *>(smem);             // [2][3][KH][SA]
    u32x4 *Bs = As + 2 * A_TILE;                             // [2][3][KH][SB]
    int *rowtab = reinterpret_cast<int *>(Bs + 2 * B_TILE);  // [ntaps][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const bool second = wave >= 4;  // waves w and w + 4 share a SIMD

    const int nwg = MT * NT;
    int tile;  // block id -> position in the XCD-contiguous order (as conv_gemm_split_kernel)
    {
        const int v = blockIdx.x, xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    const int mt = tile / NT, nt = tile - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y;
    const int z1 = z / p.zdiv, z0 = z - z1 * p.zdiv;
    const int HoWo = p.Ho * p.Wo;

    const float *__restrict__ a_base = p.a + z1 * p.a_zs1 + z0 * p.a_zs0 + (p.dyn ? (int64_t)(*p.dyn) * p.a_dyn : 0);
    const u32x4 *__restrict__ ws = reinterpret_cast<const u32x4 *>(p.w_split + z0 * p.ws_zs0);
    const int K8 = p.Kw >> 3;
    const int ldn = (int)p.ldw;

    const int aq = tid % KQ, am = tid / KQ;
    const float *__restrict__ a_thr = a_base + aq * 4;

    for (int idx = tid; idx < p.ntaps * BM; idx += NTHR) {  // gather table of the output tile
        const int tp = idx / BM, r = idx - tp * BM;
        const int m = m0 + r;
        int off = -1;
        if (m < M) {
            const int nb = m / HoWo;
            const int rem = m - nb * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            int iy = oy * p.sy + p.tap_dy[tp];
            int ix = ox * p.sx + p.tap_dx[tp];
            bool ok = true;
            if (p.pad_mode == MIT_PAD_REFLECT) {
                iy = iy < 0 ? -iy : (iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix);
            } else {
                ok = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            if (ok) off = (int)((int64_t)nb * p.a_bs + (int64_t)iy * p.a_ys + (int64_t)ix * p.a_xs + p.tap_off[tp]);
        }
        rowtab[idx] = off;
    }
    int b_src[B_ITERS], b_dst[B_ITERS];
    bool b_ok[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int c = tid + i * NTHR;
        const int pl = c / B_CPP, rem = c - pl * B_CPP;
        const int kh = rem / BN, n = rem - kh * BN;
        b_ok[i] = c < B_CELLS && (n0 + n) < ldn;
        b_src[i] = (pl * K8 + kh) * ldn + n0 + n;
        b_dst[i] = (pl * KH + kh) * SB + n;
    }
    int ld_tap = 0, ld_ci0 = 0, ld_k8 = 0;

    // two register sets: K-tile t+1 (being split in L(t)) and K-tile t+2 (in flight) — the loads of K-tile t+3 are issued into the set
    // the split has just emptied, two whole iterations ahead of their use (one workgroup per CU: nobody else covers an HBM miss)
    f32x4 a_regs[2][A_ITERS];
    u32x4 b_regs[2][B_ITERS];
    int a_offs[2][A_ITERS];

    __syncthreads();  // rowtab visible

    auto load_tile = [&](auto set_c) {  // branch-free: masked-off lanes read a valid dummy address, their values are replaced by zero in store_tile
        constexpr int S = decltype(set_c)::value;
        const int *rt = rowtab + ld_tap * BM + am;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) a_offs[S][i] = rt[i * A_MSTEP];
        const float *ak = a_thr + ld_ci0;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) a_regs[S][i] = *reinterpret_cast<const f32x4 *>(ak + (a_offs[S][i] < 0 ? 0 : a_offs[S][i]));
        const u32x4 *wk = ws + (int64_t)ld_k8 * ldn;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) b_regs[S][i] = wk[b_ok[i] ? b_src[i] : 0];
        ld_k8 += KH;
        ld_ci0 += BK;
        const bool wrap = ld_ci0 >= p.Cin;
        ld_ci0 = wrap ? 0 : ld_ci0;
        ld_tap += wrap ? 1 : 0;
    };
    auto store_tile = [&](const int buf, auto set_c) {
        constexpr int S = decltype(set_c)::value;
        f32x4 (&a_reg)[A_ITERS] = a_regs[S];
        u32x4 (&b_reg)[B_ITERS] = b_regs[S];
        int (&a_off)[A_ITERS] = a_offs[S];
        u32x2 *as2 = reinterpret_cast<u32x2 *>(As + buf * A_TILE);
        const int kh = aq >> 1, half = aq & 1;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int ml = (am + i * A_MSTEP) ^ split_swz<BK>(kh);
            u32x2 h, m, l;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if (X_NOSPLIT) {
                const u32x4 raw = __builtin_bit_cast(u32x4, a_reg[i]);
                h = u32x2{raw.x, raw.y}, m = u32x2{raw.z, raw.w}, l = u32x2{raw.x ^ raw.z, raw.y ^ raw.w};
            } else {
                split3<true>(a_off[i] < 0 ? zero : a_reg[i], h, m, l);
            }
            as2[((0 * KH + kh) * SA + ml) * 2 + half] = h;
            as2[((1 * KH + kh) * SA + ml) * 2 + half] = m;
            as2[((2 * KH + kh) * SA + ml) * 2 + half] = l;
        }
        u32x4 *bs = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i)
            if ((i + 1) * NTHR <= B_CELLS || tid + i * NTHR < B_CELLS) bs[b_dst[i]] = b_reg[i];
    };

    f32x16 acc[TM][TN];
    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;

    constexpr int EPI_FLOATS = (BM * (int)sizeof(RowOff) + 15) / 16 * 4 + (NTHR / 64) * 32 * EPI_PITCH + BM * (int)sizeof(LutOff) / 4;
    constexpr int STAGE_FLOATS = (2 * A_TILE + 2 * B_TILE) * 4;
    constexpr int SMEM_F = STAGE_FLOATS > EPI_FLOATS ? STAGE_FLOATS : EPI_FLOATS;

    const std::integral_constant<int, 0> set0;
    const std::integral_constant<int, 1> set1;
    load_tile(set0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    store_tile(0, set0);
    if (KT > 1) load_tile(set1);  // K-tiles 1 and 2 ride in the registers across the barrier
    if (KT > 2) load_tile(set0);
    __syncthreads();
    if (second) __builtin_amdgcn_s_barrier();  // the stagger

    bf16x8 af[3][TM], bf[3][TN];
    // one K-tile; NS = the register set holding K-tile kt + 1 = (kt + 1) & 1
    auto iter = [&](const int kt, auto ns_c) {
        const int cur = kt & 1;
        // ---- load segment
        const u32x4 *as = As + cur * A_TILE;
        const u32x4 *bs = Bs + cur * B_TILE + lh * SB + wn0 + li;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                af[pl][mi] = __builtin_bit_cast(bf16x8, as[(pl * KH + lh) * SA + ((wm0 + mi * 32 + li) ^ split_swz<BK>(lh))]);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[pl][ni] = __builtin_bit_cast(bf16x8, bs[(pl * KH) * SB + ni * 32]);
        }
        if (kt + 1 < KT) store_tile(cur ^ 1, ns_c);
        if (kt + 3 < KT) load_tile(ns_c);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute segment
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        if (!X_NOMFMA) {
#pragma unroll
            for (int pr = 3; pr < 9; ++pr)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kSplitPA[pr]][mi], bf[kSplitPB[pr]][ni], acc[mi][ni], 0, 0, 0);
        } else {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) asm volatile("" ::"v"(af[pl][mi]));
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) asm volatile("" ::"v"(bf[pl][ni]));
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // nothing of this segment touched memory: a bare barrier
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        int kt = 0;
        for (; kt + 1 < KT; kt += 2) {
            iter(kt, set1);
            iter(kt + 1, set0);
        }
        if (kt < KT) iter(kt, set1);
    }
    if (!second) __builtin_amdgcn_s_barrier();  // the halves meet again
    __syncthreads();                            // the epilogue reuses the staging area
    const int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    epilogue<BM, TM, TN, 0, SMEM_F, NTHR>(p, acc, smem, M, m0, n0, wm0, wn0, z1, z0, HoWo, tid_e);
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW, int NPROD, int VAR = 0>
void launch_split_pp(const MitConvGemm &p, int M, int MT, int NT, int KT, hipStream_t s) {
    constexpr int KH = BK / 8, NW = 8;
    size_t staging = (size_t)(2 * 3 * KH * BM + 2 * 3 * KH * BN) * 16 + (size_t)p.ntaps * BM * sizeof(int);
    size_t rows = ((size_t)BM * sizeof(RowOff) + 15) / 16 * 16 + (size_t)NW * 32 * EPI_PITCH * sizeof(float) + (size_t)BM * sizeof(LutOff);
    size_t smem = staging > rows ? staging : rows;
    auto kern = conv_gemm_split_pp_kernel<BM, BN, BK, WAVES_M, WAVES_N, MINW, NPROD, VAR>;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(kern), smem);
    dim3 grid(MT * NT, p.Z, 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, s, p, M, MT, NT, KT);
}

}  // namespace mitcg
