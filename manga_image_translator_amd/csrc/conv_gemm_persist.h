// conv_gemm_persist.h — grid-strided multi-tile form of conv_gemm_fast_kernel's default variant (VAR = 4: gather offsets cached per tap,
// incremental weight pointer).  NOT part of the default build's tile table: configurations exist only under MIT_CONV_EXPERIMENTS until the
// form has been measured (working notes).  Same tiling, LDS images, k-sequential accumulation order and epilogue as the fast
// kernel, so results are bitwise identical to it; what changes is the life of a workgroup:
//   * the launcher starts one workgroup per residency slot of the chip (a multiple of 8); workgroup b takes the virtual block ids
//     b, b + gridDim.x, ... — congruent mod 8, i.e. always on b's XCD — and maps each through the usual XCD-contiguous remap;
//   * before a tile's epilogue the next tile's gather table is built and its first K-tile is loaded into the staging registers (free
//     once the K loop has ended), so the prologue's global-load latency hides behind the epilogue's stores.
// Included at the end of conv_gemm_kernels.h.
#pragma once

namespace mitcg {

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_gemm_fast_persist_kernel(const MitConvGemm p, const int M, const int MT, const int NT,
                                                                         const int KT) {
    constexpr int WM = BM / WAVES_M;
    constexpr int WN = BN / WAVES_N;
    constexpr int TM = WM / 32;
    constexpr int TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "wave tile");
    constexpr int KQ = BK / 4;
    constexpr int A_ITERS = BM * KQ / 256;
    constexpr int A_MSTEP = 256 / KQ;
    constexpr int NQ = BN / 4;
    constexpr int B_ITERS = BK * NQ / 256;
    constexpr int B_KSTEP = 256 / NQ;
    static_assert(A_ITERS >= 1 && (BM * KQ) % 256 == 0, "A tile must fill the workgroup");
    static_assert(B_ITERS >= 1 && (BK * NQ) % 256 == 0, "B tile must fill the workgroup");
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
    auto remap = [&](int v) {  // virtual block id -> position in the XCD-contiguous tile order
        const int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    };
    int vblock = blockIdx.x;
    const int z = blockIdx.y;
    const int z1 = z / p.zdiv, z0 = z - z1 * p.zdiv;
    const int HoWo = p.Ho * p.Wo;

    const float *__restrict__ a_base = p.a + z1 * p.a_zs1 + z0 * p.a_zs0;
    const float *__restrict__ w_base = p.w + z1 * p.w_zs1 + z0 * p.w_zs0;

    const int aq = tid % KQ;
    const int am = tid / KQ;
    const int bn4 = tid % NQ;
    const int bk = tid / NQ;
    const float *__restrict__ a_thr = a_base + aq * 4;
    const int64_t w_kstep = (int64_t)B_KSTEP * p.ldw, w_tstep = (int64_t)BK * p.ldw;

    int m0 = 0, n0 = 0;
    bool b_ncol_ok = false;
    const float *__restrict__ w_row = w_base;  // row (kt * BK + bk) of this thread's weight column
    int ld_tap = 0, ld_ci0 = 0, ld_kt = 0;     // (tap, first channel, K-tile index) of the tile being loaded: wave-uniform
    auto setup_tile = [&](const int t) {
        const int mt = t / NT, nt = t - mt * NT;
        m0 = mt * BM;
        n0 = nt * BN;
        for (int idx = tid; idx < p.ntaps * BM; idx += 256) {
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
        b_ncol_ok = (n0 + bn4 * 4) < p.Nw;
        w_row = w_base + n0 + bn4 * 4 + (int64_t)bk * p.ldw;
        ld_tap = ld_ci0 = ld_kt = 0;
    };

    f32x4 a_reg[A_ITERS];
    f32x4 b_reg[B_ITERS];
    int a_off[A_ITERS];
    auto load_tile = [&]() {
        const float *ak = a_thr + ld_ci0;
        if (ld_ci0 == 0) {  // wave-uniform: the row offsets only change with the tap
            const int *rt = rowtab + ld_tap * BM + am;
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) a_off[i] = rt[i * A_MSTEP];
        }
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a_off[i] >= 0) v = *reinterpret_cast<const f32x4 *>(ak + a_off[i]);
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int k = ld_kt * BK + bk + i * B_KSTEP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ncol_ok && k < p.Kw) v = *reinterpret_cast<const f32x4 *>(w_row + i * w_kstep);
            b_reg[i] = v;
        }
        w_row += w_tstep;
        ++ld_kt;
        ld_ci0 += BK;
        if (ld_ci0 >= p.Cin) {
            ld_ci0 = 0;
            ++ld_tap;
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
            *reinterpret_cast<f32x4 *>(bs + kl * LDB + bn4 * 4) = b_reg[i];
        }
    };

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;
    constexpr int EPI_FLOATS = (BM * (int)sizeof(RowOff) + 15) / 16 * 4 + 4 * 32 * EPI_PITCH;
    constexpr int SMEM_F = (2 * A_TILE + 2 * B_TILE) > EPI_FLOATS ? (2 * A_TILE + 2 * B_TILE) : EPI_FLOATS;

    f32x16 acc[TM][TN];
    setup_tile(remap(vblock));
    __syncthreads();  // rowtab visible
    load_tile();
    for (;;) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            const float *as = As + cur * A_TILE + lh * LDA + wm0 + li;
            const float *bs = Bs + cur * B_TILE + lh * LDB + wn0 + li;
            float af[2][TM], bf[2][TN];
            if (kt + 1 < KT) load_tile();
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[0][mi] = as[mi * 32];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[0][ni] = bs[ni * 32];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int c = ks & 1;
                if (ks + 1 < BK / 2) {
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) af[c ^ 1][mi] = as[(2 * ks + 2) * LDA + mi * 32];
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) bf[c ^ 1][ni] = bs[(2 * ks + 2) * LDB + ni * 32];
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][mi], bf[c][ni], acc[mi][ni], 0, 0, 0);
            }
            if (kt + 1 < KT) store_tile(cur ^ 1);
            __syncthreads();
        }
        const int em0 = m0, en0 = n0;
        vblock += gridDim.x;
        const bool more = vblock < nwg;  // workgroup-uniform
        if (more) {  // the gather table is no longer read (the K loop ended on a barrier); the staging registers are free
            setup_tile(remap(vblock));
            __syncthreads();
            load_tile();  // in flight during the epilogue below
        }
        epilogue<BM, TM, TN, 0, SMEM_F>(p, acc, smem, M, em0, en0, wm0, wn0, z1, z0, HoWo);
        if (!more) break;
        __syncthreads();  // the epilogue's row table / transpose buffers live in the staging area the next store_tile(0) overwrites
    }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW>
void launch_fast_persist(const MitConvGemm &p, int M, int MT, int NT, int KT, hipStream_t s) {
    constexpr int LDA = BM + (BK == 16 ? 2 : 1);
    constexpr int LDB = BN + 4;
    size_t staging = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float) + (size_t)p.ntaps * BM * sizeof(int);
    size_t rows = ((size_t)BM * sizeof(RowOff) + 15) / 16 * 16 + (size_t)4 * 32 * EPI_PITCH * sizeof(float);
    size_t smem = staging > rows ? staging : rows;
    auto kern = conv_gemm_fast_persist_kernel<BM, BN, BK, WAVES_M, WAVES_N, MINW>;
    static bool attr_set = false;
    if (!attr_set && smem > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    static int slots = 0;  // one workgroup per residency slot, a multiple of 8 so that a workgroup's tiles stay on its XCD
    if (!slots) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 256, smem) != hipSuccess || per_cu < 1) per_cu = 2;
        (void)hipGetDevice(&dev);
        const int cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        slots = (per_cu * cus) / 8 * 8;
        if (slots < 8) slots = 8;
    }
    int gx = MT * NT;
    if (gx > slots) gx = slots;
    dim3 grid(gx, p.Z, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p, M, MT, NT, KT);
}

}  // namespace mitcg
