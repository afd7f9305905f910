// conv_gemm_split.h — the split-bf16 tiles of mit_conv_gemm (GEMM mode 6, the default, and 9: mit_gemm_mode_set / MIT_GEMM_SPLIT): kernel,
// weight packer and launcher.  No kernel here may spill to scratch memory (check .amdhsa_private_segment_fixed_size after changes).
// Included at the end of conv_gemm_kernels.h (it shares that header's epilogue, RowOff and launch conventions); instantiated by
// conv_gemm_inst5 / 6 / 7.hip through conv_gemm_cfgs.inc.
#pragma once
#include "bf16_split.h"

namespace mitcg {

// ---- split-bf16 tiles: the fp32 contraction on v_mfma_f32_32x32x16_bf16 (16x the rate of the fp32 MFMA) --------------------
// x = hi + mid + lo exactly, each term the round-to-nearest bf16 of what the previous ones left (24 = 8 + 8 + 8 significant
// bits; the residuals are exact in fp32), so a * b = sum_{p, q} a_p * b_q with every term exact in fp32.  NPROD = 9 keeps all nine
// plane pairs (error = fp32 accumulation only), 6 drops the pairs with p + q >= 3 (<= 2^-24 of the product each), 3 keeps p + q <= 1
// (a 16-bit-significand product: test ladder only).  Pairs are issued smallest first.
//   A (activations): gathered as in conv_gemm_fast_kernel (rowtab, float4 per lane), split in registers when the tile is written to
//     LDS: per plane, 16-byte cells [kh][row] of 8 consecutive k — exactly one lane's A operand (row = lane & 31, k group = lane >> 5).
//   W: split once by mit_gemm_split_pack into the same cells [plane][k / 8][n][8]; a K-tile is 3 * (BK / 8) * BN cells copied 16
//     bytes per lane, global -> VGPR -> LDS.
// The MFMA pairs A element j of k-group g with B element j of k-group g, so only the (row | column, k-group) placement matters and
// both operands use the same one.  Accumulators have the 32x32 C layout of the fp32 tiles: the epilogue is shared.
// Range precondition: finite operands with |x| <= the largest bf16 (3.39e38).  hi = bf16(x) of a larger (or infinite) x is Inf and
// the residual x - Inf is -Inf / NaN, so such an operand yields NaN here where the fp32 MFMA yields Inf or a finite value.  Activations
// and weights of the networks on this path are many orders of magnitude inside the range.
// (plane pairs kSplitPA / kSplitPB, smallest products first; NPROD takes the last NPROD entries: bf16_split.h)

// A cells are row-swizzled per k slab (row ^ kh * 64 / BK) instead of padded.  ds_write_b64 is served in groups of 16 consecutive lanes
// over 32 four-byte banks (MI355X_MICROARCH.md, LDS table): a group's 16 eight-byte halves — 16 / KQ rows x all KH slabs x 2 halves —
// must fall on 16 different bank pairs, i.e. (row ^ swz) mod 8 must differ between the slabs: kh * 8 / KH.  (Until round 6 the swizzle
// was kh * 32 / KQ, laid out for 64 banks: slab 1 landed on slab 0's banks — a 2-way conflict on every A store, 19 % of the kernel's
// LDS cycles in SQ_LDS_BANK_CONFLICT, profiles/r10p_pmc_split_tile.json.)  A ds_read_b128 lane group still reads 16 distinct cells
// of one aligned 256-byte run (the XOR permutes rows inside aligned blocks of 8).
template <int BK>
__device__ __forceinline__ constexpr int split_swz(int kh) { return kh * (64 / BK); }

// VAR bit 1: software pipeline — the next tile's operands (loaded one iteration ahead) are split and written to the other LDS buffer
// among this tile's MFMAs, and the loads of the tile after it are issued behind them.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW, int NPROD, int VAR = 0>
__global__ __launch_bounds__(256, MINW) void conv_gemm_split_kernel(const MitConvGemm p, const int M, const int MT, const int NT,
                                                                  const int KT) {
    constexpr int WM = BM / WAVES_M;
    constexpr int WN = BN / WAVES_N;
    constexpr int TM = WM / 32;
    constexpr int TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "wave tile");
    static_assert(BK == 16 || BK == 32, "K-tile of one or two bf16 MFMA steps");
    static_assert(NPROD == 3 || NPROD == 6 || NPROD == 9, "plane pairs");
    constexpr int KH = BK / 8;   // 16-byte cells along k
    constexpr int KS = BK / 16;  // MFMA steps per K-tile
    constexpr int KQ = BK / 4;   // float4 chunks along k of the fp32 A tile
    constexpr int A_ITERS = BM * KQ / 256;
    constexpr int A_MSTEP = 256 / KQ;
    static_assert(A_ITERS >= 1 && (BM * KQ) % 256 == 0, "A tile must fill the workgroup");
    constexpr bool PIPE = (VAR & 1) != 0;
    // timing ablations (WRONG results; scripts/split_check --ablate only): skip the in-loop global loads / the split arithmetic / the LDS
    // writes / the fragment reads / the barrier
    constexpr bool X_NOLOAD = (VAR & 2) != 0, X_NOSPLIT = (VAR & 4) != 0, X_NOWRITE = (VAR & 8) != 0, X_NOFRAG = (VAR & 16) != 0,
                   X_NOBAR = (VAR & 32) != 0;
    constexpr bool MID = (VAR & 128) != 0;  // with PIPE: the staging cut into steps, one placed behind each MFMA (sched_barrier keeps them there)
    constexpr bool ORD = (VAR & 256) != 0;  // with MID: fragment reads issued in the order the plane pairs consume them; next row offsets fetched in step 0
    // (multi-tile forms — 2 / 4 consecutive output tiles per workgroup, and a grid-strided persistent form with the next tile's set-up
    // issued ahead of the epilogue — were measured in round 3 and removed: 0.45-1.02x of this form, profiles/r03d_split_check_experiments.log)
    // VAR bit 1024 (round 6 experiment): the W fragments do not pass through LDS at all — W planes are stored in MFMA-operand cells, so a
    // lane's B operand of (plane, k-group lh, column) is ONE 16-byte global load (L2-resident weights, 512 contiguous bytes per
    // half-wave); the fragments of K-tile t + 1 are requested at the top of tile t.  Takes the 13-cycle ds_write_b128 of W (half of the
    // VGPR -> LDS store traffic that bounds this tile) and half of the fragment reads off the LDS; costs W's L2 -> L1 traffic twice
    // (the two waves of a column read the same cells) and nine registers.
    constexpr bool BDIR = (VAR & 1024) != 0;
    // VAR bit 2048 (round 6): operand loads through BUFFER instructions — one resource descriptor per operand (SGPRs), a 32-bit byte
    // offset per lane, the K-tile cursor as the scalar offset.  Rows that contribute zeros carry offset 2^31, past the descriptor's 2^31
    // bytes, which the hardware's range check answers with zeros: no clamp, no compare, no select on the loaded values, no 64-bit address arithmetic — 16 of the
    // 77 VALU instructions a wave issues per K-tile beside its 24 MFMAs (profiles/r10p_pmc_split_tile.json).  The launcher takes it
    // only when every A and W byte offset is below 2^31 (buf_eligible in conv_gemm.hip).
    constexpr bool BUF = (VAR & 2048) != 0;
    constexpr bool UNR2 = (VAR & 4096) != 0;  // see tile()
    constexpr bool ASM_SUB = (VAR & 64) != 0;  // residuals through v_sub_f32 inline asm: keeps the SLP vectoriser from packing them into v_pk_add_f32
    constexpr int SA = BM, SB = BN;
    constexpr int A_TILE = 3 * KH * SA, B_TILE = 3 * KH * SB;  // cells per buffer
    constexpr int B_CPP = KH * BN;                             // W cells per plane per K-tile
    constexpr int B_CELLS = 3 * B_CPP;
    constexpr int B_ITERS = (B_CELLS + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4 *As = reinterpret_cast<u32x4 *>(smem);  // [2][3][KH][SA]
    u32x4 *Bs = As + 2 * A_TILE;                  // [2][3][KH][SB]
    int *rowtab = reinterpret_cast<int *>(Bs + 2 * B_TILE);  // [ntaps][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR (wm0 / wn0 with it)
    const int li = lane & 31;
    const int lh = lane >> 5;

    const int nwg = MT * NT;
    int cur_tile;  // block id -> position in the XCD-contiguous order: consecutive tiles (n fastest) share the A panel
    {
        const int v = blockIdx.x, xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        cur_tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    int m0 = 0, n0 = 0;
    const int z = blockIdx.y;
    const int z1 = z / p.zdiv, z0 = z - z1 * p.zdiv;
    const int HoWo = p.Ho * p.Wo;

    const float *__restrict__ a_base = p.a + z1 * p.a_zs1 + z0 * p.a_zs0 + (p.dyn ? (int64_t)(*p.dyn) * p.a_dyn : 0);
    const u32x4 *__restrict__ ws = reinterpret_cast<const u32x4 *>(p.w_split + z0 * p.ws_zs0);
    const int K8 = p.Kw >> 3;     // cells along k per plane
    const int ldn = (int)p.ldw;   // cells per k-row

    const int aq = tid % KQ;
    const int am = tid / KQ;
    const float *__restrict__ a_thr = a_base + aq * 4;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a_base), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(ws), 0, 0x80000000u, 0x00020000);
    const int aq16 = aq * 16;
    // this thread's W cells: the same (plane, kh, n) in every K-tile of an output tile.  B_UNI (a plane of the K-tile is exactly one
    // cell per thread: the 128 x 128 x 16 tile): the thread's cells differ only by the plane, so one source index, one LDS index and one
    // flag stand for all of them (the other planes are a wave-uniform stride away) — six registers less on the tile that needs them
    constexpr bool B_UNI = B_CPP == 256;
    constexpr int B_IDX = B_UNI ? 1 : B_ITERS;
    int b_src[B_IDX], b_dst[B_IDX];
    bool b_ok[B_IDX];
    auto bsrc = [&](const int i) { return B_UNI ? b_src[0] + i * K8 * ldn : b_src[B_UNI ? 0 : i]; };
    auto bdst = [&](const int i) { return B_UNI ? b_dst[0] + i * KH * SB : b_dst[B_UNI ? 0 : i]; };
    auto bok = [&](const int i) { return b_ok[B_UNI ? 0 : i]; };
    int ld_tap = 0, ld_ci0 = 0, ld_k8 = 0;  // (tap, first channel, first k cell) of the tile being loaded: wave-uniform
    auto setup_tile = [&](const int t) {  // gather table, W cell addresses and load cursor of output tile t
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
            rowtab[idx] = BUF ? (off < 0 ? (int)0x80000000u : off * 4) : off;  // BUF: byte offsets; 2^31 (+ a lane's k offset) is past the descriptor
        }
#pragma unroll
        for (int i = 0; i < B_IDX; ++i) {
            const int c = tid + i * 256;
            const int pl = c / B_CPP, rem = c - pl * B_CPP;
            const int kh = rem / BN, n = rem - kh * BN;
            b_ok[i] = c < B_CELLS && (n0 + n) < ldn;
            b_src[i] = (pl * K8 + kh) * ldn + n0 + n;
            b_dst[i] = (pl * KH + kh) * SB + n;
        }
        ld_tap = ld_ci0 = ld_k8 = 0;
    };
    setup_tile(cur_tile);

    f32x4 a_reg[A_ITERS];
    u32x4 b_reg[B_ITERS];
    int a_off[A_ITERS];

    __syncthreads();  // rowtab visible

    // Branch-free on purpose: masked-off lanes read a valid dummy address and the value is replaced by zero afterwards, so a loop
    // iteration stays ONE basic block and the scheduler can place these loads, the split and the LDS writes among the MFMAs.
    auto load_tile = [&]() {
        const int *rt = rowtab + ld_tap * BM + am;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) a_off[i] = rt[i * A_MSTEP];
        const float *ak = a_thr + ld_ci0;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {  // rows that contribute zeros are masked when the tile is split (store_tile), not here: a select
            if (BUF) a_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, a_off[i] + aq16, ld_ci0 * 4, 0));
            else a_reg[i] = *reinterpret_cast<const f32x4 *>(ak + (a_off[i] < 0 ? 0 : a_off[i]));  // on the loaded value would wait for the load
        }
        const u32x4 *wk = ws + (int64_t)ld_k8 * ldn;  // split_eligible(): every K-tile lies inside the packed planes
        if (!BDIR) {
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) {
                if (BUF) b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, bok(i) ? bsrc(i) * 16 : -1, ld_k8 * ldn * 16, 0);
                else b_reg[i] = wk[bok(i) ? bsrc(i) : 0];  // columns past ldw get arbitrary finite-or-not values: never stored
            }
        }
        ld_k8 += KH;
        ld_ci0 += BK;
        const bool wrap = ld_ci0 >= p.Cin;
        ld_ci0 = wrap ? 0 : ld_ci0;
        ld_tap += wrap ? 1 : 0;
    };
    auto store_tile = [&](int buf) {
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
                split3<ASM_SUB>(!BUF && a_off[i] < 0 ? zero : a_reg[i], h, m, l);
            }
            as2[((0 * KH + kh) * SA + ml) * 2 + half] = h;
            as2[((1 * KH + kh) * SA + ml) * 2 + half] = m;
            as2[((2 * KH + kh) * SA + ml) * 2 + half] = l;
        }
        u32x4 *bs = Bs + buf * B_TILE;
        if (!BDIR) {
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i)
                if ((i + 1) * 256 <= B_CELLS || tid + i * 256 < B_CELLS) bs[bdst(i)] = b_reg[i];
        }
    };

    f32x16 acc[TM][TN];

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;

    // The staging of one tile as a sequence of small steps (MID): per A chunk three (pack a plane, write it, form the residual), one
    // per W cell write, then the loads of the following tile (row offsets, A chunks, W cells).
    constexpr int STAGE_WRITE_STEPS = 3 * A_ITERS + B_ITERS, STAGE_STEPS = STAGE_WRITE_STEPS + 1 + A_ITERS + B_ITERS;
    auto stage_step = [&](const int st, const int buf, const bool with_loads) {
        if (ORD && st == 3 * (A_ITERS - 1) + 1 && with_loads) {  // the last use of the current offsets (the zero mask of the last chunk) is behind:
            const int *rt = rowtab + ld_tap * BM + am;            // the next tile's take their registers, ahead of most of this tile's LDS writes
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) a_off[i] = rt[i * A_MSTEP];
        }
        if (st < 3 * A_ITERS) {
            const int i = st / 3, ph = st % 3;
            const int kh = aq >> 1, half = aq & 1;
            // one index register for every chunk and plane: the swizzle only touches row bits below A_MSTEP, so chunk i sits
            // i * A_MSTEP rows (an immediate) behind chunk 0
            static_assert((A_MSTEP & (A_MSTEP - 1)) == 0 && (BK / 4) * (32 / (BK / 4)) <= A_MSTEP || A_ITERS == 1, "row swizzle must stay inside a chunk's rows");
            const int ml = ((am ^ split_swz<BK>(kh)) + i * A_MSTEP);
            f32x4 &r = a_reg[i];  // the residuals replace the loaded values in place: the chunk is reloaded only after its last split step
            if (ph == 0 && !BUF) {
                const unsigned int keep = ~(unsigned int)(a_off[i] >> 31);  // rows that contribute zeros (offset -1): all bits cleared
                r.x = __uint_as_float(__float_as_uint(r.x) & keep);
                r.y = __uint_as_float(__float_as_uint(r.y) & keep);
                r.z = __uint_as_float(__float_as_uint(r.z) & keep);
                r.w = __uint_as_float(__float_as_uint(r.w) & keep);
            }
            const u32x2 pk = {pack_bf16(r.x, r.y), pack_bf16(r.z, r.w)};
            reinterpret_cast<u32x2 *>(As + buf * A_TILE)[((ph * KH + kh) * SA + ml) * 2 + half] = pk;
            if (ph < 2) {
                r.x = sub_f32<ASM_SUB>(r.x, bf16_lo(pk.x));
                r.y = sub_f32<ASM_SUB>(r.y, bf16_hi(pk.x));
                r.z = sub_f32<ASM_SUB>(r.z, bf16_lo(pk.y));
                r.w = sub_f32<ASM_SUB>(r.w, bf16_hi(pk.y));
            }
        } else if (st < STAGE_WRITE_STEPS) {
            const int j = st - 3 * A_ITERS;
            if (!BDIR && ((j + 1) * 256 <= B_CELLS || tid + j * 256 < B_CELLS)) (Bs + buf * B_TILE)[bdst(j)] = b_reg[j];
        } else if (st == STAGE_WRITE_STEPS) {
            if (!ORD) {
                const int *rt = rowtab + ld_tap * BM + am;
#pragma unroll
                for (int i = 0; i < A_ITERS; ++i) a_off[i] = rt[i * A_MSTEP];
            }
        } else if (st <= STAGE_WRITE_STEPS + A_ITERS) {
            const int i = st - STAGE_WRITE_STEPS - 1;
            if (BUF) a_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, a_off[i] + aq16, ld_ci0 * 4, 0));
            else a_reg[i] = *reinterpret_cast<const f32x4 *>(a_thr + ld_ci0 + (a_off[i] < 0 ? 0 : a_off[i]));
        } else if (st < STAGE_STEPS) {
            const int j = st - STAGE_WRITE_STEPS - 1 - A_ITERS;
            if (!BDIR) {
                if (BUF) b_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, bok(j) ? bsrc(j) * 16 : -1, ld_k8 * ldn * 16, 0);
                else b_reg[j] = (ws + (int64_t)ld_k8 * ldn)[bok(j) ? bsrc(j) : 0];
            }
            if (st == STAGE_STEPS - 1) {
                ld_k8 += KH;
                ld_ci0 += BK;
                const bool wrap = ld_ci0 >= p.Cin;
                ld_ci0 = wrap ? 0 : ld_ci0;
                ld_tap += wrap ? 1 : 0;
            }
        }
    };

    bf16x8 af[KS][3][TM], bf[KS][3][TN];
    // BDIR: the W fragments of the next K-tile, straight from the packed planes: cell (plane, k cell 2 ks + lh of the tile, column)
    u32x4 bnext[BDIR ? KS : 1][3][BDIR ? TN : 1];
    int bcol[BDIR ? TN : 1];  // this lane's cell of tile 0, plane 0, per 32-column block (columns past ldw: a valid duplicate, never stored)
    if (BDIR) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn0 + ni * 32 + li;
            bcol[ni] = lh * ldn + (n < ldn ? n : ldn - 1);
        }
    }
    auto load_bnext = [&](const int kt) {
        const u32x4 *wk = ws + (int64_t)kt * KH * ldn;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bnext[ks][pl][ni] = wk[(int64_t)(pl * K8 + 2 * ks) * ldn + bcol[ni]];
    };
    // One K-tile: fragment reads, then (PIPE) the staging of tile kt + 1 and the loads of tile kt + 2 scheduled among the MFMAs.
    auto tile = [&](const int kt, auto do_store, auto do_load, auto par_c) {
        constexpr bool DO_STORE = decltype(do_store)::value, DO_LOAD = decltype(do_load)::value;
        // VAR bit 4096: the steady loop runs two K-tiles per trip with the LDS buffer parity as a compile-time constant, so that
        // every ds_read / ds_write address is a loop-invariant register plus an immediate (the run-time parity cost a multiply and
        // an add3 per access: 11 VALU and 5 SALU per K-tile)
        constexpr int PAR = decltype(par_c)::value;
        const int cur = PAR < 0 ? (kt & 1) : PAR;
        if (!PIPE && DO_STORE && !X_NOLOAD) load_tile();
        const u32x4 *as = As + cur * A_TILE;
        const u32x4 *bs = Bs + cur * B_TILE + lh * SB + wn0 + li;
        if (!X_NOFRAG || kt == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const int pa = ORD ? (o == 0 ? 0 : 3 - o) : o;  // consumption order of the pairs: A planes 0, 2, 1 with W planes 2, 0, 1
                    const int pb = ORD ? (o == 0 ? 2 : o - 1) : o;
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        af[ks][pa][mi] = __builtin_bit_cast(bf16x8, as[(pa * KH + 2 * ks + lh) * SA + ((wm0 + mi * 32 + li) ^ split_swz<BK>(2 * ks + lh))]);
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        bf[ks][pb][ni] = BDIR ? __builtin_bit_cast(bf16x8, bnext[BDIR ? ks : 0][pb][BDIR ? ni : 0]) : __builtin_bit_cast(bf16x8, bs[(pb * KH + 2 * ks) * SB + ni * 32]);
                }
        }
        if (BDIR && DO_STORE) load_bnext(kt + 1);  // (the fragments just taken are copies: their registers are free for the next tile's)
        if (PIPE && MID) {
            constexpr int NM = KS * NPROD * TM * TN;
            constexpr int NSTEP = DO_STORE ? (DO_LOAD ? STAGE_STEPS : STAGE_WRITE_STEPS) : 0;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int ni = i % TN, mi = (i / TN) % TM, pr = 9 - NPROD + (i / (TN * TM)) % NPROD, ks = i / (TN * TM * NPROD);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][kSplitPA[pr]][mi], bf[ks][kSplitPB[pr]][ni], acc[mi][ni], 0, 0, 0);
                if (i >= 1 && i - 1 < NSTEP) stage_step(i - 1, cur ^ 1, DO_LOAD);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int st = NM - 1; st < NSTEP; ++st) stage_step(st, cur ^ 1, DO_LOAD);  // more steps than MFMAs (3-pair tiles)
            if (!X_NOBAR) __syncthreads();
            return;
        }
        if (PIPE && DO_STORE && !X_NOWRITE) store_tile(cur ^ 1);  // tile kt + 1 (in registers since the previous iteration) -> the other buffer
        if (PIPE && DO_LOAD && !X_NOLOAD) load_tile();           // tile kt + 2 on its way while this tile's MFMAs issue
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pr = 9 - NPROD; pr < 9; ++pr)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][kSplitPA[pr]][mi], bf[ks][kSplitPB[pr]][ni], acc[mi][ni], 0, 0, 0);
        if (PIPE && DO_STORE) {
            // wanted order: the fragment reads, then per MFMA a few VALU of the split, an LDS write every other MFMA, the global
            // loads behind the second half of the MFMAs
            constexpr int NM = KS * NPROD * TM * TN;
            __builtin_amdgcn_sched_group_barrier(0x100, KS * 3 * (TM + TN), 0);  // DS read
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
                if (i & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 // DS write
                if (DO_LOAD && i >= NM / 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
            }
        }
        if (!PIPE && DO_STORE && !X_NOWRITE) store_tile(cur ^ 1);
        if (!X_NOBAR) __syncthreads();
    };
    const std::integral_constant<bool, true> yes;
    const std::integral_constant<bool, false> no;
    const std::integral_constant<int, 0> par0;
    const std::integral_constant<int, 1> par1;
    const std::integral_constant<int, -1> parx;

    constexpr int EPI_FLOATS = (BM * (int)sizeof(RowOff) + 15) / 16 * 4 + 4 * 32 * EPI_PITCH + BM * (int)sizeof(LutOff) / 4;
    constexpr int STAGE_FLOATS = (2 * A_TILE + 2 * B_TILE) * 4;
    constexpr int SMEM_F = STAGE_FLOATS > EPI_FLOATS ? STAGE_FLOATS : EPI_FLOATS;

    load_tile();
    if (BDIR) load_bnext(0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    store_tile(0);
    if (PIPE && KT > 1) load_tile();  // tile 1 rides in the registers across the barrier
    __syncthreads();
    {
        int kt = 0;
        if (PIPE) {
            if (UNR2) {
                for (; kt + 3 < KT; kt += 2) {
                    tile(kt, yes, yes, par0);
                    tile(kt + 1, yes, yes, par1);
                }
            }
            for (; kt + 2 < KT; ++kt) tile(kt, yes, yes, parx);
            if (kt + 1 < KT) tile(kt++, yes, no, parx);
        } else {
            for (; kt + 1 < KT; ++kt) tile(kt, yes, no, parx);
        }
        tile(kt, no, no, parx);
    }
    if (X_NOBAR) __syncthreads();  // the epilogue reuses the staging area
    // the thread index rebuilt from the SGPR wave index and mbcnt: nothing derived from threadIdx.x has to survive the K loop (the
    // 128 x 128 tile otherwise spilled eight such registers to scratch memory; see DESIGN §7 on why no kernel of this library may use
    // scratch)
    const int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    epilogue<BM, TM, TN, 0, SMEM_F>(p, acc, smem, M, m0, n0, wm0, wn0, z1, z0, HoWo, tid_e);
}

// (A one-wave-per-32x32-block form with register-streamed operands for few-row GEMMs — a page's decoder Linears, M = 160 — was built and
// measured in round 3: bit-identical to the tiles but 10-20 % SLOWER than the 64 x 64 tile at every decoder shape.  Prefetch depth 4 / 8 /
// 12 made no difference (not load latency) and 2 / 3 / 5 row blocks per wave scaled the time linearly (0.38 us per K-step and block: every
// wave re-splits its A rows in registers — ~60 VALU per 8 floats — for 6 MFMAs); removed.  profiles/r03i_split_check_small_tiles.log)

// W [nz][Kw][ldw] fp32 -> [nz][3][Kw / 8][ldw][8] bf16 (see conv_gemm_split_kernel); one thread per (slice, k cell, column)
static __global__ __launch_bounds__(256) void gemm_split_pack_kernel(const float *__restrict__ w, const int64_t w_zs, const int K8, const int ldw,
                                                              uint16_t *__restrict__ out, const int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx % ldw);
    const int64_t t = idx / ldw;
    const int k8 = (int)(t % K8);
    const int64_t z = t / K8;
    const float *src = w + z * w_zs + (int64_t)k8 * 8 * ldw + n;
    unsigned int h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = src[(int64_t)(2 * j) * ldw], x1 = src[(int64_t)(2 * j + 1) * ldw];
        h[j] = pack_bf16(x0, x1);
        const float r0 = x0 - bf16_lo(h[j]), r1 = x1 - bf16_hi(h[j]);
        m[j] = pack_bf16(r0, r1);
        l[j] = pack_bf16(r0 - bf16_lo(m[j]), r1 - bf16_hi(m[j]));
    }
    const int64_t plane = (int64_t)K8 * ldw;  // cells
    u32x4 *o = reinterpret_cast<u32x4 *>(out) + z * 3 * plane + (int64_t)k8 * ldw + n;
    o[0] = u32x4{h[0], h[1], h[2], h[3]};
    o[plane] = u32x4{m[0], m[1], m[2], m[3]};
    o[2 * plane] = u32x4{l[0], l[1], l[2], l[3]};
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MINW, int NPROD, int VAR = 0>
void launch_split(const MitConvGemm &p, int M, int MT, int NT, int KT, hipStream_t s) {
    constexpr int KH = BK / 8;
    constexpr int SA = BM, SB = BN;
    size_t staging = (size_t)(2 * 3 * KH * SA + 2 * 3 * KH * SB) * 16 + (size_t)p.ntaps * BM * sizeof(int);
    size_t rows = ((size_t)BM * sizeof(RowOff) + 15) / 16 * 16 + (size_t)4 * 32 * EPI_PITCH * sizeof(float) + (size_t)BM * sizeof(LutOff);
    size_t smem = staging > rows ? staging : rows;
    auto kern = conv_gemm_split_kernel<BM, BN, BK, WAVES_M, WAVES_N, MINW, NPROD, VAR>;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(kern), smem);
    dim3 grid(MT * NT, p.Z, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p, M, MT, NT, KT);
}

}  // namespace mitcg
