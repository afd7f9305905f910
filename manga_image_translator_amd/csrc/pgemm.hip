// pgemm.hip — mit_pgemm: the plain GEMMs of the split-bf16 mode on operands that ARRIVE as three bf16 planes (include/mit_hip.h,
// "planar operands"), plus the stand-alone plane producer / joiner (mit_split_planes / mit_join_planes).
//
// What the K loop of conv_gemm_split_kernel does per K-tile and this kernel does not: gather-table reads, fp32 global loads into
// registers, ~60 VALU instructions of splitting per 8 activations, ds_write of both operands, two barriers around one LDS buffer pair.
// Here a K-tile (16 deep) of BOTH operands is 3 planes x 2 k-cells x (BM | BN) 16-byte cells that already have the layout of the
// v_mfma_f32_32x32x16_bf16 operands (cell (k8, r) = lane (r & 31, k8 & 1)'s eight bf16), so it goes global -> LDS by the LDS-DMA
// (global_load_lds_dwordx4: one KiB per wave-instruction, lane-linear in LDS, no VGPR and no ds_write), into a ring of NS stages:
//
//   iteration j:   s_waitcnt vmcnt((NS-2) G)        the DMA of K-tile j (mine) has landed; K-tiles j+1 .. j+NS-2 stay in flight
//                  s_barrier                        ... and everybody's; everybody is done with the fragments of K-tile j-1
//                  [epilogue of the finished output tile, if K-tile j-1 was its last]
//                  issue the G DMA pieces of K-tile j+NS-1 into the stage K-tile j-1 occupied
//                  6 (+6) ds_read_b128 fragment reads, 6 TM TN (9 TM TN) MFMAs
//
// ONE barrier per K-tile and no vmcnt(0) in steady state.  The iteration space is flat over the output tiles a workgroup owns (a
// contiguous, n-fastest run of tiles, XCD-contiguous: block b runs on XCD b % 8), so the loads of the next tile's first K-tiles are
// already in flight while a tile's epilogue runs; two such workgroups share a CU and cover each other's epilogues.
// Stores of an epilogue also count on vmcnt, in an order relative to the DMA pieces that is not relied upon: an iteration that runs an
// epilogue first drains vmcnt(0) (everything it waits for was issued at least one K-tile earlier), the NS-2 iterations after it need
// no wait at all, and from then on the counted wait is exact again whatever order the stores retire in (see wait logic below).
//
// Arithmetic: per 16-deep k step and accumulator the plane pairs in the order of the split tiles (kSplitPA / kSplitPB, smallest
// products first), fp32 accumulation in the MFMA: bit-identical to conv_gemm_split_kernel for the same operands.
// Output: fp32 row-major through a per-wave LDS transpose (the free stage of the ring), or planes again (OUTP = 1): the MFMA operands
// are swapped so that a lane holds 4 consecutive columns of ONE row, v_permlane32_swap completes them to 8 — a whole cell — and the
// lane splits and stores its cells itself, 512 contiguous bytes per half-wave and no LDS.
#include "conv_gemm_kernels.h"
#include "pgemm_rows.h"
#include "pgemm_rows_epi.h"
#include <atomic>
#include <string.h>

using namespace mitcg;

#ifdef MIT_CONV_EXPERIMENTS   // phase stamps of workgroup 0 of the few-row GEMM (100 MHz clock): scripts/dev only
__device__ unsigned long long g_rows_stamps[16];
#define MIT_ROWS_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_rows_stamps[i] = wall_clock64(); } while (0)
extern "C" int mit_dev_rows_stamps(unsigned long long *out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rows_stamps), sizeof(g_rows_stamps)) == hipSuccess ? 0 : 1; }
#else
#define MIT_ROWS_STAMP(i) do { } while (0)
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// one DMA piece: this wave's 64 lanes copy 64 x 16 bytes (per-lane source) to 1 KiB of LDS starting at the wave-uniform `dst`
__device__ __forceinline__ void dma16(const u32x4 *src, u32x4 *dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)src, (lds_void_t *)dst_wave_uniform, 16, 0, 0);
}

// LDS reads the compiler must not see: behind an LDS-DMA it would wait vmcnt(0) before any ds_read it issues itself (every DMA may alias
// the array), which drains the ring.  "=v" destinations, completion through lds_wait<N>() + lds_tie() (cdna_hip_programming.md 5.7, form ii).
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(const unsigned int addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait() {  // at most N of this wave's LDS operations outstanding (they complete in order)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_tie(bf16x8 &v) { asm volatile("" : "+v"(v)); }  // no consumer of v is scheduled above this point
__device__ __forceinline__ unsigned int lds_addr(const void *p) {
    return (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// An epilogue must leave no load behind that hipcc still counts as pending at the loop header (it would guard the first register the K
// loop redefines with vmcnt(0) on EVERY iteration): every load below is consumed unconditionally, only the stores are predicated.
#define MIT_PG_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0) only (expcnt 7, lgkmcnt 15: no wait) */

// ---- fp32 row-major epilogue of one wave's TM x TN blocks (OUTP = 0): a per-wave LDS transpose gives a lane four consecutive columns
// of one row (dwordx4 stores and residual loads), arithmetic per element in the order of mitcg::epilogue_store_vec.  Per chunk (64
// rows x 32 columns): the residual operands of the NEXT chunk are requested before this chunk's stores are issued (so waiting for them
// never waits for a store), which keeps at most one chunk's residuals in registers.  pre / post may alias c element for element (in-place
// residual layers): no __restrict__ on them, the order loads-of-a-column-before-its-store is the program's.
// Addressing: one buffer descriptor per tensor for the wave's TM*32 x TN*32 window (SGPRs) whose size is the window's VALID rows, and a
// 32-bit byte offset per access: rows past M fall outside the descriptor (loads return 0, stores are dropped by the hardware range
// check), lanes whose columns lie past N get an offset beyond any window.  No 64-bit address per access, no exec masking.
// ONE instance of the loads / transposes / stores for every (activation, pre, post) combination — wave-uniform branches around the
// parts that differ: twelve specialised copies in one kernel made hipcc hoist their common address arithmetic above the switch and
// spill.  The scheduling fences keep the next half's residual loads where they are written (hoisted above the transposes they would
// double the live residual registers).
template <int ACT>
__device__ __forceinline__ f32x4 pg_act4(f32x4 v, const float alpha) {
    v.x = apply_act<ACT>(v.x, alpha);
    v.y = apply_act<ACT>(v.y, alpha);
    v.z = apply_act<ACT>(v.z, alpha);
    v.w = apply_act<ACT>(v.w, alpha);
    return v;
}
__device__ __forceinline__ int uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *uni_ptr(T *ptr) {  // a pointer the caller knows to be wave-uniform: into SGPRs
    const uint64_t v = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}
template <int TM, int TN>
__device__ __forceinline__ void pg_store_rows(const MitPGemm &p, f32x16 (&acc)[TM][TN], float *tbuf, const int z, const int m0w, const int n0w,
                                              const int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int vr = lane >> 3, vc = (lane & 7) * 4;
    const bool has_pre = p.pre != nullptr, has_post = p.post != nullptr;
    const int rows_ok = uni(p.M - m0w < 0 ? 0 : (p.M - m0w > TM * 32 ? TM * 32 : p.M - m0w));  // z, m0w, n0w are wave-uniform
    auto window = [&](const float *base, const int64_t zs, const int64_t ld) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(uni_ptr(const_cast<float *>(base + (int64_t)z * zs + (int64_t)m0w * ld + n0w)), 0,
                                                 uni((int)(rows_ok * ld * 4)), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rc = window(p.c, p.c_zs, p.ldc);
    const __amdgpu_buffer_rsrc_t rpre = window(has_pre ? p.pre : p.c, has_pre ? p.pre_zs : p.c_zs, has_pre ? p.ld_pre : p.ldc);
    const __amdgpu_buffer_rsrc_t rpost = window(has_post ? p.post : p.c, has_post ? p.post_zs : p.c_zs, has_post ? p.ld_post : p.ldc);
    const unsigned int ldc4 = (unsigned int)p.ldc * 4u, ldpre4 = (unsigned int)p.ld_pre * 4u, ldpost4 = (unsigned int)p.ld_post * 4u;
    const bool post_first = (p.act & MIT_ACT_POST_FIRST) != 0;
    const int act = p.act & 0xff;
    // a chunk = one or two 32 x 32 blocks of one 32-column strip: the unit whose residuals are in registers at one time
    constexpr int CB = (TM < 2 || TM * TN > 4) ? 1 : 2, MCH = TM / CB, NCH = TN * MCH;  // (one block per chunk where the accumulators alone take half the registers)
    static_assert(TM % CB == 0, "chunks");
    f32x4 sc, bi, prv[CB][4], pov[CB][4];
    unsigned int colb;  // byte offset of this lane's four columns inside a window row; past N: beyond any window
    auto load_chunk = [&](const int ch) __attribute__((always_inline)) {
        const int ni = ch / MCH, m_first = (ch % MCH) * CB;
        const int n = n0w + ni * 32 + vc;
        const int nl = n < p.N ? n : 0;
        colb = n < p.N ? (unsigned int)(ni * 32 + vc) * 4u : 0xF0000000u;
        sc = f32x4{1.f, 1.f, 1.f, 1.f};
        bi = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + nl);
        if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + nl);
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int row = (unsigned int)((m_first + b) * 32 + vr + 8 * j);
                prv[b][j] = pov[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_pre) prv[b][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpre, (int)(colb + row * ldpre4), 0, 0));
                if (has_post) pov[b][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpost, (int)(colb + row * ldpost4), 0, 0));
            }
    };
    load_chunk(0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int ni = ch / MCH, m_first = (ch % MCH) * CB;
        f32x4 rv[CB][4];
#pragma unroll
        for (int b = 0; b < CB; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tbuf[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_PITCH + li] = acc[m_first + b][ni][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-synchronous exchange: LDS serves a wave's accesses in order
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[b][j] = *reinterpret_cast<const f32x4 *>(tbuf + (vr + 8 * j) * EPI_PITCH + vc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the next block overwrites the buffer
        }
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = rv[b][j];
                if (has_pre) v += prv[b][j];
                v = v * sc + bi;
                if (has_post && post_first) v += pov[b][j];
                rv[b][j] = v;
            }
        if (act == MIT_ACT_RELU) {
#pragma unroll
            for (int b = 0; b < CB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[b][j] = pg_act4<MIT_ACT_RELU>(rv[b][j], p.act_alpha);
        } else if (act == MIT_ACT_GELU) {
#pragma unroll
            for (int b = 0; b < CB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[b][j] = pg_act4<MIT_ACT_GELU>(rv[b][j], p.act_alpha);
        }
        if (has_post && !post_first) {
#pragma unroll
            for (int b = 0; b < CB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[b][j] += pov[b][j];
        }
        const unsigned int colb_st = colb;
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < NCH) load_chunk(ch + 1);  // the next chunk's residuals are requested ahead of this chunk's stores
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rv[b][j]), rc,
                                                       (int)(colb_st + (unsigned int)((m_first + b) * 32 + vr + 8 * j) * ldc4), 0, 0);
    }
}

// ---- planar epilogue (OUTP = 1 | 2): acc[mi][ni] holds D[row = n local][col = m local] (operands swapped in the MFMA), lane (li, lh):
// m = li, register r <-> n = (r & 3) + 4 lh + 8 (r >> 2).  v_permlane32_swap exchanges the upper half-wave of its first operand with the
// lower half-wave of its second: after swapping register q of cell 2 pp with register q of cell 2 pp + 1 (q < 4) a lane of half lh holds
// all eight columns of cell 2 pp + lh.  (OUTP = 2 does the same exchange with __shfl_xor(.., 32): the reference form for scripts/pgemm_check.)
template <int TM, int TN, int ACT, int OUTP>
__device__ __forceinline__ void pg_store_planes(const MitPGemm &p, f32x16 (&acc)[TM][TN], const int z, const int m0w, const int n0w,
                                                const int lane) {
    const int li = lane & 31, lh = lane >> 5;
    u32x4 *__restrict__ out = reinterpret_cast<u32x4 *>(p.c_planes + (int64_t)z * p.cp_zs);
    const int64_t plane = (int64_t)(p.N >> 3) * p.ld_cp;  // cells
    f32x4 sc[TN][2][2], bi[TN][2][2];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int n = n0w + ni * 32 + 8 * (2 * pp + lh);
            const int nl = n < p.N ? n : 0;  // N % 8 == 0
            sc[ni][pp][0] = sc[ni][pp][1] = f32x4{1.f, 1.f, 1.f, 1.f};
            bi[ni][pp][0] = bi[ni][pp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc[ni][pp][0] = *reinterpret_cast<const f32x4 *>(p.scale + nl), sc[ni][pp][1] = *reinterpret_cast<const f32x4 *>(p.scale + nl + 4);
            if (p.bias) bi[ni][pp][0] = *reinterpret_cast<const f32x4 *>(p.bias + nl), bi[ni][pp][1] = *reinterpret_cast<const f32x4 *>(p.bias + nl + 4);
        }
    MIT_PG_WAIT_LOADS();
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int n = n0w + ni * 32 + 8 * (2 * pp + lh);
            const bool n_ok = n < p.N;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                f32x4 v0, v1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned int x = __float_as_uint(acc[mi][ni][8 * pp + q]), y = __float_as_uint(acc[mi][ni][8 * pp + 4 + q]);
                    if (OUTP == 1) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);  // x' = [x.lo, y.lo], y' = [x.hi, y.hi]
                        v0[q] = __uint_as_float(sw[0]);
                        v1[q] = __uint_as_float(sw[1]);
                    } else {  // the same exchange through a cross-half shuffle: each lane sends what the other half's cell lacks
                        const unsigned int got = (unsigned int)__shfl_xor((int)(lh ? x : y), 32);
                        v0[q] = __uint_as_float(lh ? got : x);
                        v1[q] = __uint_as_float(lh ? y : got);
                    }
                }
                const int m = m0w + mi * 32 + li;
                v0 = v0 * sc[ni][pp][0] + bi[ni][pp][0];
                v1 = v1 * sc[ni][pp][1] + bi[ni][pp][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = apply_act<ACT>(v0[e], p.act_alpha);
                    v1[e] = apply_act<ACT>(v1[e], p.act_alpha);
                }
                u32x4 h, md, l;
                split8(v0, v1, h, md, l);
                if (m < p.M && n_ok) {
                    u32x4 *o = out + (int64_t)(n >> 3) * p.ld_cp + m;
                    o[0] = h;
                    o[plane] = md;
                    o[2 * plane] = l;
                }
            }
        }
    }
}

template <int TM, int TN, int OUTP>
__device__ __forceinline__ void pg_epilogue(const MitPGemm &p, f32x16 (&acc)[TM][TN], float *tbuf, const int z, const int m0w, const int n0w,
                                            const int lane) {
    if constexpr (OUTP == 0) {
        pg_store_rows<TM, TN>(p, acc, tbuf, z, m0w, n0w, lane);
    } else {
        switch (p.act & 0xff) {  // the activations the plain GEMMs of the path use; anything else is refused by the launcher
            case MIT_ACT_RELU: pg_store_planes<TM, TN, MIT_ACT_RELU, OUTP>(p, acc, z, m0w, n0w, lane); break;
            case MIT_ACT_GELU: pg_store_planes<TM, TN, MIT_ACT_GELU, OUTP>(p, acc, z, m0w, n0w, lane); break;
            default: pg_store_planes<TM, TN, MIT_ACT_NONE, OUTP>(p, acc, z, m0w, n0w, lane); break;
        }
    }
}

// VAR: variants for scripts/pgemm_check.  Timing ablations (WRONG results): 1 = no DMA pieces, 2 = no MFMAs, 4 = no barriers.  Schedules
// (same results): 16 = all DMA pieces of an iteration right behind the first fragment reads, in the shadow of their latency, instead of
// spread behind the MFMA groups; 32 = s_setprio 1 around the MFMAs
template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, int NPROD, int OUTP, int MINW, int VAR = 0>
__global__ __launch_bounds__(64 * WAVES_M *WAVES_N, MINW) void pgemm_kernel(const MitPGemm p, const int MT, const int NT, const int KT,
                                                                           const int tiles_total, const int order) {
    constexpr int NW = WAVES_M * WAVES_N, NTHR = 64 * NW;
    constexpr int KH = 2;  // 16-byte cells along k per K-tile (BK = 16: one MFMA k step)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile");
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    static_assert(NPROD == 6 || NPROD == 9, "plane pairs");
    constexpr int A_CELLS = 3 * KH * BM, B_CELLS = 3 * KH * BN;
    constexpr int GA = (A_CELLS + NTHR - 1) / NTHR, GB = (B_CELLS + NTHR - 1) / NTHR, G = GA + GB;  // DMA pieces per wave and K-tile
    constexpr int A_AREA = GA * NTHR, STAGE = (GA + GB) * NTHR;  // cells (areas rounded up to whole pieces: the surplus lanes copy a duplicate)
    static_assert(A_AREA - A_CELLS < A_CELLS && GB * NTHR - B_CELLS < B_CELLS, "pad lanes must find a duplicate cell");
    static_assert((NS - 1) * G <= 63, "vmcnt field");
    static_assert(OUTP != 0 || NW * 32 * EPI_PITCH * 4 <= STAGE * 16, "the transpose buffers of the epilogue live in one stage");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4 *const ring = reinterpret_cast<u32x4 *>(smem);  // [NS][STAGE] cells

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    // ---- the run of output tiles of this workgroup: XCD x = block % 8 owns tiles [T x / 8, T (x + 1) / 8), its workgroups equal shares
    int tile_lo, tile_n, tile_step;  // the workgroup's i-th output tile is tile_lo + i * tile_step
    {
        const unsigned int x = blockIdx.x & 7, q = blockIdx.x >> 3, nq = gridDim.x >> 3;  // gridDim.x % 8 == 0 (launcher)
        const unsigned int T = (unsigned int)tiles_total;
        const unsigned int lo = (T >> 3) * x + (((T & 7) * x) >> 3), hi = (T >> 3) * (x + 1) + (((T & 7) * (x + 1)) >> 3);  // floor(T x / 8)
        const unsigned int len = hi - lo;
        // (32-bit divisions of uniform values are expanded on the VALU: tell the compiler the results are wave-uniform)
        if (order == 0) {  // a contiguous run per workgroup: the first `rem` workgroups of the XCD take one tile more
            const unsigned int each = len / nq, rem = len - each * nq;
            tile_lo = __builtin_amdgcn_readfirstlane((int)(lo + each * q + (q < rem ? q : rem)));
            tile_n = __builtin_amdgcn_readfirstlane((int)(each + (q < rem ? 1u : 0u)));
            tile_step = 1;
        } else {  // interleaved: at any moment the XCD's workgroups are on neighbouring tiles (they share A panels and W panels in its L2)
            tile_lo = (int)(lo + q);
            tile_n = __builtin_amdgcn_readfirstlane(q < len ? (int)((len - q + nq - 1) / nq) : 0);
            tile_step = (int)nq;
        }
    }
    if (tile_n <= 0) return;  // whole workgroup: no barrier has been reached
    const int total = tile_n * KT;
    const int tiles_per_z = MT * NT;
    const int K8 = p.K >> 3;

    // ---- load cursor (K-tiles are issued in iteration order, NS - 1 ahead of the MFMAs).  A piece's source is a wave-uniform base
    // (the K-tile's first slab of the z slice: SGPRs) plus a 32-bit per-lane byte offset (the launcher checks that 3 planes fit 4 GB).
    unsigned int a_off[GA], b_off[GB];
    const char *a_kbase = nullptr, *b_kbase = nullptr;
    int ld_tile = 0, ld_kt = 0;
    auto load_tile_setup = [&]() __attribute__((always_inline)) {  // sources of K-tile 0 of output tile tile_lo + ld_tile
        const int t = tile_lo + ld_tile * tile_step;
        const int z = uni(t / tiles_per_z), tt = t - z * tiles_per_z;
        const int mt = uni(tt / NT), nt = tt - mt * NT;
        a_kbase = reinterpret_cast<const char *>(p.a_planes + (int64_t)z * p.a_zs);
        b_kbase = reinterpret_cast<const char *>(p.w_planes + (int64_t)z * p.w_zs);
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            int c = i * NTHR + tid;
            c = c < A_CELLS ? c : c - A_CELLS;  // pad lanes of the last piece copy a duplicate
            const int slab = c / BM, r = c - slab * BM, m = mt * BM + r;
            // rows past M: a valid duplicate (their results are never stored)
            a_off[i] = ((unsigned int)((slab / KH) * K8 + (slab % KH)) * (unsigned int)p.lda + (unsigned int)(m < p.M ? m : p.M - 1)) * 16u;
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            int c = i * NTHR + tid;
            c = c < B_CELLS ? c : c - B_CELLS;
            const int slab = c / BN, r = c - slab * BN, n = nt * BN + r;
            b_off[i] = ((unsigned int)((slab / KH) * K8 + (slab % KH)) * (unsigned int)p.ldw + (unsigned int)(n < (int)p.ldw ? n : (int)p.ldw - 1)) * 16u;
        }
    };
    const int64_t a_step = (int64_t)KH * p.lda * 16, b_step = (int64_t)KH * p.ldw * 16;  // bytes per K-tile
    // piece i of the K-tile being issued (A pieces first), then the cursor moves on
    auto issue_piece = [&](const int slot, const int i) __attribute__((always_inline)) {
        u32x4 *st = ring + slot * STAGE + wave * 64;
        if (i < GA)
            dma16(reinterpret_cast<const u32x4 *>(a_kbase + a_off[i < GA ? i : 0]), st + i * NTHR);
        else
            dma16(reinterpret_cast<const u32x4 *>(b_kbase + b_off[i >= GA ? i - GA : 0]), st + A_AREA + (i - GA) * NTHR);
    };
    auto issue_done = [&]() __attribute__((always_inline)) {
        a_kbase += a_step;
        b_kbase += b_step;
        if (++ld_kt == KT) {
            ld_kt = 0;
            if (++ld_tile < tile_n) load_tile_setup();
        }
    };
    auto issue = [&](const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G; ++i) issue_piece(slot, i);
        issue_done();
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    };
    const unsigned int a_frag0 = lds_addr(ring) + (unsigned int)(lh * BM + wm0 + li) * 16u;
    const unsigned int b_frag0 = lds_addr(ring) + (unsigned int)(A_AREA + lh * BN + wn0 + li) * 16u;
    // One K-tile: the fragment reads in the order the plane pairs consume them, the MFMAs of a pair as soon as its fragments are in,
    // and (do_issue) the DMA pieces of K-tile it+NS-1 spread behind the first three MFMA groups (a piece costs its wave tens of
    // cycles to issue: among MFMAs that is covered, ahead of the fragment reads it is not).
    auto compute = [&](const int slot, const int dma_slot, const bool do_issue) __attribute__((always_inline)) {
        const unsigned int aa = a_frag0 + (unsigned int)slot * (STAGE * 16u), ba = b_frag0 + (unsigned int)slot * (STAGE * 16u);
        bf16x8 af[3][TM], bf[3][TN];
        constexpr int PA[3] = {0, 2, 1}, PB[3] = {2, 0, 1};  // read group o: A plane PA[o], W plane PB[o]
        __builtin_amdgcn_sched_barrier(0);
#define MIT_PG_READ_GROUP(o)                                                                                     \
    {                                                                                                            \
        if constexpr (TM >= 1) af[PA[o]][0] = lds_read16<(PA[o] * KH * BM) * 16>(aa);                            \
        if constexpr (TM >= 2) af[PA[o]][TM >= 2 ? 1 : 0] = lds_read16<(PA[o] * KH * BM + 32) * 16>(aa);         \
        if constexpr (TM >= 3) af[PA[o]][TM >= 3 ? 2 : 0] = lds_read16<(PA[o] * KH * BM + 64) * 16>(aa);         \
        if constexpr (TM >= 4) af[PA[o]][TM >= 4 ? 3 : 0] = lds_read16<(PA[o] * KH * BM + 96) * 16>(aa);         \
        if constexpr (TN >= 1) bf[PB[o]][0] = lds_read16<(PB[o] * KH * BN) * 16>(ba);                            \
        if constexpr (TN >= 2) bf[PB[o]][TN >= 2 ? 1 : 0] = lds_read16<(PB[o] * KH * BN + 32) * 16>(ba);         \
        if constexpr (TN >= 3) bf[PB[o]][TN >= 3 ? 2 : 0] = lds_read16<(PB[o] * KH * BN + 64) * 16>(ba);         \
        if constexpr (TN >= 4) bf[PB[o]][TN >= 4 ? 3 : 0] = lds_read16<(PB[o] * KH * BN + 96) * 16>(ba);         \
    }
        static_assert(TM <= 4 && TN <= 4, "fragment read macro");
        auto tie_group = [&](const int o) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) lds_tie(af[PA[o]][mi]);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) lds_tie(bf[PB[o]][ni]);
        };
        auto mfma_pair = [&](const int pr) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    if constexpr ((VAR & 2) != 0) {  // timing ablation: no MFMAs (the fragments stay live)
                        asm volatile("" ::"v"(af[kSplitPA[pr]][mi]), "v"(bf[kSplitPB[pr]][ni]));
                    } else if constexpr (OUTP == 0)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kSplitPA[pr]][mi], bf[kSplitPB[pr]][ni], acc[mi][ni], 0, 0, 0);
                    else  // transposed result: rows = output columns
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[kSplitPB[pr]][ni], af[kSplitPA[pr]][mi], acc[mi][ni], 0, 0, 0);
                }
        };
        constexpr int PER = (G + 2) / 3;  // DMA pieces behind each of the first three MFMA groups
        int dma_calls = 0;  // (compile-time after unrolling: with VAR & 16 the first three calls issue, the later ones are empty)
        auto dma_group = [&](const int g) __attribute__((always_inline)) {
            if ((VAR & 16) != 0 && dma_calls++ >= 3) return;  // VAR & 16: the first three calls (ahead of the MFMAs) issue, the ones behind the MFMA groups are empty
            __builtin_amdgcn_sched_barrier(0);
            if (do_issue) {  // wave-uniform: a scalar branch around the pieces; ONE instance of the MFMA chain keeps the accumulators in place
                if constexpr ((VAR & 1) == 0) {
#pragma unroll
                    for (int i = g * PER; i < (g + 1) * PER && i < G; ++i) issue_piece(dma_slot, i);
                }
                if (g == 2) issue_done();
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (NPROD == 6) {  // pairs 3 .. 8 = (A0 W2) (A2 W0) (A1 W1) (A0 W1) (A1 W0) (A0 W0): group o completes pair 3 + o
            // W plane 2 and A plane 2 serve one pair each: the third group's reads are issued behind the second pair's MFMAs (long before
            // they are needed) so that their registers can be the ones planes 2 leave — twelve fragments live instead of eighteen
            MIT_PG_READ_GROUP(0)
            MIT_PG_READ_GROUP(1)
            if constexpr ((VAR & 16) != 0) {
                dma_group(0);
                dma_group(1);
                dma_group(2);
            }
            lds_wait<TM + TN>();
            tie_group(0);
            if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(1);
            mfma_pair(3);
            dma_group(0);
            lds_wait<0>();
            tie_group(1);
            mfma_pair(4);
            __builtin_amdgcn_sched_barrier(0);
            MIT_PG_READ_GROUP(2)
            dma_group(1);
            lds_wait<0>();
            tie_group(2);
            mfma_pair(5);
            dma_group(2);
            mfma_pair(6);
            mfma_pair(7);
            mfma_pair(8);
            if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(0);
        } else {
            MIT_PG_READ_GROUP(0)
            MIT_PG_READ_GROUP(1)
            MIT_PG_READ_GROUP(2)
            lds_wait<0>();
            tie_group(0);
            tie_group(1);
            tie_group(2);
#pragma unroll
            for (int pr = 0; pr < 9; ++pr) {
                mfma_pair(pr);
                if (pr < 3) dma_group(pr);
            }
        }
#undef MIT_PG_READ_GROUP
        __builtin_amdgcn_sched_barrier(0);
    };
    load_tile_setup();
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (issued < total) {
            issue(s);
            ++issued;
        }
    zero_acc();

    if constexpr ((VAR & 64) != 0) {
        // ---- ping-pong form (round 6 experiment): eight waves = two per SIMD (waves w and w + 4 share one); the second
        // half runs ONE barrier behind the first, so that between any two barriers one wave of a SIMD is in its compute segment (the 24
        // MFMAs of a K-tile, nothing else, s_setprio 1) and its partner in its load segment (fragment reads of its next K-tile, the DMA
        // pieces of K-tile + NS - 1, the counted wait).  Two barriers per K-tile; per accumulator the same pairs in the same order as
        // every other tile.  Hazards (g = global barrier count, first half: L(t) in [2t, 2t+1], C(t) in [2t+1, 2t+2]; second half one
        // later): a stage is re-filled in L(t) with K-tile t + NS - 1 after both halves' reads of K-tile t - 1 have completed
        // (lgkmcnt(0) before the barrier that ends every L); every wave has waited for its own pieces of K-tile t + 1 before barrier
        // 2t + 2, the first barrier ahead of anybody's reads of it.
        static_assert(NW == 8 && NPROD == 6 && TM <= 2 && TN <= 2, "ping-pong form: 8 waves, 6 pairs, 64 x 64 wave tile");
        const bool second = wave >= NW / 2;
        bf16x8 af[3][TM], bf[3][TN];
        int it = 0, slot = 0, fslot = NS - 1;
        for (int ti = 0;; ++ti) {
            wait_vmcnt<0>();
            wg_barrier();
            if (ti > 0) {
                const int t = tile_lo + (ti - 1) * tile_step;
                const int z = uni(t / tiles_per_z), tt = t - z * tiles_per_z;
                const int mt = uni(tt / NT), nt = tt - mt * NT;
                float *tbuf = reinterpret_cast<float *>(ring + fslot * STAGE) + wave * (32 * EPI_PITCH);
                pg_epilogue<TM, TN, OUTP>(p, acc, tbuf, z, mt * BM + wm0, nt * BN + wn0, lane);
                if (ti == tile_n) break;
                if (OUTP == 0) wg_barrier();
            }
            zero_acc();
            if (second) wg_barrier();  // the stagger
            for (int kt = 0; kt < KT; ++kt, ++it) {
                // -- load segment
                const unsigned int aa = a_frag0 + (unsigned int)slot * (STAGE * 16u), ba = b_frag0 + (unsigned int)slot * (STAGE * 16u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    af[pl][0] = pl == 0 ? lds_read16<0>(aa) : pl == 1 ? lds_read16<(1 * KH * BM) * 16>(aa) : lds_read16<(2 * KH * BM) * 16>(aa);
                    if constexpr (TM >= 2)
                        af[pl][TM >= 2 ? 1 : 0] = pl == 0 ? lds_read16<32 * 16>(aa) : pl == 1 ? lds_read16<(1 * KH * BM + 32) * 16>(aa) : lds_read16<(2 * KH * BM + 32) * 16>(aa);
                    bf[pl][0] = pl == 0 ? lds_read16<0>(ba) : pl == 1 ? lds_read16<(1 * KH * BN) * 16>(ba) : lds_read16<(2 * KH * BN) * 16>(ba);
                    if constexpr (TN >= 2)
                        bf[pl][TN >= 2 ? 1 : 0] = pl == 0 ? lds_read16<32 * 16>(ba) : pl == 1 ? lds_read16<(1 * KH * BN + 32) * 16>(ba) : lds_read16<(2 * KH * BN + 32) * 16>(ba);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool do_issue = issued < total;
                if (do_issue) {
#pragma unroll
                    for (int i = 0; i < G; ++i) issue_piece(fslot, i);
                    issue_done();
                }
                issued += do_issue ? 1 : 0;
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int ahead = issued - it - 2;  // K-tiles issued behind K-tile it + 1
                    if (ahead <= 0) wait_vmcnt<0>();
                    else if (ahead == 1 || NS == 3) wait_vmcnt<G>();
                    else wait_vmcnt<2 * G>();
                }
                lds_wait<0>();
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) lds_tie(af[pl][mi]);
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) lds_tie(bf[pl][ni]);
                }
                wg_barrier();
                // -- compute segment
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int pr = 3; pr < 9; ++pr)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kSplitPA[pr]][mi], bf[kSplitPB[pr]][ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                wg_barrier();
                fslot = slot;
                slot = slot + 1 == NS ? 0 : slot + 1;
            }
            if (!second) wg_barrier();  // the halves meet again for the epilogue
        }
        return;
    }

    // ---- the walk over this workgroup's output tiles and their K-tiles (`it` counts K-tiles over all of them; the DMA runs NS-1 of
    // them ahead, across tile boundaries).  The DMA pieces of K-tile `it` must have landed before its fragments are read.  In issue
    // order behind them: the K-tiles it+1 .. issued-1 and, after an epilogue, its stores.
    //   steady iterations (all but a few per output tile): exactly NS-2 K-tiles were issued after `it` -> vmcnt((NS-2) G)
    //   the first iteration of an output tile stores the previous one first: it drains everything (all of it was issued at least one
    //   K-tile ago), the NS-2 iterations after it need what that drain covered, and from then on every store precedes the K-tile
    //   waited for, so "at most (issued - it - 1) G operations outstanding" again implies it has landed, in whatever order stores retire
    //   the last NS-2 iterations of the run wait with the count of what is still behind them.
    // The accumulators are written by ONE chain of MFMAs in the inner loop and zeroed in the outer one: any other shape of this loop made
    // hipcc copy all of them on every iteration.
    int it = 0, slot = 0, fslot = NS - 1;  // slot: stage of K-tile it; fslot: stage of K-tile it-1 (free once the barrier is passed)
    for (int ti = 0;; ++ti) {              // one pass more than there are tiles: the last one only stores the last tile
        if (ti > 0) {
            wait_vmcnt<0>();
            if constexpr ((VAR & 4) == 0) wg_barrier();
            const int t = tile_lo + (ti - 1) * tile_step;
            const int z = uni(t / tiles_per_z), tt = t - z * tiles_per_z;
            const int mt = uni(tt / NT), nt = tt - mt * NT;
            float *tbuf = reinterpret_cast<float *>(ring + fslot * STAGE) + wave * (32 * EPI_PITCH);
            pg_epilogue<TM, TN, OUTP>(p, acc, tbuf, z, mt * BM + wm0, nt * BN + wn0, lane);
            if (ti == tile_n) break;
            if (OUTP == 0) wg_barrier();  // the transpose buffers are about to be overwritten by the DMA
        }
        zero_acc();
        const int steady_lo = ti > 0 ? NS - 1 : 0;                                // the first kt that waits with the steady count
        const int steady_hi = total - (NS - 1) - it + 1 < KT ? total - (NS - 1) - it + 1 : KT;  // iterations it' <= total - (NS-1) have NS-2 K-tiles behind them
        for (int kt = 0; kt < KT; ++kt, ++it) {
            if (kt >= steady_lo && kt < steady_hi) {
                wait_vmcnt<(NS - 2) * G>();
                if constexpr ((VAR & 4) == 0) wg_barrier();
            } else if (kt >= steady_lo) {  // the tail of the run
                const int ahead = issued - it - 1;
                if (ahead <= 0) wait_vmcnt<0>();
                else if (ahead == 1) wait_vmcnt<G>();
                else wait_vmcnt<2 * G>();
                if constexpr ((VAR & 4) == 0) wg_barrier();
            } else if (kt > 0) {  // covered by the drain ahead of the epilogue; kt == 0 has passed its barrier there too
                if constexpr ((VAR & 4) == 0) wg_barrier();
            }
            const bool do_issue = issued < total;
            compute(slot, fslot, do_issue);
            issued += do_issue ? 1 : 0;
            fslot = slot;
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
    }
}

// ---- stand-alone producer / joiner ------------------------------------------------------------------------------------------------
// 64 rows x 64 k per workgroup: coalesced float4 reads along k, through LDS, one (row, k-cell) per thread on the way out so that
// consecutive lanes write consecutive rows of a slab (16 bytes each).
__global__ __launch_bounds__(256) void split_planes_kernel(const float *__restrict__ x, const int64_t ldx, const int R, const int K,
                                                         u32x4 *__restrict__ out, const int64_t ld) {
    __shared__ float t[64][68];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int kq = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rr + 16 * i, k = k0 + kq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R && k < K) v = *reinterpret_cast<const f32x4 *>(x + (int64_t)r * ldx + k);
        *reinterpret_cast<f32x4 *>(&t[rr + 16 * i][kq * 4]) = v;
    }
    __syncthreads();
    const int64_t plane = (int64_t)(K >> 3) * ld;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (threadIdx.x >> 6) + 4 * i, r = threadIdx.x & 63;  // k-cell of the tile, row
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(&t[r][c * 8]), hi = *reinterpret_cast<const f32x4 *>(&t[r][c * 8 + 4]);
        u32x4 h, m, l;
        split8(lo, hi, h, m, l);
        const int k8 = (k0 >> 3) + c;
        if (r0 + r < R && k8 * 8 < K) {
            u32x4 *o = out + (int64_t)k8 * ld + r0 + r;
            o[0] = h;
            o[plane] = m;
            o[2 * plane] = l;
        }
    }
}

__global__ __launch_bounds__(256) void join_planes_kernel(const u32x4 *__restrict__ in, const int64_t ld, const int R, const int K8,
                                                        float *__restrict__ x, const int64_t ldx) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)R * K8) return;
    const int r = (int)(idx % R), k8 = (int)(idx / R);
    const int64_t plane = (int64_t)K8 * ld;
    const u32x4 h = in[(int64_t)k8 * ld + r], m = in[plane + (int64_t)k8 * ld + r], l = in[2 * plane + (int64_t)k8 * ld + r];
    float *o = x + (int64_t)r * ldx + k8 * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[2 * j] = bf16_lo(h[j]) + (bf16_lo(m[j]) + bf16_lo(l[j]));  // mid + lo is the (exact) first residual
        o[2 * j + 1] = bf16_hi(h[j]) + (bf16_hi(m[j]) + bf16_hi(l[j]));
    }
}

// ---- few-row GEMMs (the decoder's Linears at one page: M = lines x beams = 160 rows) ---------------------------------------------------
// A 64 x 64 split tile takes 12-17 us for such a launch whatever its size: every K-tile is a dependent chain global load -> split ->
// ds_write -> barrier -> ds_read -> MFMA.  With planar operands nothing of that chain is needed: a cell IS a lane's MFMA operand, so one
// wave per 32 x 32 block of the output streams its 3 + 3 cells per 16-deep k step straight from global memory (L2) into registers, D k
// steps ahead (buffer loads: a wave-uniform offset per step, no address arithmetic in the loop), and runs the NPROD MFMAs of the step
// back to back — no LDS, no barrier, no VALU work in the loop; the launch is as long as its one accumulator chain (K / 16 x NPROD MFMAs
// of 32 cycles) plus one load latency.  The MFMA operands are swapped as for the planar output above, so after v_permlane32_swap a lane
// holds whole cells (eight consecutive columns of one row): 16-byte fp32 stores and / or a split into planes.  Per element the pair
// order, the k order and the epilogue arithmetic of the other tiles: bit-identical results.
// Decoder extras (PgRowsExt; zero-initialised = none): the three-way column split of the q | k | v projection (MitTensorMap::nsplit),
// the device-resident step counter (the output row of the step), both output kinds at once.

// KTS: K / 16 as a compile-time constant (the decoder's K = 320 and 2048): the k loop fully unrolled, so that hipcc counts the loads in
// flight exactly — at the header of a run-time loop it waits for vmcnt(0), which empties the prefetch ring once per D steps (KTS = 0).
// Leading scalar arguments = what the first operand requests need (pointers, slab strides, tile counts): with kernel-argument preloading
// (build.py: -amdgpu-kernarg-preload-count for this unit) they arrive in SGPRs with the wave instead of behind a scalar load from the
// kernarg segment in device memory — the round trip every one of these 6 us launches began with.
template <int NPROD, int D, int KTS>
__global__ __launch_bounds__(64) void pgemm_rows_kernel(const uint16_t *pa, const uint16_t *pw, const unsigned int lda_u, const unsigned int ldw_u, const int Kq,
                                                        const int MT, const int NT, const int KT, const MitPGemm p, const PgRowsExt x) {
    MIT_ROWS_STAMP(0);
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    // block -> (column block, row block): the row blocks of a column block (same W cells) on one XCD, blocks dealt round-robin to the XCDs
    const int total = MT * NT, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int nt = t / MT, mt = t - nt * MT;
    const int m0 = mt * 32, n0 = nt * 32;
    const int K8 = Kq >> 3;
    const unsigned int a_step = lda_u * 32u, w_step = ldw_u * 32u;     // bytes per k step (two k cells)
    const unsigned int a_plane = (unsigned int)K8 * lda_u * 16u, w_plane = (unsigned int)K8 * ldw_u * 16u;
    const int z = blockIdx.y;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(z ? pa + (int64_t)z * p.a_zs : pa), 0, 3 * a_plane, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(z ? pw + (int64_t)z * p.w_zs : pw), 0, 3 * w_plane, 0x00020000);
    const unsigned int a_off = ((unsigned int)lh * lda_u + (unsigned int)(m0 + li)) * 16u;
    const unsigned int w_off = ((unsigned int)lh * ldw_u + (unsigned int)(n0 + li)) * 16u;

    u32x4 fa[D][3], fw[D][3];
    // (k steps are requested in ascending order: running offsets — with ks * step per request the fully unrolled K = 2048 form hoists
    // its 768 scalar products to the top and spills SGPRs to scratch)
    unsigned int a_run = 0, w_run = 0;
    auto issue = [&](const int d, const int /*ks: ascending, one step per call*/) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            fa[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(ra, a_off, pl * a_plane + a_run, 0);
            fw[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off, pl * w_plane + w_run, 0);
        }
        a_run += a_step;
        w_run += w_step;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto consume = [&](const int d) __attribute__((always_inline)) {
#pragma unroll
        for (int pr = 9 - NPROD; pr < 9; ++pr)   // transposed result (rows = output columns), as the planar tiles
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[d][kSplitPB[pr]]), __builtin_bit_cast(bf16x8, fa[d][kSplitPA[pr]]), acc, 0, 0, 0);
    };
    if constexpr (KTS > 0) {
#pragma unroll
        for (int d = 0; d < D && d < KTS; ++d) issue(d, d);
        __builtin_amdgcn_sched_barrier(0);   // (the fences keep the loads D steps ahead: left alone the scheduler sinks each load to its use)
        MIT_ROWS_STAMP(1);
#pragma unroll
        for (int ks = 0; ks < KTS; ++ks) {
            consume(ks % D);
            if (ks + D < KTS) issue(ks % D, ks + D);
            __builtin_amdgcn_sched_barrier(0);
#ifdef MIT_CONV_EXPERIMENTS
            if (ks == 0) { asm volatile("" : "+v"(acc)); MIT_ROWS_STAMP(2); }
#endif
        }
#ifdef MIT_CONV_EXPERIMENTS
        asm volatile("" : "+v"(acc));
        MIT_ROWS_STAMP(3);
#endif
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < KT) issue(d, d);
        int k0 = 0;
        for (; k0 + 2 * D <= KT; k0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                consume(d);
                issue(d, k0 + d + D);
            }
        }
        for (; k0 < KT; k0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (k0 + d < KT) {
                    consume(d);
                    if (k0 + d + D < KT) issue(d, k0 + d + D);
                }
            }
        }
    }

    pg_rows_epilogue(p, x, acc, m0, n0, z, lane);
    MIT_ROWS_STAMP(4);
}

// ---- the same block with K cut into four: the FFN's second Linear (K = 2048) at one page is ONE accumulator chain of 768 MFMAs per
// wave — 10 us that no prefetch shortens.  Four waves of a workgroup take a quarter of the k steps each (their own prefetch rings), three
// of them park their accumulators in LDS, wave 0 adds them in the fixed order ((p0 + p1) + p2) + p3 and runs the epilogue: 12.0 us per
// launch instead of 16.0.  Deterministic, but NOT the k-sequential sum of the other tiles: a page decoded with this kernel differs from
// the same page inside a 16-page group in the last bits of this Linear's output (fp32 rounding of three extra additions per element).
// ocr_decoder.hip uses it for few rows only and says so; MIT_OCR_FF2_SPLITK=0 keeps the one-chain kernel.
// Measured and dropped (profiles/r11j, r11k): the four slices as four one-wave workgroups on four CUs with a last-arriver reduction
// through a workspace (release fence + device-scope counter per slice) — 12.8 us; sixteen slices — 28.9 us: every device-scope
// release fence writes the XCD's L2 back, which costs more than the 786 KB of operand cells cost one CU's L2 port.
template <int NPROD, int D, int KTW>   // KTW: k steps per wave (K = 4 * 16 * KTW); leading scalars: see pgemm_rows_kernel
__global__ __launch_bounds__(256) void pgemm_rows_splitk_kernel(const uint16_t *pa, const uint16_t *pw, const unsigned int lda_u, const unsigned int ldw_u,
                                                               const int Kq, const int MT, const int NT, const MitPGemm p, const PgRowsExt x) {
    __shared__ __attribute__((aligned(16))) float part[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int total = MT * NT, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int nt = t / MT, mt = t - nt * MT;
    const int m0 = mt * 32, n0 = nt * 32;
    const int K8 = Kq >> 3;
    const unsigned int a_step = lda_u * 32u, w_step = ldw_u * 32u;
    const unsigned int a_plane = (unsigned int)K8 * lda_u * 16u, w_plane = (unsigned int)K8 * ldw_u * 16u;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(pa), 0, 3 * a_plane, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(pw), 0, 3 * w_plane, 0x00020000);
    const unsigned int ks0 = (unsigned int)wave * KTW;
    const unsigned int a_off = ((unsigned int)lh * lda_u + (unsigned int)(m0 + li)) * 16u + ks0 * a_step;
    const unsigned int w_off = ((unsigned int)lh * ldw_u + (unsigned int)(n0 + li)) * 16u + ks0 * w_step;
    u32x4 fa[D][3], fw[D][3];
    auto issue = [&](const int d, const int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            fa[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(ra, a_off, pl * a_plane + (unsigned int)ks * a_step, 0);
            fw[d][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off, pl * w_plane + (unsigned int)ks * w_step, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int d = 0; d < D && d < KTW; ++d) issue(d, d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KTW; ++ks) {
#pragma unroll
        for (int pr = 9 - NPROD; pr < 9; ++pr)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[ks % D][kSplitPB[pr]]), __builtin_bit_cast(bf16x8, fa[ks % D][kSplitPA[pr]]), acc, 0, 0, 0);
        if (ks + D < KTW) issue(ks % D, ks + D);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += part[w][r][lane];
    pg_rows_epilogue(p, x, acc, m0, n0, 0, lane);
}

template <int NPROD, int D>
void pg_rows_launch_ext(const MitPGemm &p, const PgRowsExt &x, hipStream_t s) {
    const int MT = (p.M + 31) / 32, NT = (p.N + 31) / 32, KT = p.K / 16;
    const int total = MT * NT, per = (total + 7) / 8;
    const dim3 grid(per * 8, p.Z > 0 ? p.Z : 1);
    if (KT == 20) hipLaunchKernelGGL((pgemm_rows_kernel<NPROD, D, 20>), grid, dim3(64), 0, s, p.a_planes, p.w_planes, (unsigned int)p.lda, (unsigned int)p.ldw, p.K, MT, NT, KT, p, x);
    else if (KT == 128) hipLaunchKernelGGL((pgemm_rows_kernel<NPROD, D, 128>), grid, dim3(64), 0, s, p.a_planes, p.w_planes, (unsigned int)p.lda, (unsigned int)p.ldw, p.K, MT, NT, KT, p, x);
    else hipLaunchKernelGGL((pgemm_rows_kernel<NPROD, D, 0>), grid, dim3(64), 0, s, p.a_planes, p.w_planes, (unsigned int)p.lda, (unsigned int)p.ldw, p.K, MT, NT, KT, p, x);
}

// ---- tiles and launch -------------------------------------------------------------------------------------------------------------
typedef void (*PgLaunch)(const MitPGemm &, int MT, int NT, int KT, int tiles, int grid, int order, hipStream_t);
struct PgTile {
    const char *name, *probe;  // probe: the name mit_prof_kernels_read files its launches under
    int BM, BN, nprod, outp, wgs_per_cu;
    PgLaunch launch;
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, int NPROD, int OUTP, int MINW, int VAR = 0>
void pg_launch(const MitPGemm &p, int MT, int NT, int KT, int tiles, int grid, int order, hipStream_t s) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N, KH = 2;
    constexpr int GA = (3 * KH * BM + NTHR - 1) / NTHR, GB = (3 * KH * BN + NTHR - 1) / NTHR;
    const size_t smem = (size_t)NS * (GA + GB) * NTHR * 16;
    auto kern = pgemm_kernel<BM, BN, WAVES_M, WAVES_N, NS, NPROD, OUTP, MINW, VAR>;
    static DynSmemOptIn optin;
    optin.ensure(reinterpret_cast<const void *>(kern), smem);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), smem, s, p, MT, NT, KT, tiles, order);
}

template <int NPROD, int D>
void pg_rows_launch(const MitPGemm &p, int, int, int, int, int, int, hipStream_t s) {
    pg_rows_launch_ext<NPROD, D>(p, PgRowsExt{}, s);
}
#define PG_ROWS_TILE(name, NPROD, D, OUTP) {name, "pgemm_rows_kernel<" name ">", 32, 32, NPROD, OUTP, 0, pg_rows_launch<NPROD, D>}

#define PG_TILE(name, BM, BN, WMV, WNV, NS, NPROD, OUTP, MINW, WGS) \
    {name, "pgemm_kernel<" name ">", BM, BN, NPROD, OUTP, WGS, pg_launch<BM, BN, WMV, WNV, NS, NPROD, OUTP, MINW>}
#define PG_TILE_V(name, BM, BN, WMV, WNV, NS, NPROD, OUTP, MINW, WGS, VARV) \
    {name, "pgemm_kernel<" name ">", BM, BN, NPROD, OUTP, WGS, pg_launch<BM, BN, WMV, WNV, NS, NPROD, OUTP, MINW, VARV>}
const PgTile kPgTiles[] = {
    // the shipped set (pick_tile): 128 x 128 and 128 x 64, fp32 and planar output, 6 and 9 plane pairs; 3 stages, two workgroups per CU
    PG_TILE("pg128x128s3p6", 128, 128, 2, 2, 3, 6, 0, 2, 2),       // 0
    PG_TILE("pg128x64s3p6", 128, 64, 2, 2, 3, 6, 0, 2, 2),         // 1
    PG_TILE("pg128x128s3p6P", 128, 128, 2, 2, 3, 6, 1, 2, 2),      // 2: planar output
    PG_TILE("pg128x64s3p6P", 128, 64, 2, 2, 3, 6, 1, 2, 2),        // 3
    PG_TILE("pg128x128s3p9", 128, 128, 2, 2, 3, 9, 0, 2, 2),       // 4
    PG_TILE("pg128x64s3p9", 128, 64, 2, 2, 3, 9, 0, 2, 2),         // 5
    PG_TILE("pg128x128s3p9P", 128, 128, 2, 2, 3, 9, 1, 2, 2),      // 6
    PG_TILE("pg128x64s3p9P", 128, 64, 2, 2, 3, 9, 1, 2, 2),        // 7
    // measured alternatives (scripts/pgemm_check)
    PG_TILE("pg128x128s2p6", 128, 128, 2, 2, 2, 6, 0, 2, 2),       // 8: two stages
    PG_TILE("pg128x128s4p6", 128, 128, 2, 2, 4, 6, 0, 1, 1),       // 9: four stages, one workgroup per CU
    PG_TILE("pg256x128s3p6", 256, 128, 4, 2, 3, 6, 0, 2, 1),       // 10: eight waves, wave tile 64 x 64
    PG_TILE("pg256x128s3p6P", 256, 128, 4, 2, 3, 6, 1, 2, 1),      // 11
    PG_TILE("pg128x128s3p6Q", 128, 128, 2, 2, 3, 6, 2, 2, 2),      // 12: 2 with the cell exchange through __shfl_xor instead of v_permlane32_swap
    PG_TILE("pg256x256s3p6P", 256, 256, 2, 4, 3, 6, 1, 2, 1),      // 13: eight waves, wave tile 128 x 64: 12 KB of DMA per 128 x 128 of output and K-tile (24 for tile 0); the fp32-output form does not fit 256 registers
    PG_TILE("pg256x128s4p6", 256, 128, 4, 2, 4, 6, 0, 2, 1),       // 14: 10 with four stages
    PG_TILE_V("pg128x128s3p6d", 128, 128, 2, 2, 3, 6, 0, 2, 2, 16),  // 15: tile 0 with the DMA pieces in the shadow of the fragment reads
    PG_TILE_V("pg128x128s3p6dp", 128, 128, 2, 2, 3, 6, 0, 2, 2, 48), // 16: ... and s_setprio around the MFMAs
    PG_TILE_V("pg128x128s3p6p", 128, 128, 2, 2, 3, 6, 0, 2, 2, 32),  // 17: tile 0 with s_setprio around the MFMAs
    PG_TILE_V("pg256x128s3p6pp", 256, 128, 4, 2, 3, 6, 0, 2, 1, 64),  // ping-pong halves (round 6 experiment): see the VAR & 64 loop
    PG_TILE_V("pg128x256s3p6pp", 128, 256, 2, 4, 3, 6, 0, 2, 1, 64),
    // few-row launches (the decoder's Linears at one page): one wave per 32 x 32 block, operands streamed through registers
    PG_ROWS_TILE("pgrows32d6p6", 6, 6, 0),    // 18: six k steps ahead (mit_pgemm_rows: the native decoder loop)
    PG_ROWS_TILE("pgrows32d6p6P", 6, 6, 1),   // 19
    PG_ROWS_TILE("pgrows32d6p9", 9, 6, 0),    // 20
    PG_ROWS_TILE("pgrows32d6p9P", 9, 6, 1),   // 21
    PG_ROWS_TILE("pgrows32d4p6", 6, 4, 0),    // 22: prefetch depth 4 (scripts/pgemm_check: 4 / 6 / 8 within 10 % of each other, 10 slower)
    PG_ROWS_TILE("pgrows32d8p6", 6, 8, 0),    // 23: ... 8
#ifdef MIT_CONV_EXPERIMENTS  // timing ablations of tile 0 (WRONG results; scripts/pgemm_check prints their times only)
    {"xpgNoDma", "pgemm_kernel<xpgNoDma>", 128, 128, 6, 0, 2, pg_launch<128, 128, 2, 2, 3, 6, 0, 2, 1>},
    {"xpgNoMfma", "pgemm_kernel<xpgNoMfma>", 128, 128, 6, 0, 2, pg_launch<128, 128, 2, 2, 3, 6, 0, 2, 2>},
    {"xpgNoBar", "pgemm_kernel<xpgNoBar>", 128, 128, 6, 0, 2, pg_launch<128, 128, 2, 2, 3, 6, 0, 2, 4>},
    {"xpgNoDmaNoBar", "pgemm_kernel<xpgNoDmaNoBar>", 128, 128, 6, 0, 2, pg_launch<128, 128, 2, 2, 3, 6, 0, 2, 5>},
    {"xpgNoMfmaNoBar", "pgemm_kernel<xpgNoMfmaNoBar>", 128, 128, 6, 0, 2, pg_launch<128, 128, 2, 2, 3, 6, 0, 2, 6>},
    {"xpg256NoDmaP", "pgemm_kernel<xpg256NoDmaP>", 256, 256, 6, 1, 1, pg_launch<256, 256, 2, 4, 3, 6, 1, 2, 1>},
    {"xpg256NoMfmaP", "pgemm_kernel<xpg256NoMfmaP>", 256, 256, 6, 1, 1, pg_launch<256, 256, 2, 4, 3, 6, 1, 2, 2>},
#endif
};
constexpr int kNumPgTiles = sizeof(kPgTiles) / sizeof(kPgTiles[0]);

int pg_check(const MitPGemm &p) {
    if (!p.a_planes || !p.w_planes) return mit_set_error("mit_pgemm: null operand");
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.Z <= 0) return mit_set_error("mit_pgemm: empty problem");
    if (p.K % 16) return mit_set_error("mit_pgemm: K %% 16 != 0 (K=%d)", p.K);
    if (p.lda < p.M || p.ldw < p.N) return mit_set_error("mit_pgemm: lda / ldw smaller than M / N");
    if ((int64_t)3 * (p.K >> 3) * p.lda * 16 > 0xffffffffLL || (int64_t)3 * (p.K >> 3) * p.ldw * 16 > 0xffffffffLL)
        return mit_set_error("mit_pgemm: an operand's three planes exceed 4 GB (32-bit piece offsets)");
    if ((reinterpret_cast<uintptr_t>(p.a_planes) & 15) || (reinterpret_cast<uintptr_t>(p.w_planes) & 15) || (p.a_zs & 7) || (p.w_zs & 7))
        return mit_set_error("mit_pgemm: operands must be 16-byte aligned");
    if ((!p.c) == (!p.c_planes)) return mit_set_error("mit_pgemm: exactly one of c / c_planes");
    if (p.c) {
        if ((p.N & 3) || (p.ldc & 3) || (p.c_zs & 3) || (reinterpret_cast<uintptr_t>(p.c) & 15)) return mit_set_error("mit_pgemm: c needs N, ldc %% 4 == 0 and a 16-byte aligned base");
        if (p.pre && ((p.ld_pre & 3) || (p.pre_zs & 3) || (reinterpret_cast<uintptr_t>(p.pre) & 15))) return mit_set_error("mit_pgemm: pre must be float4-addressable");
        if (p.post && ((p.ld_post & 3) || (p.post_zs & 3) || (reinterpret_cast<uintptr_t>(p.post) & 15))) return mit_set_error("mit_pgemm: post must be float4-addressable");
    } else {
        if ((p.N & 7) || p.ld_cp < p.M || (p.cp_zs & 7) || (reinterpret_cast<uintptr_t>(p.c_planes) & 15)) return mit_set_error("mit_pgemm: c_planes needs N %% 8 == 0, ld_cp >= M and a 16-byte aligned base");
        if (p.pre || p.post) return mit_set_error("mit_pgemm: pre / post are not available with planar output");
    }
    if ((reinterpret_cast<uintptr_t>(p.scale) & 15) || (reinterpret_cast<uintptr_t>(p.bias) & 15)) return mit_set_error("mit_pgemm: scale / bias must be 16-byte aligned");
    const int a = p.act & 0xff;
    if ((a != MIT_ACT_NONE && a != MIT_ACT_RELU && a != MIT_ACT_GELU) || (p.act & ~(0xff | MIT_ACT_POST_FIRST)))
        return mit_set_error("mit_pgemm: activation %d is not one of none / relu / gelu", p.act);
    return 0;
}

// workgroups per CU of the persistent grid: -1 = the tile's own figure, 0 = one workgroup per output tile, n > 0 = n per CU
int pg_wgs_override() {
    static const int v = [] {
        const char *e = getenv("MIT_PGEMM_WGS");
        return (e && *e) ? atoi(e) : -1;
    }();
    return v;
}

int g_num_cus = 0;
int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

}  // namespace

int mit_pgemm_rows(const MitPGemm &d, const PgRowsExt &x, hipStream_t s) {
    MitPGemm p = d;
    p.Z = 1;
    if (!p.a_planes || !p.w_planes || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 16) || p.lda < p.M || p.ldw < p.N)
        return mit_set_error("mit_pgemm_rows: bad operands (M=%d N=%d K=%d)", p.M, p.N, p.K);
    uint16_t *planes = p.c_planes ? p.c_planes : x.also_planes;
    if (!p.c && !planes) return mit_set_error("mit_pgemm_rows: no output");
    if (p.c_planes && x.also_planes) return mit_set_error("mit_pgemm_rows: two planar outputs");
    if ((p.N & 3) || (planes && (p.N & 7)) || (p.c && (p.ldc & 3)) || (x.nsplit & 7) || (x.nhi & 3) || (x.c_dyn & 3) || (p.post && (p.ld_post & 3)) || (p.pre && (p.ld_pre & 3)))
        return mit_set_error("mit_pgemm_rows: N / strides must keep 16-byte cells whole");
    // the operand part of pg_check: the kernel addresses the three planes through 32-bit buffer offsets and moves 16-byte pieces
    if ((int64_t)3 * (p.K >> 3) * p.lda * 16 > 0xffffffffLL || (int64_t)3 * (p.K >> 3) * p.ldw * 16 > 0xffffffffLL)
        return mit_set_error("mit_pgemm_rows: an operand's three planes exceed 4 GB (32-bit piece offsets)");
    auto unaligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    if (unaligned(p.a_planes) || unaligned(p.w_planes) || unaligned(p.c) || unaligned(planes) || unaligned(p.scale) || unaligned(p.bias) ||
        unaligned(p.pre) || unaligned(p.post))
        return mit_set_error("mit_pgemm_rows: operands, outputs, scale / bias and pre / post must be 16-byte aligned");
    const int a = p.act & 0xff;
    if (a != MIT_ACT_NONE && a != MIT_ACT_RELU && a != MIT_ACT_GELU) return mit_set_error("mit_pgemm_rows: activation %d", p.act);
    if (p.nprod == 0) p.nprod = mit_gemm_mode_get();
    const double flops = 2.0 * p.M * (double)p.N * p.K;
    const double bytes = (double)p.M * p.K * 6.0 + (double)p.K * p.N * 6.0 + (double)p.M * p.N * ((p.c ? 4.0 : 0.0) + (planes ? 6.0 : 0.0));
    MitProbeScope probe("pgemm_rows_kernel", s, bytes, flops);
    if (x.splitk && p.K == 2048 && (p.nprod == 6 || p.nprod == 9)) {   // four waves x a quarter of K (see pgemm_rows_splitk_kernel)
        const int MT = (p.M + 31) / 32, NT = (p.N + 31) / 32, total = MT * NT, per = (total + 7) / 8;
        if (p.nprod == 6) hipLaunchKernelGGL((pgemm_rows_splitk_kernel<6, 6, 32>), dim3(per * 8), dim3(256), 0, s, p.a_planes, p.w_planes, (unsigned int)p.lda, (unsigned int)p.ldw, p.K, MT, NT, p, x);
        else hipLaunchKernelGGL((pgemm_rows_splitk_kernel<9, 6, 32>), dim3(per * 8), dim3(256), 0, s, p.a_planes, p.w_planes, (unsigned int)p.lda, (unsigned int)p.ldw, p.K, MT, NT, p, x);
    } else if (p.nprod == 6) pg_rows_launch_ext<6, PG_ROWS_DEPTH>(p, x, s);
    else if (p.nprod == 9) pg_rows_launch_ext<9, PG_ROWS_DEPTH>(p, x, s);
    else return mit_set_error("mit_pgemm_rows: nprod must be 6 or 9 (got %d)", p.nprod);
    MIT_CHECK_LAUNCH("mit_pgemm_rows");
    return 0;
}

extern "C" const char *mit_pgemm_tile_name(int tile) { return tile >= 0 && tile < kNumPgTiles ? kPgTiles[tile].name : nullptr; }

extern "C" int mit_pgemm_supported(const MitPGemm *d) { return d && pg_check(*d) == 0; }

extern "C" int mit_pgemm(const MitPGemm *d, void *stream) {
    if (!d) return mit_set_error("mit_pgemm: null descriptor");
    MitPGemm p = *d;
    if (pg_check(p)) return 1;
    if (p.nprod == 0) p.nprod = mit_gemm_mode_get();
    if (p.nprod != 6 && p.nprod != 9) return mit_set_error("mit_pgemm: nprod must be 6 or 9 (got %d)", p.nprod);
    int tile = p.tile;
    const bool planar = p.c_planes != nullptr;
    if (tile < 0) {
        const int r = p.N % 128;
        const bool narrow = p.N <= 64 || (r != 0 && r <= 64);
        tile = (narrow ? 1 : 0) + (planar ? 2 : 0) + (p.nprod == 9 ? 4 : 0);
    }
    if (tile >= kNumPgTiles) return mit_set_error("mit_pgemm: bad tile %d", tile);
    const PgTile &t = kPgTiles[tile];
    if ((t.outp != 0) != planar) return mit_set_error("mit_pgemm: tile %s does not produce the requested output kind", t.name);
    if (t.nprod != p.nprod) p.nprod = t.nprod;  // an explicit tile decides
    const int MT = (p.M + t.BM - 1) / t.BM, NT = (p.N + t.BN - 1) / t.BN, KT = p.K / 16;
    const int64_t tiles = (int64_t)MT * NT * p.Z;
    if (tiles > 0x7fffffffLL) return mit_set_error("mit_pgemm: too many tiles");
    // persistent grid: wgs_per_cu workgroups on every CU when there is more than that much work, else one workgroup per tile
    const int ov = pg_wgs_override();
    const int wgs = ov >= 0 ? ov : t.wgs_per_cu;
    int64_t grid = wgs > 0 ? (int64_t)wgs * num_cus() : tiles;
    if (grid > tiles) grid = tiles;
    grid = (grid + 7) / 8 * 8;
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    const double flops = 2.0 * p.M * (double)p.N * p.K * p.Z;
    const double bytes = (double)p.Z * ((double)p.M * p.K * 6.0 + (double)p.K * p.N * 6.0 + (double)p.M * p.N * (planar ? 6.0 : 4.0));
    MitProbeScope probe(t.probe, hs, bytes, flops);
    static const int order = [] {  // MIT_PGEMM_ORDER: 0 = a contiguous run of tiles per workgroup, 1 (default) = interleaved within the XCD
        const char *e = getenv("MIT_PGEMM_ORDER");
        return (e && *e) ? atoi(e) : 1;
    }();
    t.launch(p, MT, NT, KT, (int)tiles, (int)grid, order, hs);
    MIT_CHECK_LAUNCH("mit_pgemm");
    return 0;
}

extern "C" int mit_split_planes(const float *x_dev, int64_t ldx, int R, int K, uint16_t *planes_dev, int64_t ld, void *stream) {
    if (!x_dev || !planes_dev) return mit_set_error("mit_split_planes: null pointer");
    if (R <= 0 || K <= 0 || (K & 7) || (ldx & 3) || ld < R || (reinterpret_cast<uintptr_t>(x_dev) & 15) || (reinterpret_cast<uintptr_t>(planes_dev) & 15))
        return mit_set_error("mit_split_planes: need K %% 8 == 0, ldx %% 4 == 0, ld >= R, 16-byte aligned bases");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    MitProbeScope probe("split_planes_kernel", hs, (double)R * K * 10.0);
    hipLaunchKernelGGL(split_planes_kernel, dim3((R + 63) / 64, (K + 63) / 64), dim3(256), 0, hs, x_dev, ldx, R, K, reinterpret_cast<u32x4 *>(planes_dev), ld);
    MIT_CHECK_LAUNCH("mit_split_planes");
    return 0;
}

extern "C" int mit_join_planes(const uint16_t *planes_dev, int64_t ld, int R, int K, float *x_dev, int64_t ldx, void *stream) {
    if (!x_dev || !planes_dev) return mit_set_error("mit_join_planes: null pointer");
    if (R <= 0 || K <= 0 || (K & 7) || ld < R || (reinterpret_cast<uintptr_t>(planes_dev) & 15)) return mit_set_error("mit_join_planes: need K %% 8 == 0, ld >= R");
    const int64_t total = (int64_t)R * (K >> 3);
    hipLaunchKernelGGL(join_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const u32x4 *>(planes_dev), ld, R, K >> 3, x_dev, ldx);
    MIT_CHECK_LAUNCH("mit_join_planes");
    return 0;
}
