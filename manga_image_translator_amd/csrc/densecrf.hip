// densecrf.hip — the per-text-line DenseCRF of the mask refinement, all lines of a page in one batch on the GPU.
//
// Reference: manga_translator/mask_refinement/text_mask_utils.py:68-94 (refine_mask) calls pydensecrf — DenseCRF2D with a
// Gaussian pairwise term (sxy = 1, Potts weight 3) and a bilateral one (sxy = 23, srgb = 7, Potts weight 20), 5 mean-field
// iterations, argmax.  pydensecrf's source is not under /root/reference; this is the published algorithm of the library it
// wraps (Kraehenbuehl & Koltun 2011; permutohedral lattice of Adams, Baek & Davis 2010, densecrf's permutohedral.cpp):
//   init    per pixel: elevate the d-dim feature onto the lattice hyperplane, round to the nearest remainder-0 point, rank the
//           residuals, barycentric weights, and the d+1 enclosing simplex vertices -> a hash table of lattice points
//   filter  splat (weights x Q to the vertices), one [1/2, 1, 1/2] blur along each of the d+1 lattice axes, slice back,
//           times 1 / (1 + 2^-d)
//   update  Q = softmax(-unary + 3 * gauss(Q) + 20 * bilateral(Q))
// MI355X shape of it: one thread per pixel / per lattice slot over ALL crops of a page (a page's 10-30 lines would otherwise be
// ~100 launches each); the hash table is a flat array of 64-bit packed keys (12 bits per lattice coordinate) claimed with one
// atomicCAS — no locks, no key recomputation; every crop owns a private region of the table so crops cannot alias.  The splat
// accumulates in 64-bit fixed point (2^-32) with integer atomics, so the sums are exact, order-independent and reproducible
// run to run (float atomics would not be).  All of it is HBM/atomic-bound gather-scatter; nothing here is GEMM-shaped.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr unsigned long long EMPTY = ~0ull;
// Bits per lattice coordinate in the packed 64-bit hash key: 5 coordinates x 12 bits for the bilateral kernel (positions / 23, colours / 7:
// a few hundred lattice units at most), 2 x 24 bits for the Gaussian kernel (positions / 3: a crop taller than ~1100 px — a text line that
// was given a long panel border as one of its components — leaves a 12-bit range).
template <int D>
struct KeyBits {
    static constexpr int BITS = D <= 2 ? 24 : 12;
    static constexpr int HALF = 1 << (BITS - 1);
};
constexpr double FIX_SCALE = 4294967296.0;  // 2^32

struct LatticeConsts {
    float scale[5];  // diagonal of the elevation matrix E (Adams et al. p.5), times the lattice's expected std-dev
    float alpha;     // 1 / (1 + 2^-d)
};

__device__ __forceinline__ int find_segment(const int64_t *__restrict__ off, int n, int64_t i) {  // largest c with off[c] <= i
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

template <int D>
__device__ __forceinline__ bool pack_key(const int (&k)[D], unsigned long long *out) {
    unsigned long long p = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int v = k[i] + KeyBits<D>::HALF;
        ok = ok && v >= 0 && v < (1 << KeyBits<D>::BITS);
        p = (p << KeyBits<D>::BITS) | (unsigned long long)(v & ((1 << KeyBits<D>::BITS) - 1));
    }
    *out = p;
    return ok;
}

// ---- init: lattice coordinates of every pixel, hash-table insertion of its simplex vertices --------------------------------
template <int D>
__global__ __launch_bounds__(256) void crf_init_kernel(const uint8_t *__restrict__ page, int W, const MitCrfCrop *__restrict__ crops,
                                                        const int64_t *__restrict__ pt_off, const int64_t *__restrict__ tb_off, int n_crops,
                                                        int64_t NP, float sxy, float srgb, LatticeConsts lc,
                                                        unsigned long long *__restrict__ keys, int *__restrict__ offset,
                                                        float *__restrict__ bary_out, int *__restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NP) return;
    const int c = find_segment(pt_off, n_crops, i);
    const MitCrfCrop cr = crops[c];
    const int local = (int)(i - pt_off[c]);
    const int y = local / cr.w, x = local - y * cr.w;
    float f[D];
    f[0] = (float)x / sxy;
    f[1] = (float)y / sxy;
    if constexpr (D == 5) {
        const uint8_t *p = page + ((int64_t)(cr.y + y) * W + cr.x + x) * 3;
        f[2] = (float)p[0] / srgb;
        f[3] = (float)p[1] / srgb;
        f[4] = (float)p[2] / srgb;
    }
    float el[D + 1];
    float sm = 0.f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float cf = f[j - 1] * lc.scale[j - 1];
        el[j] = sm - (float)j * cf;
        sm += cf;
    }
    el[0] = sm;
    const float down = 1.0f / (float)(D + 1), up = (float)(D + 1);
    float rem0[D + 1];
    int sum = 0;
#pragma unroll
    for (int t = 0; t <= D; ++t) {
        const float v = down * el[t];
        const float u = ceilf(v) * up, d = floorf(v) * up;
        rem0[t] = (u - el[t] < el[t] - d) ? u : d;
        sum += (int)rintf(rem0[t] * down);
    }
    int rank[D + 1];
#pragma unroll
    for (int t = 0; t <= D; ++t) rank[t] = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
        const float da = el[a] - rem0[a];
#pragma unroll
        for (int b = a + 1; b <= D; ++b) {
            if (da < el[b] - rem0[b]) rank[a]++;
            else rank[b]++;
        }
    }
#pragma unroll
    for (int t = 0; t <= D; ++t) {
        rank[t] += sum;
        if (rank[t] < 0) {
            rank[t] += D + 1;
            rem0[t] += up;
        } else if (rank[t] > D) {
            rank[t] -= D + 1;
            rem0[t] -= up;
        }
    }
    float bary[D + 2];
#pragma unroll
    for (int t = 0; t <= D + 1; ++t) bary[t] = 0.f;
#pragma unroll
    for (int a = 0; a <= D; ++a) {
        const float v = (el[a] - rem0[a]) * down;
        const int pos = D - rank[a];
#pragma unroll
        for (int t = 0; t <= D + 1; ++t) {  // bary[pos] += v; bary[pos + 1] -= v  (x +- 0 is exact)
            bary[t] += (t == pos) ? v : 0.f;
            bary[t] -= (t == pos + 1) ? v : 0.f;
        }
    }
    bary[0] = (float)((1.0 + (double)bary[D + 1]) + (double)bary[0]);
    const int64_t base = tb_off[c];
    const uint64_t cap = (uint64_t)(tb_off[c + 1] - base);
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        int key[D];
#pragma unroll
        for (int a = 0; a < D; ++a) key[a] = (int)rem0[a] + ((rank[a] <= D - r) ? r : r - (D + 1));
        unsigned long long packed;
        if (!pack_key<D>(key, &packed)) atomicExch(overflow, 1);
        uint64_t h = mix64(packed) % cap;
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[base + h], EMPTY, packed);
            if (old == EMPTY || old == packed) break;
            h = (h + 1 == cap) ? 0 : h + 1;
        }
        offset[i * (D + 1) + r] = (int)(base + h);
        bary_out[i * (D + 1) + r] = bary[r];
    }
}

template <int D>
__device__ __forceinline__ int lookup(const unsigned long long *__restrict__ keys, int64_t base, uint64_t cap, const int (&k)[D]) {
    unsigned long long packed;
    if (!pack_key<D>(k, &packed)) return -1;
    uint64_t h = mix64(packed) % cap;
    for (;;) {
        const unsigned long long cur = keys[base + h];
        if (cur == packed) return (int)(base + h);
        if (cur == EMPTY) return -1;
        h = (h + 1 == cap) ? 0 : h + 1;
    }
}

// ---- blur neighbours of every lattice point along each of the d+1 axes -----------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void crf_neighbors_kernel(const unsigned long long *__restrict__ keys, const int64_t *__restrict__ tb_off,
                                                             int n_crops, int64_t CAP, int2 *__restrict__ nbr) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= CAP) return;
    const unsigned long long packed = keys[s];
    if (packed == EMPTY) return;
    const int c = find_segment(tb_off, n_crops, s);
    const int64_t base = tb_off[c];
    const uint64_t cap = (uint64_t)(tb_off[c + 1] - base);
    int key[D];
#pragma unroll
    for (int a = 0; a < D; ++a) key[a] = (int)((packed >> (KeyBits<D>::BITS * (D - 1 - a))) & ((1 << KeyBits<D>::BITS) - 1)) - KeyBits<D>::HALF;
#pragma unroll
    for (int j = 0; j <= D; ++j) {
        int n1[D], n2[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            n1[a] = key[a] - 1;
            n2[a] = key[a] + 1;
        }
        if (j < D) {
#pragma unroll
            for (int a = 0; a < D; ++a)
                if (a == j) {
                    n1[a] = key[a] + D;
                    n2[a] = key[a] - D;
                }
        }
        nbr[(int64_t)j * CAP + s] = make_int2(lookup<D>(keys, base, cap, n1), lookup<D>(keys, base, cap, n2));
    }
}

// ---- unary term and Q0 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crf_unary_kernel(const uint8_t *__restrict__ mask, const float *__restrict__ lut, int64_t NP,
                                                         float2 *__restrict__ unary, float2 *__restrict__ q) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NP) return;
    const int m = mask[i];
    const float u0 = lut[2 * m], u1 = lut[2 * m + 1];
    unary[i] = make_float2(u0, u1);
    const float t0 = -u0, t1 = -u1, mx = fmaxf(t0, t1);
    const float e0 = expf(t0 - mx), e1 = expf(t1 - mx), s = e0 + e1;
    q[i] = make_float2(e0 / s, e1 / s);
}

// ---- filter: splat / blur / slice ------------------------------------------------------------------------------------------
// Splat = 2 (D + 1) 64-bit fixed-point atomic adds per pixel (integer sums: bit-reproducible whatever the order).  The 256 consecutive
// pixels of a workgroup share most of their lattice vertices (the bilateral lattice's cells span ~20 pixels; the 6 vertices of a
// simplex are shared by every pixel inside it), and straight to global memory those adds queue up on a few hundred addresses: 754 us
// per iteration for the bilateral term of a page's crops, 44 % of the mask-refinement stage.  Here a workgroup first sums its 256 x
// (D + 1) contributions per vertex in an LDS hash table (open addressing, ds atomics), then adds each distinct vertex ONCE to global
// memory.  Same integers, same sums.
constexpr int SPLAT_TS = 2048;   // LDS slots (>= 256 (D + 1) = 1536 at D = 5: a probe sequence always ends)
template <int D>
__global__ __launch_bounds__(256) void crf_splat_kernel(const float2 *__restrict__ q, const int *__restrict__ offset,
                                                         const float *__restrict__ bary, int64_t NP, long long *__restrict__ acc) {
    static_assert(256 * (D + 1) <= SPLAT_TS * 3 / 4, "load factor");
    __shared__ int skey[SPLAT_TS];
    __shared__ unsigned long long sval[SPLAT_TS][2];
    for (int k = threadIdx.x; k < SPLAT_TS; k += 256) {
        skey[k] = -1;
        sval[k][0] = 0ull;
        sval[k][1] = 0ull;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NP) {
        const float2 v = q[i];
        int so[D + 1];
        float wo[D + 1];
#pragma unroll
        for (int r = 0; r <= D; ++r) {
            so[r] = offset[i * (D + 1) + r];
            wo[r] = bary[i * (D + 1) + r];
        }
#pragma unroll
        for (int r = 0; r <= D; ++r) {
            const int sidx = so[r];
            const unsigned long long fx = (unsigned long long)__double2ll_rn((double)(wo[r] * v.x) * FIX_SCALE);
            const unsigned long long fy = (unsigned long long)__double2ll_rn((double)(wo[r] * v.y) * FIX_SCALE);
            unsigned int h = ((unsigned int)sidx * 2654435761u) >> (32 - 11);   // 11 bits = SPLAT_TS
            for (;;) {
                const int old = atomicCAS(&skey[h], -1, sidx);
                if (old == -1 || old == sidx) {
                    atomicAdd(&sval[h][0], fx);
                    atomicAdd(&sval[h][1], fy);
                    break;
                }
                h = (h + 1) & (SPLAT_TS - 1);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SPLAT_TS; k += 256) {
        const int sidx = skey[k];
        if (sidx >= 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[2 * (int64_t)sidx]), sval[k][0]);
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[2 * (int64_t)sidx + 1]), sval[k][1]);
        }
    }
}

__device__ __forceinline__ float2 load_val(const long long *__restrict__ acc, const float2 *__restrict__ src, int first, int s) {
    if (s < 0) return make_float2(0.f, 0.f);
    if (first) return make_float2((float)((double)acc[2 * (int64_t)s] * (1.0 / FIX_SCALE)), (float)((double)acc[2 * (int64_t)s + 1] * (1.0 / FIX_SCALE)));
    return src[s];
}

// ---- dense numbering of the lattice points ------------------------------------------------------------------------------------
// The hash tables are sized for the worst case (2 (d + 1) slots per pixel: 9 M slots for a page's crops) while the bilateral lattice
// holds a few per cent of that: scanning the whole table in each of the d + 1 blur passes of every iteration (45 launches per call) and
// clearing 16 bytes per SLOT per iteration was 2.4 ms per page of the coupled path.  After the neighbour tables are built the occupied
// slots get consecutive ids (in whatever order the waves arrive: the ids only name the points — every sum is per point and the splat
// adds integers — so the results do not depend on it), the pixels' vertex offsets and the neighbour entries are rewritten to ids, and
// everything an iteration touches (accumulators, value buffers, blur) is as long as the number of points.
constexpr int COMPACT_PER_WAVE = 2048;   // slots a wave numbers with ONE add on the shared counter (one per 64 slots: 140 k adds on one address, 0.95 ms)
__global__ __launch_bounds__(256) void crf_compact_kernel(const unsigned long long *__restrict__ keys, int64_t CAP, int *__restrict__ dense_id,
                                                           int *__restrict__ counter) {
    const int lane = threadIdx.x & 63;
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * COMPACT_PER_WAVE;
    if (w0 >= CAP) return;
    int cnt = 0;
    for (int it = 0; it < COMPACT_PER_WAVE / 64; ++it) {
        const int64_t s = w0 + it * 64 + lane;
        cnt += __popcll(__ballot(s < CAP && keys[s] != EMPTY));
    }
    int base = 0;
    if (lane == 0 && cnt) base = atomicAdd(counter, cnt);
    base = __shfl(base, 0);
    for (int it = 0; it < COMPACT_PER_WAVE / 64; ++it) {
        const int64_t s = w0 + it * 64 + lane;
        const bool occ = s < CAP && keys[s] != EMPTY;
        const unsigned long long m = __ballot(occ);
        if (s < CAP) dense_id[s] = occ ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        base += __popcll(m);
    }
}

// active[id] = slot (written into the key table's memory: the keys are not read any more), the slot's neighbour entries become ids
__global__ __launch_bounds__(256) void crf_relabel_kernel(const int *__restrict__ dense_id, int64_t CAP, int ndir, int2 *__restrict__ nbr,
                                                           int *__restrict__ active) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= CAP) return;
    const int id = dense_id[s];
    if (id < 0) return;
    active[id] = (int)s;
    for (int j = 0; j < ndir; ++j) {
        int2 n = nbr[(int64_t)j * CAP + s];
        n.x = n.x < 0 ? -1 : dense_id[n.x];
        n.y = n.y < 0 ? -1 : dense_id[n.y];
        nbr[(int64_t)j * CAP + s] = n;
    }
}

__global__ __launch_bounds__(256) void crf_relabel_offsets_kernel(const int *__restrict__ dense_id, int64_t n, int *__restrict__ offset) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) offset[i] = dense_id[offset[i]];
}

// one blur pass over the NV lattice points (ids); nbr_j is indexed by SLOT and holds ids
__global__ __launch_bounds__(256) void crf_blur_kernel(const int *__restrict__ active, const int2 *__restrict__ nbr_j,
                                                        const long long *__restrict__ acc, const float2 *__restrict__ src,
                                                        float2 *__restrict__ dst, int first, int NV) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NV) return;
    const int2 n = nbr_j[active[i]];
    const float2 o = load_val(acc, src, first, i), a = load_val(acc, src, first, n.x), b = load_val(acc, src, first, n.y);
    const float sx = a.x + b.x, sy = a.y + b.y;
    dst[i] = make_float2((float)((double)o.x + 0.5 * (double)sx), (float)((double)o.y + 0.5 * (double)sy));
}

template <int D>
__global__ __launch_bounds__(256) void crf_slice_kernel(const float2 *__restrict__ val, const int *__restrict__ offset,
                                                         const float *__restrict__ bary, int64_t NP, float alpha, float2 *__restrict__ msg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NP) return;
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        const float2 v = val[offset[i * (D + 1) + r]];
        const float w = bary[i * (D + 1) + r];
        ox += (w * v.x) * alpha;
        oy += (w * v.y) * alpha;
    }
    msg[i] = make_float2(ox, oy);
}

// ---- mean-field update (DenseCRF::inference + PottsCompatibility::apply) and the final argmax ------------------------------
__global__ __launch_bounds__(256) void crf_update_kernel(const float2 *__restrict__ unary, const float2 *__restrict__ msg_g,
                                                          const float2 *__restrict__ msg_b, float wg, float wb, int64_t NP,
                                                          float2 *__restrict__ q, uint8_t *__restrict__ out_mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NP) return;
    const float2 u = unary[i], g = msg_g[i], b = msg_b[i];
    float t0 = -u.x, t1 = -u.y;
    t0 = t0 - (-wg * g.x);
    t1 = t1 - (-wg * g.y);
    t0 = t0 - (-wb * b.x);
    t1 = t1 - (-wb * b.y);
    const float mx = fmaxf(t0, t1);
    const float e0 = expf(t0 - mx), e1 = expf(t1 - mx), s = e0 + e1;
    const float q0 = e0 / s, q1 = e1 / s;
    q[i] = make_float2(q0, q1);
    if (out_mask) out_mask[i] = q1 > q0 ? 255 : 0;  // np.argmax: the first maximum wins a tie
}

struct Layout {
    int64_t NP, cap2, cap5;
    size_t crops, pt_off, tb2_off, tb5_off, overflow, counts, unary, q, msg_g, msg_b, off2, bary2, off5, bary5, keys2, keys5, nbr2, nbr5, acc, valA, valB,
        total;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int make_layout(const MitCrfCrop *crops, int n, Layout *L, std::vector<int64_t> *pt, std::vector<int64_t> *t2, std::vector<int64_t> *t5) {
    pt->assign(n + 1, 0);
    t2->assign(n + 1, 0);
    t5->assign(n + 1, 0);
    for (int c = 0; c < n; ++c) {
        if (crops[c].w <= 0 || crops[c].h <= 0 || crops[c].x < 0 || crops[c].y < 0) return 1;
        const int64_t np = (int64_t)crops[c].w * crops[c].h;
        (*pt)[c + 1] = (*pt)[c] + np;
        (*t2)[c + 1] = (*t2)[c] + 2 * 3 * np;
        (*t5)[c + 1] = (*t5)[c] + 2 * 6 * np;
    }
    L->NP = (*pt)[n];
    L->cap2 = (*t2)[n];
    L->cap5 = (*t5)[n];
    if (L->cap5 >= (int64_t)1 << 31) return 2;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align_up(o + bytes);
        return at;
    };
    L->crops = take(sizeof(MitCrfCrop) * n);
    L->pt_off = take(8 * (size_t)(n + 1));
    L->tb2_off = take(8 * (size_t)(n + 1));
    L->tb5_off = take(8 * (size_t)(n + 1));
    L->overflow = take(4);
    L->counts = take(8);
    L->unary = take(8 * (size_t)L->NP);
    L->q = take(8 * (size_t)L->NP);
    L->msg_g = take(8 * (size_t)L->NP);
    L->msg_b = take(8 * (size_t)L->NP);
    L->off2 = take(4 * 3 * (size_t)L->NP);
    L->bary2 = take(4 * 3 * (size_t)L->NP);
    L->off5 = take(4 * 6 * (size_t)L->NP);
    L->bary5 = take(4 * 6 * (size_t)L->NP);
    L->keys2 = take(8 * (size_t)L->cap2);
    L->keys5 = take(8 * (size_t)L->cap5);
    L->nbr2 = take(8 * 3 * (size_t)L->cap2);
    L->nbr5 = take(8 * 6 * (size_t)L->cap5);
    L->acc = take(16 * (size_t)L->cap5);
    L->valA = take(8 * (size_t)L->cap5);
    L->valB = take(8 * (size_t)L->cap5);
    L->total = o;
    return 0;
}

LatticeConsts lattice_consts(int d) {
    LatticeConsts lc;
    memset(&lc, 0, sizeof(lc));
    const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (d + 1));
    for (int i = 0; i < d; ++i) lc.scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);
    lc.alpha = 1.0f / (1.0f + powf(2.0f, (float)-d));
    return lc;
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int64_t mit_densecrf_workspace_bytes(const MitCrfCrop *crops, int n_crops) {
    if (!crops || n_crops <= 0) return -1;
    Layout L;
    std::vector<int64_t> pt, t2, t5;
    if (make_layout(crops, n_crops, &L, &pt, &t2, &t5)) return -1;
    return (int64_t)L.total;
}

extern "C" int mit_densecrf_refine(const uint8_t *page_dev, int H, int W, const MitCrfCrop *crops, int n_crops, const uint8_t *mask_dev,
                                   uint8_t *out_dev, float *q_dev, float sxy_gauss, float w_gauss, float sxy_bilateral, float srgb_bilateral,
                                   float w_bilateral, int iterations, const float *unary_lut_dev, void *workspace_dev, int64_t workspace_bytes,
                                   void *stream) {
    if (!page_dev || !crops || !mask_dev || !out_dev || !unary_lut_dev || !workspace_dev) return mit_set_error("mit_densecrf_refine: null pointer");
    if (n_crops <= 0 || iterations < 1 || H <= 0 || W <= 0) return mit_set_error("mit_densecrf_refine: bad arguments");
    if (!(sxy_gauss > 0) || !(sxy_bilateral > 0) || !(srgb_bilateral > 0)) return mit_set_error("mit_densecrf_refine: kernel widths must be positive");
    for (int c = 0; c < n_crops; ++c)
        if (crops[c].x < 0 || crops[c].y < 0 || crops[c].w <= 0 || crops[c].h <= 0 || crops[c].x + crops[c].w > W || crops[c].y + crops[c].h > H)
            return mit_set_error("mit_densecrf_refine: crop %d (%d, %d, %d x %d) is empty or outside the %d x %d page", c, crops[c].x, crops[c].y,
                                 crops[c].w, crops[c].h, W, H);
    Layout L;
    std::vector<int64_t> pt, t2, t5;
    const int lay = make_layout(crops, n_crops, &L, &pt, &t2, &t5);
    if (lay) return mit_set_error("mit_densecrf_refine: batch too large for 32-bit lattice indices (split the crops)");
    if ((int64_t)L.total > workspace_bytes) return mit_set_error("mit_densecrf_refine: workspace too small (%lld < %zu bytes)", (long long)workspace_bytes, L.total);
    if (reinterpret_cast<uintptr_t>(workspace_dev) & 255) return mit_set_error("mit_densecrf_refine: workspace must be 256-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_dev);
    auto at = [&](size_t off) { return static_cast<void *>(ws + off); };
    // small tables (pageable host memory: hipMemcpyAsync stages them before returning)
    MIT_CHECK_HIP(hipMemcpyAsync(at(L.crops), crops, sizeof(MitCrfCrop) * n_crops, hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipMemcpyAsync(at(L.pt_off), pt.data(), 8 * (size_t)(n_crops + 1), hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipMemcpyAsync(at(L.tb2_off), t2.data(), 8 * (size_t)(n_crops + 1), hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipMemcpyAsync(at(L.tb5_off), t5.data(), 8 * (size_t)(n_crops + 1), hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));  // the host vectors die with this call
    MIT_CHECK_HIP(hipMemsetAsync(at(L.overflow), 0, 4, st));
    MIT_CHECK_HIP(hipMemsetAsync(at(L.keys2), 0xff, 8 * (size_t)L.cap2, st));
    MIT_CHECK_HIP(hipMemsetAsync(at(L.keys5), 0xff, 8 * (size_t)L.cap5, st));
    MIT_CHECK_HIP(hipMemsetAsync(at(L.counts), 0, 8, st));
    const MitCrfCrop *d_crops = static_cast<const MitCrfCrop *>(at(L.crops));
    const int64_t *d_pt = static_cast<const int64_t *>(at(L.pt_off)), *d_t2 = static_cast<const int64_t *>(at(L.tb2_off)),
                  *d_t5 = static_cast<const int64_t *>(at(L.tb5_off));
    int *d_over = static_cast<int *>(at(L.overflow));
    float2 *unary = static_cast<float2 *>(at(L.unary)), *q = static_cast<float2 *>(at(L.q)), *msg_g = static_cast<float2 *>(at(L.msg_g)),
           *msg_b = static_cast<float2 *>(at(L.msg_b));
    int *off2 = static_cast<int *>(at(L.off2)), *off5 = static_cast<int *>(at(L.off5));
    float *bary2 = static_cast<float *>(at(L.bary2)), *bary5 = static_cast<float *>(at(L.bary5));
    unsigned long long *keys2 = static_cast<unsigned long long *>(at(L.keys2)), *keys5 = static_cast<unsigned long long *>(at(L.keys5));
    int2 *nbr2 = static_cast<int2 *>(at(L.nbr2)), *nbr5 = static_cast<int2 *>(at(L.nbr5));
    long long *acc = static_cast<long long *>(at(L.acc));
    float2 *valA = static_cast<float2 *>(at(L.valA)), *valB = static_cast<float2 *>(at(L.valB));
    const LatticeConsts lc2 = lattice_consts(2), lc5 = lattice_consts(5);
    const int64_t NP = L.NP;
    {
        // algorithmic bytes of the whole refinement: per pixel and iteration, both filters read Q and write a message (16 B each) and
        // touch their (d+1) vertex values twice (splat + slice, 8 B each); the lattice blur reads/writes every slot d+1 times
        const double per_iter = (double)NP * (2 * 16.0 + (3 + 6) * 2 * 8.0) + 16.0 * (3.0 * L.cap2 + 6.0 * L.cap5) * 0.5;
        MitProbeScope probe("densecrf_refine", st, per_iter * iterations);
        hipLaunchKernelGGL(crf_init_kernel<2>, dim3(blocks(NP)), dim3(256), 0, st, page_dev, W, d_crops, d_pt, d_t2, n_crops, NP, sxy_gauss, 1.0f, lc2,
                           keys2, off2, bary2, d_over);
        hipLaunchKernelGGL(crf_init_kernel<5>, dim3(blocks(NP)), dim3(256), 0, st, page_dev, W, d_crops, d_pt, d_t5, n_crops, NP, sxy_bilateral,
                           srgb_bilateral, lc5, keys5, off5, bary5, d_over);
        hipLaunchKernelGGL(crf_neighbors_kernel<2>, dim3(blocks(L.cap2)), dim3(256), 0, st, keys2, d_t2, n_crops, L.cap2, nbr2);
        hipLaunchKernelGGL(crf_neighbors_kernel<5>, dim3(blocks(L.cap5)), dim3(256), 0, st, keys5, d_t5, n_crops, L.cap5, nbr5);
        hipLaunchKernelGGL(crf_unary_kernel, dim3(blocks(NP)), dim3(256), 0, st, mask_dev, unary_lut_dev, NP, unary, q);
        // dense ids (see crf_compact_kernel): the id tables sit in the accumulator area until the first iteration clears it, the id ->
        // slot lists take the key tables' place
        int *d_counts = static_cast<int *>(at(L.counts));
        int *id5 = reinterpret_cast<int *>(acc), *id2 = id5 + L.cap5;
        int *act2 = reinterpret_cast<int *>(keys2), *act5 = reinterpret_cast<int *>(keys5);
        hipLaunchKernelGGL(crf_compact_kernel, dim3((unsigned)((L.cap2 + 4 * COMPACT_PER_WAVE - 1) / (4 * COMPACT_PER_WAVE))), dim3(256), 0, st, keys2, L.cap2, id2, d_counts);
        hipLaunchKernelGGL(crf_compact_kernel, dim3((unsigned)((L.cap5 + 4 * COMPACT_PER_WAVE - 1) / (4 * COMPACT_PER_WAVE))), dim3(256), 0, st, keys5, L.cap5, id5, d_counts + 1);
        hipLaunchKernelGGL(crf_relabel_kernel, dim3(blocks(L.cap2)), dim3(256), 0, st, id2, L.cap2, 3, nbr2, act2);
        hipLaunchKernelGGL(crf_relabel_kernel, dim3(blocks(L.cap5)), dim3(256), 0, st, id5, L.cap5, 6, nbr5, act5);
        hipLaunchKernelGGL(crf_relabel_offsets_kernel, dim3(blocks(3 * NP)), dim3(256), 0, st, id2, 3 * NP, off2);
        hipLaunchKernelGGL(crf_relabel_offsets_kernel, dim3(blocks(6 * NP)), dim3(256), 0, st, id5, 6 * NP, off5);
        int nv[2] = {0, 0};
        MIT_CHECK_HIP(hipMemcpyAsync(nv, d_counts, 8, hipMemcpyDeviceToHost, st));
        MIT_CHECK_HIP(hipStreamSynchronize(st));
        const int nv2 = nv[0], nv5 = nv[1], nvmax = nv2 > nv5 ? nv2 : nv5;
        if (nv2 <= 0 || nv5 <= 0 || nv2 > L.cap2 || nv5 > L.cap5) return mit_set_error("mit_densecrf_refine: lattice point count out of range (%d, %d)", nv2, nv5);
        MIT_CHECK_HIP(hipMemsetAsync(valA, 0, 8 * (size_t)nvmax, st));
        MIT_CHECK_HIP(hipMemsetAsync(valB, 0, 8 * (size_t)nvmax, st));
        for (int it = 0; it < iterations; ++it) {
            // Gaussian term (d = 2)
            MIT_CHECK_HIP(hipMemsetAsync(acc, 0, 16 * (size_t)nv2, st));
            hipLaunchKernelGGL(crf_splat_kernel<2>, dim3(blocks(NP)), dim3(256), 0, st, q, off2, bary2, NP, acc);
            float2 *src = valA, *dst = valB;
            for (int j = 0; j <= 2; ++j) {
                hipLaunchKernelGGL(crf_blur_kernel, dim3(blocks(nv2)), dim3(256), 0, st, act2, nbr2 + (int64_t)j * L.cap2, acc, src, dst, j == 0, nv2);
                float2 *t = src;
                src = dst;
                dst = t;
            }
            hipLaunchKernelGGL(crf_slice_kernel<2>, dim3(blocks(NP)), dim3(256), 0, st, src, off2, bary2, NP, lc2.alpha, msg_g);
            // bilateral term (d = 5)
            MIT_CHECK_HIP(hipMemsetAsync(acc, 0, 16 * (size_t)nv5, st));
            hipLaunchKernelGGL(crf_splat_kernel<5>, dim3(blocks(NP)), dim3(256), 0, st, q, off5, bary5, NP, acc);
            src = valA;
            dst = valB;
            for (int j = 0; j <= 5; ++j) {
                hipLaunchKernelGGL(crf_blur_kernel, dim3(blocks(nv5)), dim3(256), 0, st, act5, nbr5 + (int64_t)j * L.cap5, acc, src, dst, j == 0, nv5);
                float2 *t = src;
                src = dst;
                dst = t;
            }
            hipLaunchKernelGGL(crf_slice_kernel<5>, dim3(blocks(NP)), dim3(256), 0, st, src, off5, bary5, NP, lc5.alpha, msg_b);
            hipLaunchKernelGGL(crf_update_kernel, dim3(blocks(NP)), dim3(256), 0, st, unary, msg_g, msg_b, w_gauss, w_bilateral, NP, q,
                               it == iterations - 1 ? out_dev : static_cast<uint8_t *>(nullptr));
        }
    }
    MIT_CHECK_LAUNCH("mit_densecrf_refine");
    if (q_dev) MIT_CHECK_HIP(hipMemcpyAsync(q_dev, q, 8 * (size_t)NP, hipMemcpyDeviceToDevice, st));
    int overflow = 0;
    MIT_CHECK_HIP(hipMemcpyAsync(&overflow, d_over, 4, hipMemcpyDeviceToHost, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));
    if (overflow) return mit_set_error("mit_densecrf_refine: a lattice coordinate left the key range of the packed hash key (crop far larger than a page)");
    return 0;
}
