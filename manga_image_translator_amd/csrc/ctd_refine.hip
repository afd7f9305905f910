// ctd_refine.hip — the ctd detector's refine_mask on the GPU, all text lines of a page per launch.
//
// Reference: manga_translator/detection/ctd_utils/textmask.py:158-174 (refine_mask) -> :56-71 get_topk_masklist, :44-54
// get_otsuthresh_masklist, :29-42 minxor_thresh, :74-132 merge_mask_list.  Per text-line window: candidate binarisations of
// the crop (the 3 dominant grey levels +-30, an Otsu split of the best colour channel), each taken as is or complemented
// — whichever is closer (byte-xor sum) to the network's mask — then merged connected component by connected component
// wherever that brings the result closer to the eroded, thresholded prediction, and the holes of the result filled the same
// way.  The reference does this with cv2 (cvtColor, erode, inRange, threshold, connectedComponentsWithStats, bitwise_xor) on
// the CPU, ~10^5 small array operations per page.
//
// Observation that makes it parallel: with 0/255 images, "merge component c if sum(xor(merged | c, pred)) < sum(xor(merged,
// pred))" only depends on c's own not-yet-merged pixels (#pred set > #pred clear among them), and the components of ONE
// candidate mask are disjoint — so every component of a candidate is decided independently and only the (at most 4)
// candidates of a line are sequential.  Connected components are a lock-free union-find over the concatenated crops of all
// lines (atomicMin links, 8-connectivity, links never cross a crop); per-component statistics are integer atomics on the root
// pixel, so the result does not depend on scheduling.  Three phases with two tiny host round trips (per-line 256-bin
// histograms -> numpy's own histogram / top-k / Otsu arithmetic; per-candidate xor sums -> candidate order).  Integer work,
// atomic/HBM-bound; bit-identical to hostglue.refine_mask (which is pinned to the reference's textmask.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

struct Geo {
    const MitRefineWindow *win;  // device
    const int64_t *pt_off;       // device, n + 1
    int n;
    int64_t P;
};

__device__ __forceinline__ int find_line(const int64_t *__restrict__ off, int n, int64_t i) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Adds 1 to counters[key] for every lane with `on` set, one atomic per distinct key in the wave (adjacent pixels mostly share
// a grey level / a component root, so plain per-lane atomics would serialise on one address).  Must be called by all 64 lanes.
__device__ __forceinline__ void wave_count(int *__restrict__ counters, int64_t key, bool on) {
    unsigned long long todo = __ballot(on);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int64_t k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(on && key == k0);
        if (lane == leader) atomicAdd(&counters[k0], __popcll(same));
        todo &= ~same;
    }
}

// The same for a 64-bit sum of per-lane values.
__device__ __forceinline__ void wave_sum(unsigned long long *__restrict__ sums, int key, bool on, int v) {
    unsigned long long todo = __ballot(on);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, leader);
        const bool mine = on && key == k0;
        const unsigned long long same = __ballot(mine);
        int t = mine ? v : 0;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == leader) atomicAdd(&sums[k0], (unsigned long long)t);
        todo &= ~same;
    }
}

// ---- phase A: grey level, eroded prediction, per-line histograms -----------------------------------------------------------
// hist[line][0] = grey levels under erode3x3(msk) > 127 (textmask.py:58-60); hist[line][1 + c] = channel c over the crop (Otsu).
__global__ __launch_bounds__(256) void refine_prepare_kernel(const uint8_t *__restrict__ page, const uint8_t *__restrict__ pred, int W, Geo g,
                                                              int *__restrict__ line_of, uint8_t *__restrict__ grey,
                                                              uint8_t *__restrict__ pred_bin, int *__restrict__ hist) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = p < g.P;
    int l = 0, gr = 0, c0 = 0, c1 = 0, c2 = 0, er = 0;
    if (ok) {
        l = find_line(g.pt_off, g.n, p);
        const MitRefineWindow w = g.win[l];
        const int cw = w.x2 - w.x1, ch = w.y2 - w.y1;
        const int local = (int)(p - g.pt_off[l]);
        const int y = local / cw, x = local - y * cw;
        line_of[p] = l;
        const uint8_t *px = page + ((int64_t)(w.y1 + y) * W + w.x1 + x) * 3;
        c0 = px[0], c1 = px[1], c2 = px[2];
        gr = (c0 * 1868 + c1 * 9617 + c2 * 4899 + 8192) >> 14;  // cv2.COLOR_BGR2GRAY applied to the RGB page, as the reference does
        grey[p] = (uint8_t)gr;
        const uint8_t *m = pred + (int64_t)(w.y1 + y) * W + w.x1 + x;
        int cr = m[0];  // erosion inside the crop: outside counts as +inf (cv2.erode's default border)
        er = 255;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if (y + dy < 0 || y + dy >= ch || x + dx < 0 || x + dx >= cw) continue;
                const int v = m[(int64_t)dy * W + dx];
                er = min(er, v);
                if (dx == 0 || dy == 0) cr = min(cr, v);
            }
        pred_bin[p] = cr > 60 ? 255 : 0;  // merge_mask_list: erode(cross 3x3), threshold(pred_thresh * 2 = 60) (:77-81)
    }
    // The four histograms of the workgroup's FIRST line are counted in LDS (a workgroup's 256 consecutive pixels belong to one line, two
    // at a line's end) and added to the line's 1024 global counters once per bin; pixels of a following line go to global memory
    // directly.  (Counted per wave with one global atomic per distinct key — up to 64 rounds of ballots per histogram on a textured crop —
    // this kernel was 1.05 ms per page of the coupled path.)  Integer counts: the same histograms.
    __shared__ int sh[1024];
    __shared__ int l0s;
    for (int b = threadIdx.x; b < 1024; b += 256) sh[b] = 0;
    if (threadIdx.x == 0) l0s = l;   // (thread 0 holds the block's first pixel; blocks are only launched for p0 < P)
    __syncthreads();
    const int l0 = l0s;
    if (ok && l == l0) {
        if (er > 127) atomicAdd(&sh[gr], 1);
        atomicAdd(&sh[256 + c0], 1);
        atomicAdd(&sh[512 + c1], 1);
        atomicAdd(&sh[768 + c2], 1);
    } else if (ok) {
        int *dst = hist + (int64_t)l * 1024;
        if (er > 127) atomicAdd(dst + gr, 1);
        atomicAdd(dst + 256 + c0, 1);
        atomicAdd(dst + 512 + c1, 1);
        atomicAdd(dst + 768 + c2, 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 1024; b += 256) {
        const int v = sh[b];
        if (v) atomicAdd(hist + (int64_t)l0 * 1024 + b, v);
    }
}

// candidate k of a line: kind 0 = none, 1 = inRange(grey, lo, hi), 2 + c = channel c > lo; invert = take the complement
__device__ __forceinline__ int cand_bit(const MitRefineCand &c, int gr, const uint8_t *px) {
    int t;
    if (c.kind == 1) t = gr >= c.lo && gr <= c.hi;
    else t = px[c.kind - 2] > c.lo;
    return t ^ c.invert;
}

// ---- phase B: xor sums of the six raw candidates against the crop of the network's mask -------------------------------------
// sums[line][k] = sum over the crop of (threshed_k ^ msk) as bytes; the complement's sum is 255 * npix - that.
__global__ __launch_bounds__(256) void refine_score_kernel(const uint8_t *__restrict__ page, const uint8_t *__restrict__ pred, int W, Geo g,
                                                            const int *__restrict__ line_of, const uint8_t *__restrict__ grey,
                                                            const MitRefineCand *__restrict__ cands, unsigned long long *__restrict__ sums) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = p < g.P;
    int l = 0, m = 0, gr = 0;
    const uint8_t *px = page;
    if (ok) {
        l = line_of[p];
        const MitRefineWindow w = g.win[l];
        const int cw = w.x2 - w.x1;
        const int local = (int)(p - g.pt_off[l]);
        const int y = local / cw, x = local - y * cw;
        px = page + ((int64_t)(w.y1 + y) * W + w.x1 + x) * 3;
        m = pred[(int64_t)(w.y1 + y) * W + w.x1 + x];
        gr = grey[p];
    }
    // sums of the workgroup's first line in LDS, one global add per candidate and workgroup (as refine_prepare_kernel's histograms)
    __shared__ unsigned long long ssum[6];
    __shared__ int l0s;
    if (threadIdx.x < 6) ssum[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) l0s = l;
    __syncthreads();
    const int l0 = l0s;
    for (int k = 0; k < 6; ++k) {
        MitRefineCand c = cands[l * 6 + k];
        const bool on = ok && c.kind != 0;
        if (!on) c.kind = 1;
        const int t = cand_bit(c, gr, px);
        const int v = t ? 255 - m : m;
        int tv = (on && l == l0) ? v : 0;   // wave-level sum first (every lane takes part): one LDS add per wave and candidate
        for (int o = 32; o > 0; o >>= 1) tv += __shfl_xor(tv, o);
        if ((threadIdx.x & 63) == 0 && tv) atomicAdd(&ssum[k], (unsigned long long)tv);
        if (on && l != l0) atomicAdd(&sums[l * 6 + k], (unsigned long long)v);
    }
    __syncthreads();
    if (threadIdx.x < 6 && ssum[threadIdx.x]) atomicAdd(&sums[l0 * 6 + threadIdx.x], ssum[threadIdx.x]);
}

// ---- phase C: component-wise merge -----------------------------------------------------------------------------------------
struct CompStats {
    int *area, *gain, *loss, *hv;
};

__device__ __forceinline__ int uf_find(const int *__restrict__ L, int a) {
    int r = a;
    for (;;) {
        const int q = __atomic_load_n(&L[r], __ATOMIC_RELAXED);
        if (q == r) return r;
        r = q;
    }
}

__device__ __forceinline__ void uf_union(int *__restrict__ L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[a], b);  // a > b: hang the larger root under the smaller
        if (old == a) return;
        a = old;  // someone re-rooted a meanwhile: retry from there
    }
}

// mode 0: candidate slot s of each line; mode 1: the complement of merged (hole filling)
__global__ __launch_bounds__(256) void refine_label_init_kernel(const uint8_t *__restrict__ page, int W, Geo g, const int *__restrict__ line_of,
                                                                 const uint8_t *__restrict__ grey, const MitRefineCand *__restrict__ order,
                                                                 int slot, int mode, const uint8_t *__restrict__ merged,
                                                                 uint8_t *__restrict__ cand, int *__restrict__ L, CompStats st,
                                                                 int *__restrict__ zero_area) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = p < g.P;
    const int l = ok ? line_of[p] : 0;
    int t = 1, x = 0;
    if (ok) {
        const MitRefineWindow w = g.win[l];
        const int cw = w.x2 - w.x1;
        const int local = (int)(p - g.pt_off[l]);
        const int y = local / cw;
        x = local - y * cw;
        if (mode == 0) {
            const MitRefineCand c = order[l * 4 + slot];
            t = c.kind == 0 ? 0 : cand_bit(c, grey[p], page + ((int64_t)(w.y1 + y) * W + w.x1 + x) * 3);
        } else {
            t = merged[p] ? 0 : 1;
        }
    }
    if (mode == 1) wave_count(zero_area, l, ok && !t);
    // The 64 pixels of a wave are consecutive: a horizontal run of candidates inside it starts its life already linked (every pixel
    // points at the run's first pixel), so refine_label_link_kernel only has to join runs across wave boundaries and rows.
    const int lane = threadIdx.x & 63;
    const unsigned long long cm = __ballot(ok && t);
    const bool start = ok && t && (lane == 0 || !((cm >> (lane - 1)) & 1ull) || x == 0);
    const unsigned long long sm = __ballot(start);
    if (!ok) return;
    cand[p] = (uint8_t)t;
    L[p] = t ? (int)p - (lane - (63 - __clzll((long long)(sm & ((2ull << lane) - 1ull))))) : -1;
    st.area[p] = 0;
    st.gain[p] = 0;
    st.loss[p] = 0;
    st.hv[p] = 0;
}

__global__ __launch_bounds__(256) void refine_label_link_kernel(Geo g, const int *__restrict__ line_of, const uint8_t *__restrict__ cand,
                                                                 int *__restrict__ L) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.P || !cand[p]) return;
    const int l = line_of[p];
    const MitRefineWindow w = g.win[l];
    const int cw = w.x2 - w.x1;
    const int local = (int)(p - g.pt_off[l]);
    const int y = local / cw, x = local - y * cw;
    // Only the unions that can join two sets: (left) runs are pre-linked inside a wave, so only its first lane looks left; (up) not
    // when the left neighbour and the pixel above it are candidates too — the left neighbour (or a pixel further left along the two
    // solid rows) makes that link; (diagonals) only when neither the pixel above nor the horizontal neighbour below the diagonal can.
    const bool left = x > 0 && cand[p - 1];
    if (left && (threadIdx.x & 63) == 0) uf_union(L, (int)p, (int)p - 1);
    if (y > 0) {
        const bool up = cand[p - cw] != 0, ul = x > 0 && cand[p - cw - 1], ur = x + 1 < cw && cand[p - cw + 1];
        if (up && !(left && ul)) uf_union(L, (int)p, (int)p - cw);
        if (ul && !up && !left) uf_union(L, (int)p, (int)p - cw - 1);
        if (ur && !up && !(x + 1 < cw && cand[p + 1])) uf_union(L, (int)p, (int)p - cw + 1);
    }
}

// every pixel points at its root: the statistics and apply passes then find it in one step
__global__ __launch_bounds__(256) void refine_label_flatten_kernel(int64_t P, const uint8_t *__restrict__ cand, int *__restrict__ L) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || !cand[p]) return;
    const int r = uf_find(L, (int)p);
    if (r != (int)p) __atomic_store_n(&L[p], r, __ATOMIC_RELAXED);
}

__global__ __launch_bounds__(256) void refine_label_stats_kernel(Geo g, const int *__restrict__ line_of, const uint8_t *__restrict__ cand,
                                                                  int *__restrict__ L, const uint8_t *__restrict__ merged,
                                                                  const uint8_t *__restrict__ pred_bin, CompStats st) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = p < g.P && cand[p];
    int r = 0;
    bool fresh = false, set = false, hv = false;
    if (on) {
        r = uf_find(L, (int)p);
        fresh = !merged[p];
        set = pred_bin[p] != 0;
        const int l = line_of[p];
        const MitRefineWindow w = g.win[l];
        const int cw = w.x2 - w.x1, ch = w.y2 - w.y1;
        const int local = (int)(p - g.pt_off[l]);
        const int y = local / cw, x = local - y * cw;
        hv = (x > 0 && cand[p - 1]) || (x + 1 < cw && cand[p + 1]) || (y > 0 && cand[p - cw]) || (y + 1 < ch && cand[p + cw]);
    }
    wave_count(st.area, r, on);
    wave_count(st.gain, r, on && fresh && set);
    wave_count(st.loss, r, on && fresh && !set);
    if (on && hv) st.hv[r] = 1;  // every writer stores the same value
}

// candidate pass: a component joins when it is not a speck (bounding box w * h < 3 <=> one pixel, or two pixels side by side) and
// more of its not-yet-merged pixels are set in the prediction than clear
__global__ __launch_bounds__(256) void refine_apply_kernel(int64_t P, const uint8_t *__restrict__ cand, const int *__restrict__ L, CompStats st,
                                                            uint8_t *__restrict__ merged) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || !cand[p]) return;
    const int r = uf_find(L, (int)p);
    const int a = st.area[r];
    if (a == 1 || (a == 2 && st.hv[r])) return;
    if (st.gain[r] > st.loss[r]) merged[p] = 255;
}

// hole pass, step 1: the largest complement-component area of each line, then how many share it and the largest below it
__global__ __launch_bounds__(256) void refine_hole_max_kernel(int64_t P, const int *__restrict__ line_of, const uint8_t *__restrict__ cand,
                                                               const int *__restrict__ L, CompStats st, int *__restrict__ max1) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || !cand[p] || L[p] != (int)p) return;
    atomicMax(&max1[line_of[p]], st.area[p]);
}

__global__ __launch_bounds__(256) void refine_hole_second_kernel(int64_t P, const int *__restrict__ line_of, const uint8_t *__restrict__ cand,
                                                                  const int *__restrict__ L, CompStats st, const int *__restrict__ max1,
                                                                  int *__restrict__ cnt_eq, int *__restrict__ max_lt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || !cand[p] || L[p] != (int)p) return;
    const int l = line_of[p], a = st.area[p];
    if (a == max1[l]) atomicAdd(&cnt_eq[l], 1);
    else atomicMax(&max_lt[l], a);
}

// hole pass, step 2: threshold = second largest of {area of the mask itself} + {complement-component areas} (the largest when
// there is only one entry); components below it are tried like candidates (textmask.py:113-131)
__global__ __launch_bounds__(256) void refine_hole_apply_kernel(int64_t P, const int *__restrict__ line_of, const uint8_t *__restrict__ cand,
                                                                 const int *__restrict__ L, CompStats st, const int *__restrict__ zero_area,
                                                                 const int *__restrict__ max1, const int *__restrict__ cnt_eq,
                                                                 const int *__restrict__ max_lt, uint8_t *__restrict__ merged) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || !cand[p]) return;
    const int l = line_of[p];
    // multiset: z (once), m1 (cnt_eq times, cnt_eq >= 1 here because this pixel's component exists), m2 = max_lt (0 = none)
    const int z = zero_area[l], m1 = max1[l], m2 = max_lt[l], c1 = cnt_eq[l];
    int thresh;
    if (z >= m1) thresh = m1;                      // z is the largest (or ties it): the next one down is m1
    else thresh = c1 >= 2 ? m1 : max(z, m2);       // m1 is the largest: second is m1 again, or the larger of z and m2
    const int r = uf_find(L, (int)p);
    if (st.area[r] < thresh && st.gain[r] > st.loss[r]) merged[p] = 255;
}

__global__ __launch_bounds__(256) void refine_scatter_kernel(int W, Geo g, const int *__restrict__ line_of, const uint8_t *__restrict__ merged,
                                                              uint8_t *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.P || !merged[p]) return;
    const int l = line_of[p];
    const MitRefineWindow w = g.win[l];
    const int cw = w.x2 - w.x1;
    const int local = (int)(p - g.pt_off[l]);
    const int y = local / cw, x = local - y * cw;
    out[(int64_t)(w.y1 + y) * W + w.x1 + x] = 255;  // bitwise_or of the lines' windows (:173): every writer stores the same byte
}

struct Layout {
    int64_t P;
    size_t win, pt_off, hist, cands, order, sums, zero_area, max1, cnt_eq, max_lt, line_of, grey, pred_bin, cand, merged, L, area, gain, loss, hv,
        total;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int make_layout(const MitRefineWindow *win, int n, Layout *L, std::vector<int64_t> *pt) {
    pt->assign(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (win[i].x2 <= win[i].x1 || win[i].y2 <= win[i].y1 || win[i].x1 < 0 || win[i].y1 < 0) return 1;
        (*pt)[i + 1] = (*pt)[i] + (int64_t)(win[i].x2 - win[i].x1) * (win[i].y2 - win[i].y1);
    }
    L->P = (*pt)[n];
    if (L->P >= (int64_t)1 << 31) return 2;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align_up(o + bytes);
        return at;
    };
    const size_t P = (size_t)L->P;
    L->win = take(sizeof(MitRefineWindow) * n);
    L->pt_off = take(8 * (size_t)(n + 1));
    L->hist = take(4 * 1024 * (size_t)n);
    L->cands = take(sizeof(MitRefineCand) * 6 * n);
    L->order = take(sizeof(MitRefineCand) * 4 * n);
    L->sums = take(8 * 6 * (size_t)n);
    L->zero_area = take(4 * (size_t)n);
    L->max1 = take(4 * (size_t)n);
    L->cnt_eq = take(4 * (size_t)n);
    L->max_lt = take(4 * (size_t)n);
    L->line_of = take(4 * P);
    L->grey = take(P);
    L->pred_bin = take(P);
    L->cand = take(P);
    L->merged = take(P);
    L->L = take(4 * P);
    L->area = take(4 * P);
    L->gain = take(4 * P);
    L->loss = take(4 * P);
    L->hv = take(4 * P);
    L->total = o;
    return 0;
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

int check_common(const char *fn, const void *page, const void *pred, int H, int W, const MitRefineWindow *win, int n, const void *ws,
                 int64_t ws_bytes, Layout *L, std::vector<int64_t> *pt) {
    if (!page || !pred || !win || !ws) return mit_set_error("%s: null pointer", fn);
    if (n <= 0 || H <= 0 || W <= 0) return mit_set_error("%s: bad arguments", fn);
    for (int i = 0; i < n; ++i)
        if (win[i].x1 < 0 || win[i].y1 < 0 || win[i].x2 > W || win[i].y2 > H || win[i].x2 <= win[i].x1 || win[i].y2 <= win[i].y1)
            return mit_set_error("%s: window %d (%d, %d)-(%d, %d) is empty or outside the %d x %d page", fn, i, win[i].x1, win[i].y1, win[i].x2,
                                 win[i].y2, W, H);
    if (make_layout(win, n, L, pt)) return mit_set_error("%s: too many window pixels for 32-bit labels (split the lines)", fn);
    if ((int64_t)L->total > ws_bytes) return mit_set_error("%s: workspace too small (%lld < %zu bytes)", fn, (long long)ws_bytes, L->total);
    if (reinterpret_cast<uintptr_t>(ws) & 255) return mit_set_error("%s: workspace must be 256-byte aligned", fn);
    return 0;
}

}  // namespace

extern "C" int64_t mit_ctd_refine_workspace_bytes(const MitRefineWindow *windows, int n) {
    if (!windows || n <= 0) return -1;
    Layout L;
    std::vector<int64_t> pt;
    if (make_layout(windows, n, &L, &pt)) return -1;
    return (int64_t)L.total;
}

extern "C" int mit_ctd_refine_hist(const uint8_t *page_dev, const uint8_t *pred_dev, int H, int W, const MitRefineWindow *windows, int n,
                                   int *hist_host, void *workspace_dev, int64_t workspace_bytes, void *stream) {
    Layout L;
    std::vector<int64_t> pt;
    if (check_common("mit_ctd_refine_hist", page_dev, pred_dev, H, W, windows, n, workspace_dev, workspace_bytes, &L, &pt)) return 1;
    if (!hist_host) return mit_set_error("mit_ctd_refine_hist: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_dev);
    MIT_CHECK_HIP(hipMemcpyAsync(ws + L.win, windows, sizeof(MitRefineWindow) * n, hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipMemcpyAsync(ws + L.pt_off, pt.data(), 8 * (size_t)(n + 1), hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));
    MIT_CHECK_HIP(hipMemsetAsync(ws + L.hist, 0, 4 * 1024 * (size_t)n, st));
    Geo g{reinterpret_cast<const MitRefineWindow *>(ws + L.win), reinterpret_cast<const int64_t *>(ws + L.pt_off), n, L.P};
    {
        MitProbeScope probe("ctd_refine_prepare", st, 6.0 * (double)L.P);
        hipLaunchKernelGGL(refine_prepare_kernel, dim3(blocks(L.P)), dim3(256), 0, st, page_dev, pred_dev, W, g, reinterpret_cast<int *>(ws + L.line_of),
                           reinterpret_cast<uint8_t *>(ws + L.grey), reinterpret_cast<uint8_t *>(ws + L.pred_bin), reinterpret_cast<int *>(ws + L.hist));
    }
    MIT_CHECK_LAUNCH("mit_ctd_refine_hist");
    MIT_CHECK_HIP(hipMemcpyAsync(hist_host, ws + L.hist, 4 * 1024 * (size_t)n, hipMemcpyDeviceToHost, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int mit_ctd_refine_scores(const uint8_t *page_dev, const uint8_t *pred_dev, int H, int W, const MitRefineWindow *windows, int n,
                                     const MitRefineCand *cands_host, uint64_t *sums_host, void *workspace_dev, int64_t workspace_bytes,
                                     void *stream) {
    Layout L;
    std::vector<int64_t> pt;
    if (check_common("mit_ctd_refine_scores", page_dev, pred_dev, H, W, windows, n, workspace_dev, workspace_bytes, &L, &pt)) return 1;
    if (!cands_host || !sums_host) return mit_set_error("mit_ctd_refine_scores: null pointer");
    for (int i = 0; i < 6 * n; ++i)
        if (cands_host[i].kind < 0 || cands_host[i].kind > 4) return mit_set_error("mit_ctd_refine_scores: candidate kind out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_dev);
    MIT_CHECK_HIP(hipMemcpyAsync(ws + L.cands, cands_host, sizeof(MitRefineCand) * 6 * n, hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));
    MIT_CHECK_HIP(hipMemsetAsync(ws + L.sums, 0, 8 * 6 * (size_t)n, st));
    Geo g{reinterpret_cast<const MitRefineWindow *>(ws + L.win), reinterpret_cast<const int64_t *>(ws + L.pt_off), n, L.P};
    {
        MitProbeScope probe("ctd_refine_scores", st, 5.0 * (double)L.P);
        hipLaunchKernelGGL(refine_score_kernel, dim3(blocks(L.P)), dim3(256), 0, st, page_dev, pred_dev, W, g, reinterpret_cast<const int *>(ws + L.line_of),
                           reinterpret_cast<const uint8_t *>(ws + L.grey), reinterpret_cast<const MitRefineCand *>(ws + L.cands),
                           reinterpret_cast<unsigned long long *>(ws + L.sums));
    }
    MIT_CHECK_LAUNCH("mit_ctd_refine_scores");
    MIT_CHECK_HIP(hipMemcpyAsync(sums_host, ws + L.sums, 8 * 6 * (size_t)n, hipMemcpyDeviceToHost, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int mit_ctd_refine_merge(const uint8_t *page_dev, const uint8_t *pred_dev, int H, int W, const MitRefineWindow *windows, int n,
                                    const MitRefineCand *order_host, uint8_t *out_dev, void *workspace_dev, int64_t workspace_bytes,
                                    void *stream) {
    Layout L;
    std::vector<int64_t> pt;
    if (check_common("mit_ctd_refine_merge", page_dev, pred_dev, H, W, windows, n, workspace_dev, workspace_bytes, &L, &pt)) return 1;
    if (!order_host || !out_dev) return mit_set_error("mit_ctd_refine_merge: null pointer");
    for (int i = 0; i < 4 * n; ++i)
        if (order_host[i].kind < 0 || order_host[i].kind > 4) return mit_set_error("mit_ctd_refine_merge: candidate kind out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_dev);
    MIT_CHECK_HIP(hipMemcpyAsync(ws + L.order, order_host, sizeof(MitRefineCand) * 4 * n, hipMemcpyHostToDevice, st));
    MIT_CHECK_HIP(hipStreamSynchronize(st));
    Geo g{reinterpret_cast<const MitRefineWindow *>(ws + L.win), reinterpret_cast<const int64_t *>(ws + L.pt_off), n, L.P};
    const int *line_of = reinterpret_cast<const int *>(ws + L.line_of);
    const uint8_t *grey = reinterpret_cast<const uint8_t *>(ws + L.grey), *pred_bin = reinterpret_cast<const uint8_t *>(ws + L.pred_bin);
    uint8_t *cand = reinterpret_cast<uint8_t *>(ws + L.cand), *merged = reinterpret_cast<uint8_t *>(ws + L.merged);
    int *lab = reinterpret_cast<int *>(ws + L.L);
    CompStats cs{reinterpret_cast<int *>(ws + L.area), reinterpret_cast<int *>(ws + L.gain), reinterpret_cast<int *>(ws + L.loss),
                 reinterpret_cast<int *>(ws + L.hv)};
    const MitRefineCand *order = reinterpret_cast<const MitRefineCand *>(ws + L.order);
    int *zero_area = reinterpret_cast<int *>(ws + L.zero_area), *max1 = reinterpret_cast<int *>(ws + L.max1),
        *cnt_eq = reinterpret_cast<int *>(ws + L.cnt_eq), *max_lt = reinterpret_cast<int *>(ws + L.max_lt);
    const unsigned nb = blocks(L.P);
    MIT_CHECK_HIP(hipMemsetAsync(merged, 0, (size_t)L.P, st));
    MIT_CHECK_HIP(hipMemsetAsync(ws + L.zero_area, 0, L.line_of - L.zero_area, st));  // zero_area, max1, cnt_eq, max_lt
    {
        // algorithmic bytes: five labelling rounds (4 candidates + holes), each reading the crop bytes and touching the 4-byte label
        // and the four per-pixel statistics words about twice
        MitProbeScope probe("ctd_refine_merge", st, 5.0 * (double)L.P * (4.0 + 2.0 * 20.0));
        for (int s = 0; s < 4; ++s) {
            hipLaunchKernelGGL(refine_label_init_kernel, dim3(nb), dim3(256), 0, st, page_dev, W, g, line_of, grey, order, s, 0, merged, cand, lab, cs,
                               zero_area);
            hipLaunchKernelGGL(refine_label_link_kernel, dim3(nb), dim3(256), 0, st, g, line_of, cand, lab);
            hipLaunchKernelGGL(refine_label_flatten_kernel, dim3(nb), dim3(256), 0, st, L.P, cand, lab);
            hipLaunchKernelGGL(refine_label_stats_kernel, dim3(nb), dim3(256), 0, st, g, line_of, cand, lab, merged, pred_bin, cs);
            hipLaunchKernelGGL(refine_apply_kernel, dim3(nb), dim3(256), 0, st, L.P, cand, lab, cs, merged);
        }
        hipLaunchKernelGGL(refine_label_init_kernel, dim3(nb), dim3(256), 0, st, page_dev, W, g, line_of, grey, order, 0, 1, merged, cand, lab, cs,
                           zero_area);
        hipLaunchKernelGGL(refine_label_link_kernel, dim3(nb), dim3(256), 0, st, g, line_of, cand, lab);
        hipLaunchKernelGGL(refine_label_flatten_kernel, dim3(nb), dim3(256), 0, st, L.P, cand, lab);
        hipLaunchKernelGGL(refine_label_stats_kernel, dim3(nb), dim3(256), 0, st, g, line_of, cand, lab, merged, pred_bin, cs);
        hipLaunchKernelGGL(refine_hole_max_kernel, dim3(nb), dim3(256), 0, st, L.P, line_of, cand, lab, cs, max1);
        hipLaunchKernelGGL(refine_hole_second_kernel, dim3(nb), dim3(256), 0, st, L.P, line_of, cand, lab, cs, max1, cnt_eq, max_lt);
        hipLaunchKernelGGL(refine_hole_apply_kernel, dim3(nb), dim3(256), 0, st, L.P, line_of, cand, lab, cs, zero_area, max1, cnt_eq, max_lt, merged);
        hipLaunchKernelGGL(refine_scatter_kernel, dim3(nb), dim3(256), 0, st, W, g, line_of, merged, out_dev);
    }
    MIT_CHECK_LAUNCH("mit_ctd_refine_merge");
    return 0;
}
