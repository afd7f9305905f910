// ctd_boxes.hip — SegDetectorRepresenter.boxes_from_bitmap on the GPU, all pages of a batch per launch chain.
//
// Reference: manga_translator/detection/ctd_utils/utils/db_utils.py:127-216 (boxes_from_bitmap, unclip, get_mini_boxes, box_score_fast)
// and detection/default_utils/dbnet_utils.py:97-190 — cv2.findContours(RETR_LIST) -> per contour cv2.minAreaRect / boxPoints ->
// cv2.fillPoly + cv2.mean score -> pyclipper JT_ROUND offset -> minAreaRect again -> scaled, rounded, clipped int64 corners.
// The host form of the same chain is csrc/hostglue.hip (mit_boxes_from_bitmap), pinned to the reference's Python by
// tests/golden/boxes.npz; this file must return what that routine returns (tests/test_ctd_boxes_gpu.py).
//
// What makes border following parallel.  Suzuki-Abe scans the image in raster order and starts a border where marks left by earlier
// traces allow it; the marks only decide WHERE a border is first met, the walk itself reads "is this pixel set".  With 8-connected
// foreground / 4-connected background every border separates one foreground component S from one background component B, and:
//   * the outer border of S is first met at S's first pixel in raster order (its left neighbour lies above S's top row's
//     neighbourhood, hence in the surrounding background);
//   * the border between S and a hole B (a background component that does not reach the image frame) is first met at the left
//     neighbour of B's first pixel (nothing but B's own border trace can have marked that pixel negative);
//   * no pixel is both, so the discovery order is the raster order of those start pixels.
// (Checked against the sequential algorithm on random bitmaps incl. dense noise: same contours, same start points, same order.)
// So: one union-find labelling of the padded bitmap (both classes in one array, links only within a class, root = smallest raster
// index = the component's first pixel), an ordered compaction of the start pixels per page, and ONE WAVE PER BORDER that walks it
// (lane 0) and does the geometry cooperatively:
//   hull            corner points only (a point between two equal steps is never a hull vertex) -> bitonic sort in LDS -> monotone chain
//   minAreaRect     one hull edge per lane, the first smallest area in edge order wins (as the sequential loop's strict <)
//   score           crossings of a traced contour are integers (unit steps): sorted (y, x) keys give every row's even-odd intervals;
//                   outline pixels outside them are added; pred is summed in double per lane, then over the wave
//   unclip          ClipperOffset round join on the 4-corner path, in double, on lane 0 (<= a few hundred points)
// Arithmetic is the host routine's, operation for operation (doubles; -ffp-contract=off), except the order of the score's double
// summation (a different rounding in the 53rd bit, which the float32 score does not see) and the device's libm in the round join.
// A contour longer than BFB_CAP points or with more than BFB_HCAP corner points does not fit the wave's LDS: its page is flagged and the
// caller runs the host routine for that page (same results by construction).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

constexpr int BFB_CAP = 8192;    // contour points per border held in LDS
constexpr int BFB_HCAP = 4096;   // corner points / hull vertices / offset-polygon points

struct BfbParams {
    int64_t pred_bs, bitmap_bs;   // elements between consecutive pages of pred / bitmap (H * W when dense)
    int B, H, W, Hp, Wp, dest_w, dest_h, max_cand;
    float thresh, unclip, min_sside, box_thresh, min_sside_out;
    int roll_start;
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
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// ---- 1. padded binary image + initial labels: every pixel points at the first pixel of its horizontal run inside the wave ----
__global__ __launch_bounds__(256) void bfb_init_kernel(BfbParams p, const float *__restrict__ pred, const uint8_t *__restrict__ bitmap,
                                                        uint8_t *__restrict__ F, int *__restrict__ L) {
    const int64_t PP = (int64_t)p.Hp * p.Wp;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = g < PP * p.B;
    int b = 0, q = 0, x = 0, y = 0;
    uint8_t f = 0;
    if (ok) {
        b = (int)(g / PP);
        q = (int)(g - (int64_t)b * PP);
        y = q / p.Wp;
        x = q - y * p.Wp;
        if (y >= 1 && y <= p.H && x >= 1 && x <= p.W) {
            const int64_t src = (int64_t)(y - 1) * p.W + (x - 1);
            f = bitmap ? (bitmap[b * p.bitmap_bs + src] != 0) : (pred[b * p.pred_bs + src] > p.thresh);
        }
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long ones = __ballot(ok && f);
    const bool prev_same = lane > 0 && x > 0 && ((((ones >> (lane - 1)) & 1ull) != 0) == (f != 0));
    const unsigned long long starts = __ballot(ok && !prev_same);
    if (!ok) return;
    F[g] = f;
    L[g] = q - (lane - (63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)))));
}

// ---- 2. links: foreground 8-connected, background 4-connected (only the unions that can join two sets) ----
__global__ __launch_bounds__(256) void bfb_link_kernel(BfbParams p, const uint8_t *__restrict__ F, int *__restrict__ L) {
    const int64_t PP = (int64_t)p.Hp * p.Wp;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= PP * p.B) return;
    const int b = (int)(g / PP);
    const int q = (int)(g - (int64_t)b * PP);
    const int y = q / p.Wp, x = q - y * p.Wp;
    const uint8_t *Fb = F + (int64_t)b * PP;
    int *Lb = L + (int64_t)b * PP;
    const uint8_t f = Fb[q];
    const bool left = x > 0 && Fb[q - 1] == f;
    if (left && (threadIdx.x & 63) == 0) uf_union(Lb, q, q - 1);
    if (y > 0) {
        const bool up = Fb[q - p.Wp] == f, ul = x > 0 && Fb[q - p.Wp - 1] == f;
        if (up && !(left && ul)) uf_union(Lb, q, q - p.Wp);
        if (f) {  // diagonals join foreground only
            const bool ur = x + 1 < p.Wp && Fb[q - p.Wp + 1] == f;
            if (ul && !up && !left) uf_union(Lb, q, q - p.Wp - 1);
            if (ur && !up && !(x + 1 < p.Wp && Fb[q + 1] == f)) uf_union(Lb, q, q - p.Wp + 1);
        }
    }
}

// start of a border at padded pixel q: 1 = outer border (q is a foreground root), 2 = hole border (q + 1 is a background root other
// than the frame's, which is pixel 0), 0 = none
__device__ __forceinline__ int start_kind(const BfbParams &p, const uint8_t *__restrict__ Fb, const int *__restrict__ Lb, int q, int x) {
    if (!Fb[q]) return 0;
    if (__atomic_load_n(&Lb[q], __ATOMIC_RELAXED) == q) return 1;
    if (x + 1 < p.Wp && !Fb[q + 1] && __atomic_load_n(&Lb[q + 1], __ATOMIC_RELAXED) == q + 1) return 2;
    return 0;
}

// ---- 3a. starts per row (one wave per padded row) ----
__global__ __launch_bounds__(64) void bfb_rowcount_kernel(BfbParams p, const uint8_t *__restrict__ F, const int *__restrict__ L, int *__restrict__ rowcnt) {
    const int y = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int64_t PP = (int64_t)p.Hp * p.Wp;
    const uint8_t *Fb = F + (int64_t)b * PP;
    const int *Lb = L + (int64_t)b * PP;
    int c = 0;
    for (int x = lane; x < p.Wp; x += 64) c += start_kind(p, Fb, Lb, y * p.Wp + x, x) != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) rowcnt[b * p.Hp + y] = c;
}
// ---- 3b. exclusive scan over the rows of a page (one wave per page) ----
__global__ __launch_bounds__(64) void bfb_rowscan_kernel(BfbParams p, const int *__restrict__ rowcnt, int *__restrict__ rowoff, int *__restrict__ total) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int per = (p.Hp + 63) / 64, y0 = lane * per, y1 = min(p.Hp, y0 + per);
    int s = 0;
    for (int y = y0; y < y1; ++y) s += rowcnt[b * p.Hp + y];
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    int run = incl - s;
    for (int y = y0; y < y1; ++y) {
        rowoff[b * p.Hp + y] = run;
        run += rowcnt[b * p.Hp + y];
    }
    if (lane == 63) total[b] = incl;
}
// ---- 3c. the last max_cand starts of a page, last found first (OpenCV's list order): slot = total - 1 - discovery index ----
__global__ __launch_bounds__(64) void bfb_starts_kernel(BfbParams p, const uint8_t *__restrict__ F, const int *__restrict__ L, const int *__restrict__ rowoff,
                                                         const int *__restrict__ total, int *__restrict__ starts) {
    const int y = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int64_t PP = (int64_t)p.Hp * p.Wp;
    const uint8_t *Fb = F + (int64_t)b * PP;
    const int *Lb = L + (int64_t)b * PP;
    int run = rowoff[b * p.Hp + y];
    const int tot = total[b];
    for (int x0 = 0; x0 < p.Wp; x0 += 64) {
        const int x = x0 + lane;
        const int k = x < p.Wp ? start_kind(p, Fb, Lb, y * p.Wp + x, x) : 0;
        const unsigned long long m = __ballot(k != 0);
        if (k) {
            const int idx = run + __popcll(m & ((1ull << lane) - 1ull));
            const int slot = tot - 1 - idx;
            if (slot < p.max_cand) starts[b * p.max_cand + slot] = (y * p.Wp + x) | (k == 2 ? (int)0x80000000 : 0);
        }
        run += __popcll(m);
    }
}

// ---- 4. one block of four waves per border ----
constexpr int NT = 256;  // threads per border: four waves share the sorts, the rectangle fit and the row sums; the walk itself is serial
__device__ __forceinline__ void block_sync() { __syncthreads(); }

// ascending bitonic sort of a[0 .. N) (N a power of two, LDS) by the block
__device__ void block_sort_u32(uint32_t *a, int N) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (N >> 1); t += NT) {  // one compare-exchange per thread and step: pair (i, i | j) with bit j of i clear
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint32_t u = a[i], v = a[l];
                if (((i & k) == 0) ? (u > v) : (u < v)) a[i] = v, a[l] = u;
            }
            block_sync();
        }
}
__device__ __forceinline__ int pow2_at_least(int n) {
    int N = 2;
    while (N < n) N <<= 1;
    return N;
}

// lane 0's small work arrays live in LDS: indexed by run-time values, they would otherwise be placed in scratch memory
struct Lane0Scratch {
    float box[4][2], tmp[4][2];
    int idx[4];
    long long px[4], py[4], q[4][2];
    double nx[4], ny[4];
    double red_area[NT / 64];
    int red_edge[NT / 64];
    double red_sum[NT / 64];
    long long red_cnt[NT / 64];
};

// Convex hull of the sorted keys key[0 .. n) (duplicates allowed; decode -> integer point): the monotone chain of hostglue.hip's
// convex_hull as its two independent scans — the lower chain (ascending, thread 0 of wave 0, stack growing up from hull[0]) and the upper
// chain (descending, thread 0 of wave 1, stack growing down from hull[cap - 1]) run side by side; cross products are exact in 64-bit
// integers, the pop rule (<= 0) is the host's, so the vertex list is the host's: lower + upper without its two end points, i.e.
// counter-clockwise from the smallest (x, y).  A duplicate of the stack's top pops it and takes its place (cross product 0): the
// explicit de-duplication of the host form is not needed.  The top two stack entries stay in registers.
// Whole block must call; returns the vertex count in *out_n (LDS), or -1 when the two stacks meet (caller: overflow).
template <typename Dec>
__device__ void hull_chains(const uint32_t *key, int n, uint32_t *hull, int cap, Dec dec, int *out_n, int *tmp2) {
    const int tid = threadIdx.x;
    auto crs = [&](uint32_t o, uint32_t a, uint32_t b) {
        int ox, oy, ax, ay, bx, by;
        dec(o, ox, oy), dec(a, ax, ay), dec(b, bx, by);
        return (int64_t)(ax - ox) * (by - oy) - (int64_t)(ay - oy) * (bx - ox);
    };
    if (tid == 0 || tid == 64) {
        const bool up = tid == 64;
        int k = 0;
        uint32_t t1 = 0, t2 = 0;  // stack[k - 1], stack[k - 2]
        auto at = [&](int j) -> uint32_t & { return hull[up ? cap - 1 - j : j]; };
        for (int s = 0; s < n; ++s) {
            const uint32_t pt = key[up ? n - 1 - s : s];
            while (k >= 2 && crs(t2, t1, pt) <= 0) {
                --k;
                t1 = t2;
                if (k >= 2) t2 = at(k - 2);
            }
            if (k < cap) at(k) = pt;
            ++k;
            t2 = t1, t1 = pt;
        }
        tmp2[up ? 1 : 0] = k;
    }
    block_sync();
    if (tid == 0) {
        const int kl = tmp2[0], ku = tmp2[1];
        int cnt;
        if (kl + ku > cap) cnt = -1;
        else {
            cnt = kl;
            for (int j = 1; j + 1 < ku; ++j) hull[cnt++] = hull[cap - 1 - j];
            if (cnt == 2 && hull[0] == hull[1]) cnt = 1;  // every point the same: the host's n == 1 case
        }
        *out_n = cnt;
    }
    block_sync();
}

// minAreaRect + boxPoints over hull[0 .. n) (hostglue.hip's min_area_rect after its convex_hull): the whole block; box / sside in LDS.
template <typename Dec>
__device__ void min_area_rect_block(const uint32_t *hull, int n, Dec dec, Lane0Scratch *w, float *sside) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float (*box)[2] = w->box;
    if (n == 0) {
        if (tid == 0) {
            for (int i = 0; i < 4; ++i) box[i][0] = box[i][1] = 0.f;
            *sside = 0.f;
        }
        block_sync();
        return;
    }
    if (n == 1) {
        if (tid == 0) {
            int x, y;
            dec(hull[0], x, y);
            for (int i = 0; i < 4; ++i) box[i][0] = (float)x, box[i][1] = (float)y;
            *sside = 0.f;
        }
        block_sync();
        return;
    }
    const int edges = n == 2 ? 1 : n;
    auto edge_fit = [&](int e, double &ux, double &uy, double &mnu, double &mxu, double &mnv, double &mxv) -> bool {
        int ax, ay, bx, by;
        dec(hull[e], ax, ay), dec(hull[(e + 1) % n], bx, by);
        ux = (double)bx - (double)ax, uy = (double)by - (double)ay;
        const double len = sqrt(ux * ux + uy * uy);
        if (len == 0) return false;
        ux /= len;
        uy /= len;
        mnu = 1e300, mxu = -1e300, mnv = 1e300, mxv = -1e300;
        for (int i = 0; i < n; ++i) {
            int qx, qy;
            dec(hull[i], qx, qy);
            const double u = (double)qx * ux + (double)qy * uy, v = -(double)qx * uy + (double)qy * ux;
            mnu = fmin(mnu, u), mxu = fmax(mxu, u), mnv = fmin(mnv, v), mxv = fmax(mxv, v);
        }
        return true;
    };
    double best = -1.0;
    int best_e = 0x7fffffff;
    for (int e = tid; e < edges; e += NT) {
        double ux, uy, a, b, c, d;
        if (!edge_fit(e, ux, uy, a, b, c, d)) continue;
        const double area = (b - a) * (d - c);
        if (best_e == 0x7fffffff || area < best) best = area, best_e = e;
    }
    auto better = [](double ob, int oe, double mb, int me) { return oe != 0x7fffffff && (me == 0x7fffffff || ob < mb || (ob == mb && oe < me)); };
    for (int o = 32; o > 0; o >>= 1) {  // the sequential loop keeps the FIRST edge of the smallest area
        const double ob = __shfl_xor(best, o);
        const int oe = __shfl_xor(best_e, o);
        if (better(ob, oe, best, best_e)) best = ob, best_e = oe;
    }
    if (lane == 0) w->red_area[wv] = best, w->red_edge[wv] = best_e;
    block_sync();
    if (tid == 0) {
        for (int k = 1; k < NT / 64; ++k)
            if (better(w->red_area[k], w->red_edge[k], best, best_e)) best = w->red_area[k], best_e = w->red_edge[k];
        double bux = 1, buy = 0, a = 0, b = 0, c = 0, d = 0;
        if (best_e != 0x7fffffff) edge_fit(best_e, bux, buy, a, b, c, d);
#define MIT_BFB_CORNER(i, u, v) box[i][0] = (float)((u) * bux - (v) * buy), box[i][1] = (float)((u) * buy + (v) * bux)
        MIT_BFB_CORNER(0, a, c);
        MIT_BFB_CORNER(1, b, c);
        MIT_BFB_CORNER(2, b, d);
        MIT_BFB_CORNER(3, a, d);
#undef MIT_BFB_CORNER
        *sside = (float)fmin(b - a, d - c);
    }
    block_sync();
}

// get_mini_boxes' order: stable sort by x, then [tl, tr, br, bl] (db_utils.py:175-196); thread 0, everything in LDS
__device__ void mini_box_order(Lane0Scratch *w) {
    float (*box)[2] = w->box, (*q)[2] = w->tmp;
    int *idx = w->idx;
    for (int i = 0; i < 4; ++i) idx[i] = i;
    for (int i = 1; i < 4; ++i)  // insertion sort = stable
        for (int j = i; j > 0 && box[idx[j]][0] < box[idx[j - 1]][0]; --j) {
            const int t = idx[j];
            idx[j] = idx[j - 1];
            idx[j - 1] = t;
        }
    for (int i = 0; i < 4; ++i) q[i][0] = box[idx[i]][0], q[i][1] = box[idx[i]][1];
    int i1, i4, i2, i3;
    if (q[1][1] > q[0][1]) i1 = 0, i4 = 1; else i1 = 1, i4 = 0;
    if (q[3][1] > q[2][1]) i2 = 2, i3 = 3; else i2 = 3, i3 = 2;
    idx[0] = i1, idx[1] = i2, idx[2] = i3, idx[3] = i4;
    for (int i = 0; i < 4; ++i) box[i][0] = q[idx[i]][0], box[i][1] = q[idx[i]][1];
}

__device__ __forceinline__ int64_t cround(double v) { return (int64_t)(v < 0 ? v - 0.5 : v + 0.5); }

// ClipperOffset, JT_ROUND, ET_CLOSEDPOLYGON on the 4-corner path (hostglue.hip's clipper_offset_round); thread 0.
// out: keys ((x + 32768) << 16 | (y + 32768)), returns the count (0: nothing, -1: does not fit).
__device__ int clipper_offset_round(Lane0Scratch *w, double delta, uint32_t *out, int cap) {
    float (*box)[2] = w->box;
    long long *px = w->px, *py = w->py;
    double *nx = w->nx, *ny = w->ny;
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        const int64_t x = (int64_t)box[i][0], y = (int64_t)box[i][1];
        if (n == 0 || px[n - 1] != x || py[n - 1] != y) px[n] = x, py[n] = y, ++n;
    }
    while (n > 1 && px[0] == px[n - 1] && py[0] == py[n - 1]) --n;
    if (n < 3 || delta <= 0) return 0;
    double area = 0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        area += (double)px[i] * py[j] - (double)px[j] * py[i];
    }
    if (area < 0)
        for (int i = 0; i < n / 2; ++i) {
            long long t = px[i];
            px[i] = px[n - 1 - i], px[n - 1 - i] = t;
            t = py[i];
            py[i] = py[n - 1 - i], py[n - 1 - i] = t;
        }
    const double two_pi = 6.283185307179586476925286766559, pi = 3.141592653589793238;
    const double def_arc_tolerance = 0.25;
    double yy = def_arc_tolerance;
    if (yy > fabs(delta) * def_arc_tolerance) yy = fabs(delta) * def_arc_tolerance;
    double steps = pi / acos(1 - yy / fabs(delta));
    if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
    const double m_sin = sin(two_pi / steps), m_cos = cos(two_pi / steps), steps_per_rad = steps / two_pi;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        const double ddx = (double)(px[j] - px[i]), ddy = (double)(py[j] - py[i]);
        const double f = 1.0 / sqrt(ddx * ddx + ddy * ddy);
        nx[i] = ddy * f, ny[i] = -ddx * f;
    }
    int cnt = 0;
    bool fits = true;
    auto push = [&](int64_t x, int64_t y) {
        if (cnt < cap && x > -32768 && x < 32768 && y > -32768 && y < 32768) out[cnt++] = ((uint32_t)(x + 32768) << 16) | (uint32_t)(y + 32768);
        else fits = false;
    };
    int k = n - 1;
    for (int j = 0; j < n; ++j) {
        double sinA = nx[k] * ny[j] - nx[j] * ny[k];
        auto add = [&](double ax, double ay) { push(cround((double)px[j] + ax * delta), cround((double)py[j] + ay * delta)); };
        bool done = false;
        if (fabs(sinA * delta) < 1.0) {
            const double cosA = nx[k] * nx[j] + ny[j] * ny[k];
            if (cosA > 0) {
                add(nx[k], ny[k]);
                done = true;
            }
        } else if (sinA > 1.0) sinA = 1.0;
        else if (sinA < -1.0) sinA = -1.0;
        if (!done) {
            if (sinA * delta < 0) {
                add(nx[k], ny[k]);
                push(px[j], py[j]);
                add(nx[j], ny[j]);
            } else {
                const double a = atan2(sinA, nx[k] * nx[j] + ny[k] * ny[j]);
                int st = (int)cround(steps_per_rad * fabs(a));
                if (st < 1) st = 1;
                double X = nx[k], Y = ny[k], X2;
                for (int i = 0; i < st; ++i) {
                    add(X, Y);
                    X2 = X;
                    X = X * m_cos - m_sin * Y;
                    Y = X2 * m_sin + Y * m_cos;
                }
                add(nx[j], ny[j]);
            }
        }
        k = j;
    }
    return fits ? cnt : -1;
}

// smallest index i in the sorted array a[0 .. n) with a[i] >= v
__device__ __forceinline__ int lower_bound_u32(const uint32_t *a, int n, uint32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// (development aid: mit_boxes_debug_stamps(dev buffer of 8 x max_candidates x B int64) makes every border record wall_clock64() at its
// phase boundaries — walk, rectangle, score, offset, end; scripts/bench_boxes.py --stamps prints the slowest border's phases)
__device__ long long *g_bfb_stamps = nullptr;
#define MIT_BFB_STAMP(k)                                                                                              \
    if (g_bfb_stamps && tid == 0) g_bfb_stamps[((int64_t)b * p.max_cand + slot) * 8 + (k)] = (long long)wall_clock64()

// the 8-neighbourhood in clockwise order from east (image coordinates, y down): E, SE, S, SW, W, NW, N, NE
__device__ __forceinline__ int dir_dx(int d) { return (int)((0x901Au >> (2 * d)) & 3u) - 1; }   // 1, 1, 0, -1, -1, -1, 0, 1
__device__ __forceinline__ int dir_dy(int d) { return (int)((0x01A9u >> (2 * d)) & 3u) - 1; }   // 0, 1, 1, 1, 0, -1, -1, -1

__global__ __launch_bounds__(NT) void bfb_border_kernel(BfbParams p, const float *__restrict__ pred, const uint8_t *__restrict__ F,
                                                        const int *__restrict__ starts, const int *__restrict__ total,
                                                        int64_t *__restrict__ boxes, float *__restrict__ scores, int *__restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) uint32_t bfb_lds[];
    uint32_t *pts = bfb_lds;            // [BFB_CAP] the contour in walk order (x | y << 16), later sorted as (y, x) keys
    uint32_t *aux = bfb_lds + BFB_CAP;  // [BFB_CAP] the walk's window of F, then corner keys + hull, crossing keys, the offset polygon + its hull
    __shared__ Lane0Scratch sh_w;
    __shared__ int sh_m, sh_go, sh_cnt;
    __shared__ float sh_sside;
    __shared__ double sh_score;

    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tot = total[b];
    if (slot >= tot || slot >= p.max_cand) return;
    const int64_t PP = (int64_t)p.Hp * p.Wp;
    const uint8_t *Fb = F + (int64_t)b * PP;
    const float *predb = pred + (int64_t)b * p.pred_bs;
    const int Wp = p.Wp;

    MIT_BFB_STAMP(0);
    // ---- the walk (hostglue.hip find_contours, steps 3.1 - 3.5, on the binary image) ----
    // Every thread runs the same walk on the same state (uniform control flow).  The pixels come from a 256 x 64 window of F kept in LDS
    // as bits, re-centred by the whole block (coalesced byte loads, one ballot per 64 pixels) when the walk comes within a pixel of its
    // edge; a step reads the three rows around the pixel at once (six 32-bit LDS words, one latency), forms the 8-neighbour mask in
    // registers and takes the next border pixel with one count-leading-zeros: the first set neighbour counter-clockwise from `from`.
    int n = 0;
    {
        uint32_t *tile = aux;  // [64 rows][9 words]: 256 columns + one spare word so that a 64-bit window never leaves the row
        constexpr int TP = 9;
        int tx0 = -(1 << 20), ty0 = -(1 << 20);
        auto tile_load = [&](const int cy, const int cx) {
            tx0 = (cx - 128) & ~31, ty0 = cy - 32;
            block_sync();  // nobody still reads the old window
            for (int r = wv; r < 64; r += NT / 64) {
                const int y = ty0 + r;
                uint8_t v[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int x = tx0 + 64 * jj + lane;
                    v[jj] = (y >= 0 && y < p.Hp && x >= 0 && x < Wp) ? Fb[y * Wp + x] : (uint8_t)0;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned long long mk = __ballot(v[jj] != 0);
                    if (lane == 0) tile[r * TP + 2 * jj] = (uint32_t)mk, tile[r * TP + 2 * jj + 1] = (uint32_t)(mk >> 32);
                }
                if (lane == 0) tile[r * TP + 8] = 0u;
            }
            block_sync();
        };
        // bits (x - 1, x, x + 1) of window row ry, as bits 0..2
        auto row3 = [&](const int ry, const int rx) -> uint32_t {
            const int k = (rx - 1) >> 5, sh = (rx - 1) & 31;
            const unsigned long long wd = (unsigned long long)tile[ry * TP + k] | ((unsigned long long)tile[ry * TP + k + 1] << 32);
            return (uint32_t)(wd >> sh) & 7u;
        };
        // the 8 neighbours of (cy, cx) as a mask in clockwise order from east: bit d = neighbour in direction d
        auto nbr_mask = [&](const int cy, const int cx) -> uint32_t {
            if (cy - 1 < ty0 || cy + 1 >= ty0 + 64 || cx - 1 < tx0 || cx + 1 >= tx0 + 256) tile_load(cy, cx);
            const int ry = cy - ty0, rx = cx - tx0;
            const uint32_t up = row3(ry - 1, rx), mid = row3(ry, rx), dn = row3(ry + 1, rx);
            // E = mid bit 2, SE = dn bit 2, S = dn bit 1, SW = dn bit 0, W = mid bit 0, NW = up bit 0, N = up bit 1, NE = up bit 2
            return ((mid >> 2) & 1u) | (((dn >> 2) & 1u) << 1) | (((dn >> 1) & 1u) << 2) | ((dn & 1u) << 3) | ((mid & 1u) << 4) | ((up & 1u) << 5) |
                   (((up >> 1) & 1u) << 6) | (((up >> 2) & 1u) << 7);
        };
        const int sv = starts[b * p.max_cand + slot];
        const int q0 = sv & 0x7fffffff;
        const bool hole = sv < 0;
        const int i = q0 / Wp, j = q0 - i * Wp;
        const int start = hole ? 0 : 4;  // direction of (i2, j2) seen from (i, j): east for a hole border, west for an outer one
        // (3.1) clockwise from `start`: the first set neighbour in the order start, start + 1, ...
        const uint32_t m0 = nbr_mask(i, j);
        int found = -1;
        if (m0) {
            const uint32_t rot = ((m0 | (m0 << 8)) >> start) & 0xffu;  // bit k = neighbour start + k
            found = (start + (__ffs((int)rot) - 1)) & 7;
        }
        if (found < 0) {
            if (tid == 0) pts[0] = (uint32_t)(j - 1) | ((uint32_t)(i - 1) << 16);
            n = 1;
        } else {
            const int i1 = i + dir_dy(found), j1 = j + dir_dx(found);
            int ci = i, cj = j, from = found;  // direction of the previous pixel seen from the current one
            for (;;) {
                const uint32_t m = nbr_mask(ci, cj);
                // (3.3) counter-clockwise from `from`: directions from - 1, from - 2, ..., from - 8 (= from): the highest set bit of the
                // mask rotated so that bit k stands for direction from + k (k = 7 is from - 1, k = 0 is from itself, examined last)
                const uint32_t rot = ((m | (m << 8)) >> from) & 0xffu;
                const int nd = (from + (31 - __clz((int)rot))) & 7;
                if (tid == 0 && n < BFB_CAP) pts[n] = (uint32_t)(cj - 1) | ((uint32_t)(ci - 1) << 16);
                ++n;
                const int ni = ci + dir_dy(nd), nj = cj + dir_dx(nd);
                if (ni == i && nj == j && ci == i1 && cj == j1) break;
                from = (nd + 4) & 7;  // the pixel we leave, seen from the one we enter
                ci = ni, cj = nj;
            }
        }
        n = __builtin_amdgcn_readfirstlane(n);
        if (tid == 0) sh_go = 1, sh_cnt = 0;
    }
    block_sync();
    if (n > BFB_CAP) {
        if (tid == 0) atomicOr(&overflow[b], 1);
        return;
    }
    MIT_BFB_STAMP(1);
    if (g_bfb_stamps && tid == 0) g_bfb_stamps[((int64_t)b * p.max_cand + slot) * 8 + 7] = n;
    auto dec_pt = [](uint32_t k, int &x, int &y) { x = (int)(k >> 16), y = (int)(k & 0xffffu); };  // (x << 16 | y) keys: sort by x, then y

    // ---- minAreaRect of the contour: corner points -> sort -> hull -> calipers ----
    {
        uint32_t *ck = aux, *hull = aux + BFB_HCAP;
        // corners: a point whose incoming and outgoing steps are equal lies inside a straight run (n >= 3 only; shorter contours keep
        // all); appended in any order — they are sorted next
        for (int i = tid; i < n; i += NT) {
            const uint32_t c = pts[i];
            const int cx = (int)(c & 0xffffu), cy = (int)(c >> 16);
            bool keep = true;
            if (n >= 3) {
                const uint32_t a = pts[i == 0 ? n - 1 : i - 1], d = pts[i + 1 == n ? 0 : i + 1];
                const int ax = (int)(a & 0xffffu), ay = (int)(a >> 16), dx = (int)(d & 0xffffu), dy = (int)(d >> 16);
                keep = !((cx - ax) == (dx - cx) && (cy - ay) == (dy - cy));
            }
            if (keep) {
                const int at = atomicAdd(&sh_cnt, 1);
                if (at < BFB_HCAP) ck[at] = ((uint32_t)cx << 16) | (uint32_t)cy;
            }
        }
        block_sync();
        const int m = sh_cnt;
        if (m > BFB_HCAP) {
            if (tid == 0) atomicOr(&overflow[b], 1);
            return;
        }
        const int N = pow2_at_least(m);
        for (int i = m + tid; i < N; i += NT) ck[i] = 0xffffffffu;
        block_sync();
        block_sort_u32(ck, N);
        hull_chains(ck, m, hull, BFB_CAP - BFB_HCAP, dec_pt, &sh_m, sh_w.idx);
        if (sh_m < 0) {
            if (tid == 0) atomicOr(&overflow[b], 1);
            return;
        }
        if (tid == 0) sh_cnt = 0;
        block_sync();
        min_area_rect_block(hull, sh_m, dec_pt, &sh_w, &sh_sside);
        if (tid == 0) {
            mini_box_order(&sh_w);
            if (sh_sside < p.min_sside) sh_go = 0;
        }
        block_sync();
        if (!sh_go) return;
    }

    MIT_BFB_STAMP(2);
    // ---- box_score_fast: mean of pred over the outline pixels and the even-odd interior ----
    {
        // crossing keys (y << 16 | x): an edge between consecutive points with different y crosses the row of its upper end, at that end's x
        for (int i = tid; i < n && n >= 2; i += NT) {
            const uint32_t a = pts[i], d = pts[i + 1 == n ? 0 : i + 1];
            const int ay = (int)(a >> 16), dy = (int)(d >> 16);
            if (ay != dy) aux[atomicAdd(&sh_cnt, 1)] = ay < dy ? (((uint32_t)ay << 16) | (a & 0xffffu)) : (((uint32_t)dy << 16) | (d & 0xffffu));
        }
        block_sync();
        const int m = sh_cnt;
        // (the contour's points x | y << 16 are (y, x) keys as they are)
        const int Nc = pow2_at_least(m > 0 ? m : 1), Np = pow2_at_least(n);
        for (int i = m + tid; i < Nc; i += NT) aux[i] = 0xffffffffu;
        for (int i = n + tid; i < Np; i += NT) pts[i] = 0xffffffffu;
        block_sync();
        if (m > 0) block_sort_u32(aux, Nc);
        block_sort_u32(pts, Np);
        const int ymin = (int)(pts[0] >> 16), ymax = (int)(pts[n - 1] >> 16);
        double sum = 0;
        long long cnt = 0;
        // a wave per row (rows interleaved over the waves), the pixels of a row spread over its lanes: a row's crossings [c0, c1) and
        // outline points [o0, o1) are runs of the two sorted arrays
        for (int y = ymin + wv; y <= ymax; y += NT / 64) {
            const uint32_t lo = (uint32_t)y << 16, hi = ((uint32_t)y + 1u) << 16;
            const int c0 = lower_bound_u32(aux, m, lo), c1 = lower_bound_u32(aux, m, hi);
            const int o0 = lower_bound_u32(pts, n, lo), o1 = lower_bound_u32(pts, n, hi);
            const float *row = predb + (int64_t)y * p.W;
            int prev_r = -1;  // right end of the intervals counted so far (touching intervals share a pixel)
            for (int c = c0; c + 1 < c1; c += 2) {
                const int xl = (int)(aux[c] & 0xffffu), xr = (int)(aux[c + 1] & 0xffffu);
                for (int x = (xl > prev_r ? xl : prev_r + 1) + lane; x <= xr; x += 64) sum += (double)row[x], ++cnt;
                if (xr > prev_r) prev_r = xr;
            }
            const int npair = (c1 - c0) & ~1;  // an unpaired last crossing (never for a closed contour) bounds nothing, as in the host loop
            for (int o = o0 + lane; o < o1; o += 64) {  // outline pixels outside every interval, each once
                const uint32_t key = pts[o];
                if (o > o0 && pts[o - 1] == key) continue;
                const int lt = lower_bound_u32(aux + c0, npair, key), le = lower_bound_u32(aux + c0, npair, key + 1u);
                if ((lt & 1) || le > lt) continue;  // inside [xs[2k], xs[2k+1]] for some k
                sum += (double)row[key & 0xffffu], ++cnt;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o);
            cnt += __shfl_xor(cnt, o);
        }
        if (lane == 0) sh_w.red_sum[wv] = sum, sh_w.red_cnt[wv] = cnt;
        block_sync();
        if (tid == 0) {
            for (int k = 1; k < NT / 64; ++k) sum += sh_w.red_sum[k], cnt += sh_w.red_cnt[k];
            sh_score = cnt ? sum / (double)cnt : 0.0;
            if ((double)p.box_thresh > sh_score) sh_go = 0;
        }
        block_sync();
        if (!sh_go) return;
    }

    MIT_BFB_STAMP(3);
    // ---- unclip + minAreaRect of the offset polygon + scaling (db_utils.py:147-170) ----
    auto dec_off = [](uint32_t k, int &x, int &y) { x = (int)(k >> 16) - 32768, y = (int)(k & 0xffffu) - 32768; };
    uint32_t *ek = aux, *ehull = aux + BFB_HCAP;
    if (tid == 0) {
        float (*box)[2] = sh_w.box;
        double area = 0, length = 0;
        for (int i = 0; i < 4; ++i) {
            const int j = (i + 1) & 3;
            area += (double)box[i][0] * box[j][1] - (double)box[j][0] * box[i][1];
            length += hypot((double)box[j][0] - box[i][0], (double)box[j][1] - box[i][1]);
        }
        area = fabs(area) * 0.5;
        int m = 0;
        if (length > 0) m = clipper_offset_round(&sh_w, area * (double)p.unclip / length, ek, BFB_HCAP);
        if (m < 0) atomicOr(&overflow[b], 1);
        sh_m = m;
        if (m <= 0) sh_go = 0;
    }
    block_sync();
    if (!sh_go) return;
    {
        const int m = sh_m, N = pow2_at_least(m);
        for (int i = m + tid; i < N; i += NT) ek[i] = 0xffffffffu;
        block_sync();
        block_sort_u32(ek, N);
        hull_chains(ek, m, ehull, BFB_CAP - BFB_HCAP, dec_off, &sh_m, sh_w.idx);
        if (sh_m < 0) {
            if (tid == 0) atomicOr(&overflow[b], 1);
            return;
        }
        min_area_rect_block(ehull, sh_m, dec_off, &sh_w, &sh_sside);
        if (tid == 0) {
            mini_box_order(&sh_w);
            float (*ebox)[2] = sh_w.box;
            const float esside = sh_sside;
            if (!(esside < p.min_sside_out)) {
                long long (*q)[2] = sh_w.q;
                for (int i = 0; i < 4; ++i) {
                    const float fx = nearbyintf(ebox[i][0] / (float)p.W * (float)p.dest_w), fy = nearbyintf(ebox[i][1] / (float)p.H * (float)p.dest_h);
                    q[i][0] = (long long)fminf(fmaxf(fx, 0.f), (float)p.dest_w);
                    q[i][1] = (long long)fminf(fmaxf(fy, 0.f), (float)p.dest_h);
                }
                int st = 0;
                if (p.roll_start) {
                    long long bestv = q[0][0] + q[0][1];
                    for (int i = 1; i < 4; ++i)
                        if (q[i][0] + q[i][1] < bestv) bestv = q[i][0] + q[i][1], st = i;
                }
                int64_t *bo = boxes + ((int64_t)b * p.max_cand + slot) * 8;
                for (int i = 0; i < 4; ++i) bo[2 * i] = q[(st + i) & 3][0], bo[2 * i + 1] = q[(st + i) & 3][1];
                scores[(int64_t)b * p.max_cand + slot] = (float)sh_score;
            }
        }
    }
    MIT_BFB_STAMP(4);
}

}  // namespace

extern "C" int mit_boxes_debug_stamps(void *stamps_dev) {
    long long *v = static_cast<long long *>(stamps_dev);
    MIT_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bfb_stamps), &v, sizeof(v)));
    return 0;
}

extern "C" int64_t mit_boxes_from_bitmap_dev_workspace_bytes(int B, int H, int W, int max_candidates) {
    if (B <= 0 || H <= 0 || W <= 0 || max_candidates <= 0) return 0;
    const int64_t PP = (int64_t)(H + 2) * (W + 2);
    int64_t n = 0;
    n += (PP * B + 255) / 256 * 256;                 // F
    n += PP * B * 4;                                 // L
    n += (int64_t)B * (H + 2) * 4 * 2;               // rowcnt, rowoff
    n += (int64_t)B * 4;                             // total
    n += (int64_t)B * max_candidates * 4;            // starts
    return n + 1024;
}

extern "C" int mit_boxes_from_bitmap_dev(const float *pred_dev, int64_t pred_bs, const uint8_t *bitmap_dev, int64_t bitmap_bs, float thresh, int B, int H, int W, int dest_w, int dest_h,
                                         int max_candidates, float unclip_ratio, float min_sside, float box_thresh, float min_sside_out, int roll_start,
                                         void *workspace_dev, int64_t workspace_bytes, int64_t *boxes_dev, float *scores_dev, int *counts_dev,
                                         int *overflow_dev, void *stream) {
    if (!pred_dev || !workspace_dev || !boxes_dev || !scores_dev || !counts_dev || !overflow_dev) return mit_set_error("mit_boxes_from_bitmap_dev: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || max_candidates <= 0) return mit_set_error("mit_boxes_from_bitmap_dev: bad size");
    if (H > 32000 || W > 32000 || (int64_t)(H + 2) * (W + 2) > 0x7fffffffLL) return mit_set_error("mit_boxes_from_bitmap_dev: map too large (16-bit point coordinates)");
    if (B > 65535 || max_candidates > 65535) return mit_set_error("mit_boxes_from_bitmap_dev: too many pages / candidates for one launch");
    if (workspace_bytes < mit_boxes_from_bitmap_dev_workspace_bytes(B, H, W, max_candidates)) return mit_set_error("mit_boxes_from_bitmap_dev: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    BfbParams p;
    p.pred_bs = pred_bs > 0 ? pred_bs : (int64_t)H * W, p.bitmap_bs = bitmap_bs > 0 ? bitmap_bs : (int64_t)H * W;
    p.B = B, p.H = H, p.W = W, p.Hp = H + 2, p.Wp = W + 2, p.dest_w = dest_w, p.dest_h = dest_h, p.max_cand = max_candidates;
    p.thresh = thresh, p.unclip = unclip_ratio, p.min_sside = min_sside, p.box_thresh = box_thresh, p.min_sside_out = min_sside_out, p.roll_start = roll_start;
    const int64_t PP = (int64_t)p.Hp * p.Wp;
    char *w = static_cast<char *>(workspace_dev);
    uint8_t *F = reinterpret_cast<uint8_t *>(w);
    w += (PP * B + 255) / 256 * 256;
    int *L = reinterpret_cast<int *>(w);
    w += PP * B * 4;
    int *rowcnt = reinterpret_cast<int *>(w);
    w += (int64_t)B * p.Hp * 4;
    int *rowoff = reinterpret_cast<int *>(w);
    w += (int64_t)B * p.Hp * 4;
    int *starts = reinterpret_cast<int *>(w);
    w += (int64_t)B * max_candidates * 4;
    MIT_CHECK_HIP(hipMemsetAsync(boxes_dev, 0, sizeof(int64_t) * 8 * (size_t)B * max_candidates, s));
    MIT_CHECK_HIP(hipMemsetAsync(scores_dev, 0, sizeof(float) * (size_t)B * max_candidates, s));
    MIT_CHECK_HIP(hipMemsetAsync(overflow_dev, 0, sizeof(int) * (size_t)B, s));
    const int64_t total_px = PP * B;
    const unsigned blocks = (unsigned)((total_px + 255) / 256);
    hipLaunchKernelGGL(bfb_init_kernel, dim3(blocks), dim3(256), 0, s, p, pred_dev, bitmap_dev, F, L);
    hipLaunchKernelGGL(bfb_link_kernel, dim3(blocks), dim3(256), 0, s, p, F, L);
    hipLaunchKernelGGL(bfb_rowcount_kernel, dim3(p.Hp, B), dim3(64), 0, s, p, F, L, rowcnt);
    hipLaunchKernelGGL(bfb_rowscan_kernel, dim3(B), dim3(64), 0, s, p, rowcnt, rowoff, counts_dev);
    hipLaunchKernelGGL(bfb_starts_kernel, dim3(p.Hp, B), dim3(64), 0, s, p, F, L, rowoff, counts_dev, starts);
    static DynSmemOptIn optin;
    const size_t lds = (size_t)2 * BFB_CAP * sizeof(uint32_t);
    optin.ensure(reinterpret_cast<const void *>(bfb_border_kernel), lds + 1024);   // 64 KB of dynamic LDS beside the static variables: above the default limit
    hipLaunchKernelGGL(bfb_border_kernel, dim3(max_candidates, B), dim3(NT), lds, s, p, pred_dev, F, starts, counts_dev, boxes_dev, scores_dev, overflow_dev);
    MIT_CHECK_LAUNCH("mit_boxes_from_bitmap_dev");
    return 0;
}
