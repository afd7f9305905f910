// hostglue.hip — host-side (CPU, no kernels) geometry the detectors need between the dense network and the text lines:
// threshold bitmap -> contours -> min-area rectangles -> score -> "unclip" (polygon offset) -> boxes.
//
// Native restatement of what the reference does through OpenCV + pyclipper + shapely in
//   SegDetectorRepresenter.boxes_from_bitmap / get_mini_boxes / box_score_fast / unclip
//   (manga_translator/detection/ctd_utils/utils/db_utils.py:127-216 and default_utils/dbnet_utils.py:97-190).
// None of those libraries is available where this repo is built or tested, so each primitive follows the published
// algorithm of the library routine the reference calls (PARITY UNPINNED against the real libraries):
//   cv2.findContours(RETR_LIST, CHAIN_APPROX_SIMPLE)  Suzuki-Abe border following (outer + hole borders); contours are
//                                                     returned last-found-first like OpenCV's list mode; all border
//                                                     points are kept (CHAIN_APPROX_SIMPLE only drops collinear points,
//                                                     which changes neither the hull, the filled region nor min/max)
//   cv2.minAreaRect + cv2.boxPoints                   convex hull + rotating calipers over hull edges
//   cv2.fillPoly + cv2.mean(mask)                     even-odd scanline fill at pixel centres + the outline pixels
//   pyclipper offset, JT_ROUND, ET_CLOSEDPOLYGON      ClipperOffset::DoOffset / OffsetPoint / DoRound (arc tolerance 0.25),
//                                                     integer coordinates (pyclipper truncates the float box)
//   shapely Polygon.area / .length                    shoelace area, perimeter
// It lives in the C-ABI library so the plugins need no Python-level OpenCV stand-in on the per-page path.
// Further down: the host half of the mask refinement (mit_mask_assign_lines / mit_mask_line_crops — connected components as row runs
// and their assignment to text lines, mask_refinement/text_mask_utils.py:100-170).

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

struct Pt { int x, y; };
struct Pd { double x, y; };

// ---- Suzuki-Abe border following on a 0/1 image with a one-pixel zero frame ----
void find_contours(const uint8_t *bitmap, int H, int W, std::vector<std::vector<Pt>> &out) {
    const int Wp = W + 2, Hp = H + 2;
    std::vector<int> f((size_t)Wp * Hp, 0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) f[(size_t)(y + 1) * Wp + x + 1] = bitmap[(size_t)y * W + x] ? 1 : 0;
    // 8-neighbourhood in clockwise order starting east (image coordinates, y down): E, SE, S, SW, W, NW, N, NE
    static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int dy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    auto dir_of = [&](int fx, int fy, int tx, int ty) {
        for (int d = 0; d < 8; ++d)
            if (fx + dx[d] == tx && fy + dy[d] == ty) return d;
        return -1;
    };
    int nbd = 1;
    for (int i = 1; i <= H; ++i) {
        for (int j = 1; j <= W; ++j) {
            const int v = f[(size_t)i * Wp + j];
            if (v == 0) continue;
            int i2, j2;
            if (v == 1 && f[(size_t)i * Wp + j - 1] == 0) {  // outer border start
                i2 = i;
                j2 = j - 1;
            } else if (v >= 1 && f[(size_t)i * Wp + j + 1] == 0) {  // hole border start
                i2 = i;
                j2 = j + 1;
            } else {
                continue;
            }
            ++nbd;
            std::vector<Pt> contour;
            // (3.1) clockwise search around (i, j) starting from (i2, j2)
            int start = dir_of(j, i, j2, i2);
            int found = -1;
            for (int s = 0; s < 8; ++s) {
                const int d = (start + s) & 7;
                if (f[(size_t)(i + dy[d]) * Wp + j + dx[d]] != 0) {
                    found = d;
                    break;
                }
            }
            if (found < 0) {  // isolated pixel
                f[(size_t)i * Wp + j] = -nbd;
                contour.push_back({j - 1, i - 1});
                out.push_back(contour);
                continue;
            }
            const int i1 = i + dy[found], j1 = j + dx[found];
            int pi = i1, pj = j1;  // (i2, j2) := (i1, j1)
            int ci = i, cj = j;    // (i3, j3) := (i, j)
            for (;;) {
                // (3.3) counter-clockwise search around (ci, cj) starting after (pi, pj)
                const int from = dir_of(cj, ci, pj, pi);
                int nd = -1;
                bool east_zero_examined = false;
                for (int s = 1; s <= 8; ++s) {
                    const int d = (from - s + 16) & 7;  // counter-clockwise = decreasing index in the clockwise table
                    if (f[(size_t)(ci + dy[d]) * Wp + cj + dx[d]] != 0) {
                        nd = d;
                        break;
                    }
                    if (d == 0) east_zero_examined = true;  // pixel (ci, cj + 1) was examined and is 0
                }
                contour.push_back({cj - 1, ci - 1});
                // (3.4)
                int &cur = f[(size_t)ci * Wp + cj];
                if (east_zero_examined) cur = -nbd;
                else if (cur == 1) cur = nbd;
                const int ni = ci + dy[nd], nj = cj + dx[nd];
                // (3.5)
                if (ni == i && nj == j && ci == i1 && cj == j1) break;
                pi = ci;
                pj = cj;
                ci = ni;
                cj = nj;
            }
            out.push_back(contour);
        }
    }
    std::reverse(out.begin(), out.end());  // OpenCV's list mode hands back the last-found contour first
}

// ---- convex hull (monotone chain) ----
double cross(const Pd &o, const Pd &a, const Pd &b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

std::vector<Pd> convex_hull(std::vector<Pd> p) {
    std::sort(p.begin(), p.end(), [](const Pd &a, const Pd &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const Pd &a, const Pd &b) { return a.x == b.x && a.y == b.y; }), p.end());
    const int n = (int)p.size();
    if (n < 3) return p;
    std::vector<Pd> h(2 * n);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k;
        h[k++] = p[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k;
        h[k++] = p[i];
    }
    h.resize(k - 1);
    return h;
}

// ---- minAreaRect + boxPoints: 4 corners (float32 like cv2.boxPoints) and the short side ----
void min_area_rect(const std::vector<Pd> &pts, float box[4][2], float *sside) {
    std::vector<Pd> h = convex_hull(pts);
    const int n = (int)h.size();
    if (n == 0) {
        memset(box, 0, sizeof(float) * 8);
        *sside = 0;
        return;
    }
    if (n == 1) {
        for (int i = 0; i < 4; ++i) box[i][0] = (float)h[0].x, box[i][1] = (float)h[0].y;
        *sside = 0;
        return;
    }
    double best = -1, bux = 1, buy = 0, bmin_u = 0, bmax_u = 0, bmin_v = 0, bmax_v = 0;
    const int edges = n == 2 ? 1 : n;
    for (int e = 0; e < edges; ++e) {
        const Pd &a = h[e], &b = h[(e + 1) % n];
        double ux = b.x - a.x, uy = b.y - a.y;
        const double len = sqrt(ux * ux + uy * uy);
        if (len == 0) continue;
        ux /= len;
        uy /= len;
        double mnu = 1e300, mxu = -1e300, mnv = 1e300, mxv = -1e300;
        for (const Pd &q : h) {
            const double u = q.x * ux + q.y * uy, v = -q.x * uy + q.y * ux;
            mnu = std::min(mnu, u);
            mxu = std::max(mxu, u);
            mnv = std::min(mnv, v);
            mxv = std::max(mxv, v);
        }
        const double area = (mxu - mnu) * (mxv - mnv);
        if (best < 0 || area < best) {
            best = area;
            bux = ux;
            buy = uy;
            bmin_u = mnu, bmax_u = mxu, bmin_v = mnv, bmax_v = mxv;
        }
    }
    const double us[4] = {bmin_u, bmax_u, bmax_u, bmin_u}, vs[4] = {bmin_v, bmin_v, bmax_v, bmax_v};
    for (int i = 0; i < 4; ++i) {
        box[i][0] = (float)(us[i] * bux - vs[i] * buy);
        box[i][1] = (float)(us[i] * buy + vs[i] * bux);
    }
    *sside = (float)std::min(bmax_u - bmin_u, bmax_v - bmin_v);
}

// get_mini_boxes' point order (db_utils.py:175-196): sort by x, then [tl, tr, br, bl]
void mini_box_order(float box[4][2]) {
    int idx[4] = {0, 1, 2, 3};
    std::stable_sort(idx, idx + 4, [&](int a, int b) { return box[a][0] < box[b][0]; });
    float p[4][2];
    for (int i = 0; i < 4; ++i) p[i][0] = box[idx[i]][0], p[i][1] = box[idx[i]][1];
    int i1, i4, i2, i3;
    if (p[1][1] > p[0][1]) i1 = 0, i4 = 1; else i1 = 1, i4 = 0;
    if (p[3][1] > p[2][1]) i2 = 2, i3 = 3; else i2 = 3, i3 = 2;
    const int order[4] = {i1, i2, i3, i4};
    float r[4][2];
    for (int i = 0; i < 4; ++i) r[i][0] = p[order[i]][0], r[i][1] = p[order[i]][1];
    memcpy(box, r, sizeof(r));
}

// ---- box_score_fast: mean of pred over fillPoly(contour) (outline pixels + even-odd interior) ----
double polygon_mean(const float *pred, int H, int W, const std::vector<Pt> &poly) {
    int xmin = INT32_MAX, xmax = INT32_MIN, ymin = INT32_MAX, ymax = INT32_MIN;
    for (const Pt &p : poly) {
        xmin = std::min(xmin, p.x), xmax = std::max(xmax, p.x);
        ymin = std::min(ymin, p.y), ymax = std::max(ymax, p.y);
    }
    xmin = std::max(0, std::min(xmin, W - 1)), xmax = std::max(0, std::min(xmax, W - 1));
    ymin = std::max(0, std::min(ymin, H - 1)), ymax = std::max(0, std::min(ymax, H - 1));
    const int mw = xmax - xmin + 1, mh = ymax - ymin + 1;
    std::vector<uint8_t> mask((size_t)mw * mh, 0);
    const int n = (int)poly.size();
    auto set = [&](int x, int y) {
        x -= xmin, y -= ymin;
        if (x >= 0 && x < mw && y >= 0 && y < mh) mask[(size_t)y * mw + x] = 1;
    };
    // outline: straight pixel runs between consecutive vertices (contour steps are 8-connected unit moves or their runs)
    for (int i = 0; i < n; ++i) {
        Pt a = poly[i], b = poly[(i + 1) % n];
        int sx = (b.x > a.x) - (b.x < a.x), sy = (b.y > a.y) - (b.y < a.y);
        int adx = abs(b.x - a.x), ady = abs(b.y - a.y);
        if (adx == ady || adx == 0 || ady == 0) {
            int steps = std::max(adx, ady);
            for (int s = 0; s <= steps; ++s) set(a.x + s * sx, a.y + s * sy);
        } else {  // general segment (offset polygons are never scored, but keep the routine total): Bresenham
            int x = a.x, y = a.y, err = adx - ady;
            for (;;) {
                set(x, y);
                if (x == b.x && y == b.y) break;
                const int e2 = 2 * err;
                if (e2 > -ady) err -= ady, x += sx;
                if (e2 < adx) err += adx, y += sy;
            }
        }
    }
    // interior: even-odd rule sampled at pixel centres.  An edge (a, b) crosses the scan line y when min(a.y, b.y) <= y < max(a.y, b.y)
    // (integer coordinates): every edge is filed under the rows it crosses once — a traced contour's edges are unit steps, so that is
    // O(edges) instead of O(rows x edges) — and a row's crossings are sorted as before (their order of discovery never mattered).
    std::vector<std::vector<double>> rows((size_t)mh);
    for (int i = 0; i < n; ++i) {
        const Pt a = poly[i], b = poly[(i + 1) % n];
        if (a.y == b.y) continue;
        const int y0 = std::max(std::min(a.y, b.y), ymin), y1 = std::min(std::max(a.y, b.y), ymax + 1);   // rows [y0, y1)
        for (int y = y0; y < y1; ++y) rows[(size_t)(y - ymin)].push_back(a.x + ((double)y - a.y) * (double)(b.x - a.x) / (double)(b.y - a.y));
    }
    for (int y = ymin; y <= ymax; ++y) {
        std::vector<double> &xs = rows[(size_t)(y - ymin)];
        std::sort(xs.begin(), xs.end());
        for (size_t k = 0; k + 1 < xs.size(); k += 2)
            for (int x = (int)ceil(xs[k]); x <= (int)floor(xs[k + 1]); ++x) set(x, y);
    }
    double sum = 0;
    int64_t cnt = 0;
    for (int y = 0; y < mh; ++y)
        for (int x = 0; x < mw; ++x)
            if (mask[(size_t)y * mw + x]) sum += pred[(size_t)(y + ymin) * W + x + xmin], ++cnt;
    return cnt ? sum / (double)cnt : 0.0;
}

// ---- pyclipper offset of a closed polygon, JT_ROUND (ClipperOffset::DoOffset / OffsetPoint / DoRound) ----
int64_t cround(double v) { return (int64_t)(v < 0 ? v - 0.5 : v + 0.5); }

void clipper_offset_round(const float box[4][2], double delta, std::vector<Pd> &out) {
    struct IP { int64_t x, y; };
    std::vector<IP> path;
    for (int i = 0; i < 4; ++i) {
        IP p = {(int64_t)box[i][0], (int64_t)box[i][1]};  // pyclipper truncates float coordinates to integers
        if (path.empty() || path.back().x != p.x || path.back().y != p.y) path.push_back(p);
    }
    while (path.size() > 1 && path.front().x == path.back().x && path.front().y == path.back().y) path.pop_back();
    const int n = (int)path.size();
    out.clear();
    if (n < 3 || delta <= 0) return;
    double area = 0;
    for (int i = 0; i < n; ++i) {
        const IP &a = path[i], &b = path[(i + 1) % n];
        area += (double)a.x * b.y - (double)b.x * a.y;
    }
    if (area < 0) std::reverse(path.begin(), path.end());  // FixOrientations
    const double two_pi = 6.283185307179586476925286766559, pi = 3.141592653589793238;
    const double def_arc_tolerance = 0.25;
    double y = def_arc_tolerance;  // ArcTolerance = 0.25 (pyclipper default)
    if (y > fabs(delta) * def_arc_tolerance) y = fabs(delta) * def_arc_tolerance;
    double steps = pi / acos(1 - y / fabs(delta));
    if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
    const double m_sin = sin(two_pi / steps), m_cos = cos(two_pi / steps), steps_per_rad = steps / two_pi;
    std::vector<Pd> normals(n);
    for (int i = 0; i < n; ++i) {
        const IP &a = path[i], &b = path[(i + 1) % n];
        double ddx = (double)(b.x - a.x), ddy = (double)(b.y - a.y);
        const double f = 1.0 / sqrt(ddx * ddx + ddy * ddy);
        normals[i] = {ddy * f, -ddx * f};
    }
    int k = n - 1;
    for (int j = 0; j < n; ++j) {
        double sinA = normals[k].x * normals[j].y - normals[j].x * normals[k].y;
        const IP &pt = path[j];
        auto add = [&](double nx, double ny) { out.push_back({(double)cround(pt.x + nx * delta), (double)cround(pt.y + ny * delta)}); };
        if (fabs(sinA * delta) < 1.0) {
            const double cosA = normals[k].x * normals[j].x + normals[j].y * normals[k].y;
            if (cosA > 0) {
                add(normals[k].x, normals[k].y);
                k = j;
                continue;
            }
        } else if (sinA > 1.0) sinA = 1.0;
        else if (sinA < -1.0) sinA = -1.0;
        if (sinA * delta < 0) {
            add(normals[k].x, normals[k].y);
            out.push_back({(double)pt.x, (double)pt.y});
            add(normals[j].x, normals[j].y);
        } else {  // DoRound
            const double a = atan2(sinA, normals[k].x * normals[j].x + normals[k].y * normals[j].y);
            const int st = std::max((int)cround(steps_per_rad * fabs(a)), 1);
            double X = normals[k].x, Y = normals[k].y, X2;
            for (int i = 0; i < st; ++i) {
                add(X, Y);
                X2 = X;
                X = X * m_cos - m_sin * Y;
                Y = X2 * m_sin + Y * m_cos;
            }
            add(normals[j].x, normals[j].y);
        }
        k = j;
    }
}

}  // namespace

extern "C" int mit_boxes_from_bitmap(const float *pred, const uint8_t *bitmap, int H, int W, int dest_w, int dest_h, int max_candidates,
                                     float unclip_ratio, float min_sside, float box_thresh, float min_sside_out, int roll_start,
                                     int64_t *boxes_out, float *scores_out, int *n_out) {
    if (!pred || !bitmap || !boxes_out || !scores_out || !n_out) return mit_set_error("mit_boxes_from_bitmap: null pointer");
    if (H <= 0 || W <= 0 || max_candidates <= 0) return mit_set_error("mit_boxes_from_bitmap: bad size");
    std::vector<std::vector<Pt>> contours;
    find_contours(bitmap, H, W, contours);
    const int n = (int)std::min<size_t>(contours.size(), (size_t)max_candidates);
    memset(boxes_out, 0, sizeof(int64_t) * 8 * (size_t)n);
    memset(scores_out, 0, sizeof(float) * (size_t)n);
    // every contour is independent (own output slot, read-only inputs): a page's 20-40 text-line contours are spread over a few
    // threads (rectangle fit + polygon mean + Clipper offset cost ~0.1 ms each); results do not depend on the thread count
    auto one = [&](int idx) {
        const std::vector<Pt> &c = contours[idx];
        std::vector<Pd> pts(c.size());
        for (size_t i = 0; i < c.size(); ++i) pts[i] = {(double)c[i].x, (double)c[i].y};
        float box[4][2], sside;
        min_area_rect(pts, box, &sside);
        mini_box_order(box);
        if (sside < min_sside) return;
        const double score = polygon_mean(pred, H, W, c);
        if (box_thresh > score) return;
        // unclip: shapely area / length of the float box, Clipper round offset, min-area rectangle of the result
        double area = 0, length = 0;
        for (int i = 0; i < 4; ++i) {
            const int j = (i + 1) & 3;
            area += (double)box[i][0] * box[j][1] - (double)box[j][0] * box[i][1];
            length += hypot((double)box[j][0] - box[i][0], (double)box[j][1] - box[i][1]);
        }
        area = fabs(area) * 0.5;
        if (length <= 0) return;
        std::vector<Pd> expanded;
        clipper_offset_round(box, area * unclip_ratio / length, expanded);
        if (expanded.empty()) return;
        float ebox[4][2], esside;
        min_area_rect(expanded, ebox, &esside);
        mini_box_order(ebox);
        if (esside < min_sside_out) return;
        int64_t q[4][2];
        for (int i = 0; i < 4; ++i) {
            const float fx = nearbyintf(ebox[i][0] / (float)W * (float)dest_w), fy = nearbyintf(ebox[i][1] / (float)H * (float)dest_h);
            q[i][0] = (int64_t)std::min(std::max(fx, 0.f), (float)dest_w);
            q[i][1] = (int64_t)std::min(std::max(fy, 0.f), (float)dest_h);
        }
        int start = 0;
        if (roll_start) {  // dbnet_utils.py:139-140: start from the corner with the smallest x + y
            int64_t best = q[0][0] + q[0][1];
            for (int i = 1; i < 4; ++i)
                if (q[i][0] + q[i][1] < best) best = q[i][0] + q[i][1], start = i;
        }
        for (int i = 0; i < 4; ++i) {
            boxes_out[(size_t)idx * 8 + 2 * i] = q[(start + i) & 3][0];
            boxes_out[(size_t)idx * 8 + 2 * i + 1] = q[(start + i) & 3][1];
        }
        scores_out[idx] = (float)score;
    };
    static const int max_threads = [] {
        const char *v = getenv("MIT_HOST_THREADS");
        int t = v && *v ? atoi(v) : 8;
        return t < 1 ? 1 : (t > 64 ? 64 : t);
    }();
    const int nthreads = n >= 8 ? std::min(max_threads, n / 4) : 1;
    if (nthreads <= 1) {
        for (int idx = 0; idx < n; ++idx) one(idx);
    } else {
        std::vector<std::thread> pool;
        std::atomic<int> next{0};
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&] {
                for (int idx = next.fetch_add(1); idx < n; idx = next.fetch_add(1)) one(idx);
            });
        for (auto &th : pool) th.join();
    }
    *n_out = n;
    return 0;
}

// ---- distance between two quadrilaterals (textline.polygon_distance, operation for operation) -------------------------------------
namespace {
struct P2 {
    double x, y;
};
inline double pd_orient(const P2 &p, const P2 &q, const P2 &r) { return (q.x - p.x) * (r.y - p.y) - (q.y - p.y) * (r.x - p.x); }
inline bool pd_on(const P2 &p, const P2 &q, const P2 &r) {
    return std::min(p.x, q.x) <= r.x && r.x <= std::max(p.x, q.x) && std::min(p.y, q.y) <= r.y && r.y <= std::max(p.y, q.y);
}
inline bool pd_segs_intersect(const P2 &a, const P2 &b, const P2 &c, const P2 &d) {
    const double o1 = pd_orient(a, b, c), o2 = pd_orient(a, b, d), o3 = pd_orient(c, d, a), o4 = pd_orient(c, d, b);
    if (((o1 > 0) != (o2 > 0)) && ((o3 > 0) != (o4 > 0)) && o1 * o2 != 0 && o3 * o4 != 0) return true;
    return (o1 == 0 && pd_on(a, b, c)) || (o2 == 0 && pd_on(a, b, d)) || (o3 == 0 && pd_on(c, d, a)) || (o4 == 0 && pd_on(c, d, b));
}
inline bool pd_point_in_quad(const P2 &p, const P2 *poly) {
    bool inside = false;
    for (int i = 0; i < 4; ++i) {
        const P2 &a = poly[i], &b = poly[(i + 1) & 3];
        if ((a.y > p.y) != (b.y > p.y) && p.x < (b.x - a.x) * (p.y - a.y) / (b.y - a.y) + a.x) inside = !inside;
    }
    return inside;
}
inline double pd_seg_point(const P2 &p, const P2 &a, const P2 &b) {
    const double abx = b.x - a.x, aby = b.y - a.y, apx = p.x - a.x, apy = p.y - a.y;
    const double den = abx * abx + aby * aby;
    const double t = den == 0 ? 0.0 : std::min(1.0, std::max(0.0, (apx * abx + apy * aby) / den));
    const double dx = apx - t * abx, dy = apy - t * aby;
    return std::sqrt(dx * dx + dy * dy);
}
double quad_distance(const P2 *pa, const P2 *pb) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (pd_segs_intersect(pa[i], pa[(i + 1) & 3], pb[j], pb[(j + 1) & 3])) return 0.0;
    if (pd_point_in_quad(pb[0], pa) || pd_point_in_quad(pa[0], pb)) return 0.0;
    double d = 1e300;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            d = std::min(d, pd_seg_point(pa[i], pb[j], pb[(j + 1) & 3]));
            d = std::min(d, pd_seg_point(pb[j], pa[i], pa[(i + 1) & 3]));
        }
    return d;
}
}  // namespace

extern "C" int mit_quad_pair_distances(const double *quads, int n, const int32_t *pairs, int m, double *out) {
    if (m < 0 || n < 0 || (m > 0 && (!quads || !pairs || !out))) return mit_set_error("mit_quad_pair_distances: null pointer");
    const P2 *q = reinterpret_cast<const P2 *>(quads);
    for (int k = 0; k < m; ++k) {
        const int u = pairs[2 * k], v = pairs[2 * k + 1];
        if (u < 0 || v < 0 || u >= n || v >= n) return mit_set_error("mit_quad_pair_distances: pair %d = (%d, %d) outside [0, %d)", k, u, v, n);
        out[k] = quad_distance(q + 4 * (int64_t)u, q + 4 * (int64_t)v);
    }
    return 0;
}

extern "C" int mit_find_contours_count(const uint8_t *bitmap, int H, int W, int *n_contours, int64_t *n_points) {
    if (!bitmap || !n_contours || !n_points) return mit_set_error("mit_find_contours_count: null pointer");
    std::vector<std::vector<Pt>> contours;
    find_contours(bitmap, H, W, contours);
    *n_contours = (int)contours.size();
    int64_t np = 0;
    for (auto &c : contours) np += (int64_t)c.size();
    *n_points = np;
    return 0;
}

// ---- mask refinement: connected components of the working-scale mask and their assignment to text lines ----------------------
// The host half of complete_mask (mask_refinement/text_mask_utils.py:100-170) ahead of the per-line DenseCRF: outline every line's
// bounding box with zeros, label the 8-connected components, give each component of more than 9 pixels to the line whose polygon
// covers most of its bounding rectangle (or, when none does, to the nearest line within half a glyph).  Components are kept as row
// runs (no page-sized label image, no page-sized image per line): a line's component image is painted from the runs on demand.
namespace {

struct RunUF {
    std::vector<int> parent;
    int find(int a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    }
    void unite(int a, int b) {
        a = find(a), b = find(b);
        if (a != b) parent[a < b ? b : a] = a < b ? a : b;
    }
};

// area of (polygon) ∩ (axis-aligned rectangle): Sutherland-Hodgman against the four sides in turn, shoelace formula
double clip_poly_rect_area(const double *pts, int V, double x0, double y0, double x1, double y1) {
    Pd a[32], b[32];
    int na = 0;
    for (int i = 0; i < V && i < 16; ++i) a[na++] = Pd{pts[2 * i], pts[2 * i + 1]};
    Pd *cur = a, *nxt = b;
    for (int side = 0; side < 4; ++side) {
        if (na == 0) return 0.0;
        const int axis = side >> 1;
        const double bound = side == 0 ? x0 : side == 1 ? x1 : side == 2 ? y0 : y1;
        const bool keep_ge = (side & 1) == 0;
        int nn = 0;
        for (int i = 0; i < na; ++i) {
            const Pd p = cur[i], q = cur[(i + 1) % na];
            const double pc = axis ? p.y : p.x, qc = axis ? q.y : q.x;
            const double dp = keep_ge ? pc - bound : bound - pc, dq = keep_ge ? qc - bound : bound - qc;
            if (dp >= 0 && nn < 32) nxt[nn++] = p;
            if (((dp > 0 && dq < 0) || (dp < 0 && dq > 0)) && nn < 32) {
                const double t = dp / (dp - dq);
                nxt[nn++] = Pd{p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
            }
        }
        std::swap(cur, nxt);
        na = nn;
    }
    if (na < 3) return 0.0;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < na; ++i) {
        const Pd p = cur[i], q = cur[(i + 1) % na];
        s1 += p.x * q.y;
        s2 += p.y * q.x;
    }
    return fabs(s1 - s2) / 2;
}

double poly_area(const double *pts, int V) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < V; ++i) {
        const int j = (i + 1) % V;
        s1 += pts[2 * i] * pts[2 * j + 1];
        s2 += pts[2 * i + 1] * pts[2 * j];
    }
    return fabs(s1 - s2) / 2;
}

// distance from a point to a polygon (0 inside): even-odd crossing test, then the nearest point of every edge
double poly_point_distance(const double *pts, int V, double px, double py) {
    bool inside = false;
    for (int i = 0; i < V; ++i) {
        const int j = (i + 1) % V;
        const double ax = pts[2 * i], ay = pts[2 * i + 1], bx = pts[2 * j], by = pts[2 * j + 1];
        if ((ay > py) != (by > py) && px < (bx - ax) * (py - ay) / (by - ay) + ax) inside = !inside;
    }
    if (inside) return 0.0;
    double best = INFINITY;
    for (int i = 0; i < V; ++i) {
        const int j = (i + 1) % V;
        const double ax = pts[2 * i], ay = pts[2 * i + 1], bx = pts[2 * j], by = pts[2 * j + 1];
        const double abx = bx - ax, aby = by - ay, den = abx * abx + aby * aby;
        double t = den == 0 ? 0.0 : ((px - ax) * abx + (py - ay) * aby) / den;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double dx = px - (ax + t * abx), dy = py - (ay + t * aby);
        best = std::min(best, sqrt(dx * dx + dy * dy));
    }
    return best;
}

}  // namespace

extern "C" int mit_mask_assign_lines(uint8_t *mask, int H, int W, const int32_t *boxes_xywh, const double *polys, const double *font_size,
                                     int M, int V, double keep_threshold, MitMaskRun *runs, int64_t runs_cap, int32_t *assign,
                                     int64_t assign_cap, int32_t *line_rects, int64_t *n_runs_out, int32_t *n_comp_out) {
    if (!mask || !runs || !assign || !line_rects || !n_runs_out || !n_comp_out || (M > 0 && (!boxes_xywh || !polys || !font_size)))
        return mit_set_error("mit_mask_assign_lines: null pointer");
    if (H <= 0 || W <= 0 || M < 0 || V < 3 || V > 16) return mit_set_error("mit_mask_assign_lines: bad shape (H %d W %d M %d V %d)", H, W, M, V);
    // cv2.rectangle(mask, (x, y), (x + w, y + h), 0, 1): one-pixel outline, inclusive corners, clipped (text_mask_utils.py:105-106)
    for (int i = 0; i < M; ++i) {
        const int x = boxes_xywh[4 * i], y = boxes_xywh[4 * i + 1], w = boxes_xywh[4 * i + 2], h = boxes_xywh[4 * i + 3];
        const int xa = std::max(x, 0), xb = std::min(x + w, W - 1), ya = std::max(y, 0), yb = std::min(y + h, H - 1);
        if (xa > xb || ya > yb) continue;
        for (int yy : {y, y + h})
            if (yy >= 0 && yy < H) memset(mask + (int64_t)yy * W + xa, 0, (size_t)(xb - xa + 1));
        for (int xx : {x, x + w})
            if (xx >= 0 && xx < W)
                for (int yy = ya; yy <= yb; ++yy) mask[(int64_t)yy * W + xx] = 0;
    }
    // row runs, united with the runs of the row above that touch them (8-connectivity: a run reaches one pixel to either side)
    int64_t nr = 0;
    RunUF uf;
    int64_t prev0 = 0, prev1 = 0;
    for (int y = 0; y < H; ++y) {
        const uint8_t *row = mask + (int64_t)y * W;
        const int64_t row0 = nr;
        int x = 0;
        while (x < W) {
            if (!row[x]) {
                while (x + 8 <= W) {  // zeros, 8 bytes at a time
                    uint64_t v;
                    memcpy(&v, row + x, 8);
                    if (v) break;
                    x += 8;
                }
                while (x < W && !row[x]) ++x;
                if (x >= W) break;
            }
            const int x0 = x;
            while (x < W && row[x]) ++x;
            if (nr >= runs_cap) return mit_set_error("mit_mask_assign_lines: more than %lld runs", (long long)runs_cap);
            runs[nr] = MitMaskRun{y, x0, x, (int32_t)nr};
            uf.parent.push_back((int)nr);
            ++nr;
        }
        int64_t p = prev0;
        for (int64_t r = row0; r < nr; ++r) {
            while (p < prev1 && runs[p].x1 < runs[r].x0) ++p;  // ends left of the run, not even diagonally adjacent
            for (int64_t q = p; q < prev1 && runs[q].x0 <= runs[r].x1; ++q) uf.unite((int)q, (int)r);
        }
        prev0 = row0, prev1 = nr;
    }
    // component numbers in raster order of their first pixel (1..n), area and bounding rectangle
    std::vector<int> id(nr, 0);
    struct Stat { int64_t area; int x0, y0, x1, y1; };
    std::vector<Stat> st(1);
    for (int64_t r = 0; r < nr; ++r) {
        const int root = uf.find((int)r);
        if (!id[root]) {
            id[root] = (int)st.size();
            st.push_back(Stat{0, INT32_MAX, INT32_MAX, 0, 0});
        }
        const int c = id[root];
        MitMaskRun &ru = runs[r];
        ru.comp = c;
        Stat &s = st[c];
        s.area += ru.x1 - ru.x0;
        s.x0 = std::min(s.x0, ru.x0), s.x1 = std::max(s.x1, ru.x1), s.y0 = std::min(s.y0, ru.y), s.y1 = std::max(s.y1, ru.y + 1);
    }
    const int n = (int)st.size() - 1;
    if (n + 1 > assign_cap) return mit_set_error("mit_mask_assign_lines: %d components, room for %lld", n, (long long)assign_cap - 1);
    std::vector<double> area2(M), pminx(M), pminy(M), pmaxx(M), pmaxy(M);
    for (int i = 0; i < M; ++i) {
        const double *p = polys + (int64_t)i * V * 2;
        area2[i] = poly_area(p, V);
        pminx[i] = pmaxx[i] = p[0], pminy[i] = pmaxy[i] = p[1];
        for (int v = 1; v < V; ++v) {
            pminx[i] = std::min(pminx[i], p[2 * v]), pmaxx[i] = std::max(pmaxx[i], p[2 * v]);
            pminy[i] = std::min(pminy[i], p[2 * v + 1]), pmaxy[i] = std::max(pmaxy[i], p[2 * v + 1]);
        }
    }
    for (int i = 0; i < M; ++i) line_rects[4 * i] = line_rects[4 * i + 1] = line_rects[4 * i + 2] = line_rects[4 * i + 3] = -1;
    assign[0] = -1;
    std::vector<float> ratio(M);
    for (int c = 1; c <= n; ++c) {
        assign[c] = -1;
        const Stat &s = st[c];
        if (s.area <= 9 || M == 0) continue;
        const int x1 = s.x0, y1 = s.y0, w1 = s.x1 - s.x0, h1 = s.y1 - s.y0;
        int avg = 0;
        float best = -1.f;
        for (int i = 0; i < M; ++i) {  // overlap of the line polygon with the component's rectangle over the smaller of the two areas (:129-130), kept in fp32
            float r = 0.f;
            if (pminx[i] <= x1 + w1 && pmaxx[i] >= x1 && pminy[i] <= y1 + h1 && pmaxy[i] >= y1)
                r = (float)(clip_poly_rect_area(polys + (int64_t)i * V * 2, V, x1, y1, x1 + w1, y1 + h1) / std::min((double)s.area, area2[i]));
            ratio[i] = r;
            if (r > best) best = r, avg = i;  // first maximum, like argmax
        }
        if ((double)s.area >= area2[avg]) continue;
        if (ratio[avg] <= (float)keep_threshold) {
            const double cx = x1 + w1 / 2.0, cy = y1 + h1 / 2.0;
            float dbest = INFINITY;
            int di = 0;
            for (int i = 0; i < M; ++i) {
                const float d = (float)poly_point_distance(polys + (int64_t)i * V * 2, V, cx, cy);
                if (d < dbest) dbest = d, di = i;  // first minimum, like argmin
            }
            avg = di;
            const double unit = std::max(std::min(std::min(font_size[avg], (double)w1), (double)h1), 10.0);
            if ((double)dbest >= 0.5 * unit) continue;
        }
        assign[c] = avg;
        int32_t *r = line_rects + 4 * avg;
        if (r[0] < 0) r[0] = x1, r[1] = y1, r[2] = x1 + w1, r[3] = y1 + h1;
        else r[0] = std::min(r[0], x1), r[1] = std::min(r[1], y1), r[2] = std::max(r[2], x1 + w1), r[3] = std::max(r[3], y1 + h1);
    }
    *n_runs_out = nr;
    *n_comp_out = n;
    return 0;
}

extern "C" int mit_mask_line_crops(const MitMaskRun *runs, int64_t n_runs, const int32_t *assign, const int32_t *jobs, int n_jobs, uint8_t *out,
                                   const int64_t *offsets) {
    if (!runs || !assign || !jobs || !out || !offsets) return mit_set_error("mit_mask_line_crops: null pointer");
    int64_t total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int w = jobs[5 * j + 3], h = jobs[5 * j + 4];
        if (w < 0 || h < 0) return mit_set_error("mit_mask_line_crops: job %d has a negative size", j);
        total = std::max(total, offsets[j] + (int64_t)w * h);
    }
    memset(out, 0, (size_t)total);
    // jobs of a line (normally one): lines are few, so a small per-line list is enough
    int max_line = -1;
    for (int j = 0; j < n_jobs; ++j) max_line = std::max(max_line, jobs[5 * j]);
    std::vector<std::vector<int>> by_line(max_line + 1);
    for (int j = 0; j < n_jobs; ++j)
        if (jobs[5 * j] >= 0) by_line[jobs[5 * j]].push_back(j);
    for (int64_t r = 0; r < n_runs; ++r) {
        const MitMaskRun &ru = runs[r];
        const int line = assign[ru.comp];
        if (line < 0 || line > max_line) continue;
        for (int j : by_line[line]) {
            const int x = jobs[5 * j + 1], y = jobs[5 * j + 2], w = jobs[5 * j + 3], h = jobs[5 * j + 4];
            if (ru.y < y || ru.y >= y + h) continue;
            const int a = std::max(ru.x0, x), b = std::min(ru.x1, x + w);
            if (a < b) memset(out + offsets[j] + (int64_t)(ru.y - y) * w + (a - x), 255, (size_t)(b - a));
        }
    }
    return 0;
}
