"""Text-line geometry on the host: the ``Quadrilateral`` type that crosses every stage boundary and the
plan for rectifying a line into a 48-px-high crop.

Follows the reference's ``sort_pnts`` / ``Quadrilateral`` (/root/reference/manga_translator/utils/generic.py:324-443)
and the geometry half of ``Quadrilateral.get_transformed_region`` (:445-481).  This is the interface type of the plugin
boundary and its decision logic is integer-exact: ``sort_pnts`` (:334-353) and the predicate chain of
``quadrilateral_can_merge_region`` (:656-695) are restated step for step — same comparisons, same thresholds, in the same order —
because any other formulation changes which lines merge; results are pinned to the reference's own functions
(tests/golden/textline.npz, direction.npz).  The pixel half of that function
(cv2.warpPerspective + cv2.rotate) runs on the GPU: ``warp_plan`` only produces the crop rectangle, the destination
size and the inverse homography that ``mit_ocr_warp_lines`` (csrc/ocr_warp.hip) consumes.
"""
from __future__ import annotations

import functools
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


def _sort_pnts_np(pts: np.ndarray) -> Tuple[np.ndarray, bool]:
    """generic.py:324-354 operation for operation (numpy on 16 x 2 and 4 x 2 arrays: ~40-80 us per quadrilateral).  sort_pnts hands it
    every input whose result could depend on how numpy's (unstable) argsort orders equal keys."""
    pts = np.asarray(pts)
    if pts.shape != (4, 2):
        raise ValueError(f"sort_pnts expects 4 points, got shape {pts.shape}")
    pairwise_vec = (pts[:, None] - pts[None]).reshape((16, -1))
    pairwise_vec_norm = np.linalg.norm(pairwise_vec, axis=1)
    long_side_ids = np.argsort(pairwise_vec_norm)[[8, 10]]
    long_side_vecs = pairwise_vec[long_side_ids]
    inner_prod = (long_side_vecs[0] * long_side_vecs[1]).sum()
    if inner_prod < 0:
        long_side_vecs[0] = -long_side_vecs[0]
    struc_vec = np.abs(long_side_vecs.mean(axis=0))
    is_vertical = bool(struc_vec[0] <= struc_vec[1])
    if is_vertical:
        pts = pts[np.argsort(pts[:, 1])]
        pts = pts[[*np.argsort(pts[:2, 0]), *np.argsort(pts[2:, 0])[::-1] + 2]]
        return pts, is_vertical
    pts = pts[np.argsort(pts[:, 0])]
    pts_sorted = np.zeros_like(pts)
    pts_sorted[[0, 3]] = sorted(pts[[0, 1]], key=lambda x: x[1])
    pts_sorted[[1, 2]] = sorted(pts[[2, 3]], key=lambda x: x[1])
    return pts_sorted, is_vertical


def sort_pnts(pts: np.ndarray) -> Tuple[np.ndarray, bool]:
    """Canonical corner order [tl, tr, br, bl] and the vertical flag (generic.py:324-354).  ONE path in every environment: this
    restatement, pinned to the reference's function by tests/golden/textline.npz — the product never executes reference code
    other than the plugin base classes it subclasses.

    The reference's numpy formulation costs 40-80 us per quadrilateral (a page's 32 lines: 1.5-2.5 ms of every B = 1 OCR call).  This is
    the same decision sequence on Python scalars: the entries 8 and 10 of the ascending pairwise-vector norms, the sign fix, the
    |mean| comparison, the two-level corner ordering.  Wherever the outcome could depend on the ORDER numpy's argsort gives equal keys
    (a tie class at position 8 or 10 holding non-parallel vectors — e.g. an exact square —, a zero inner product, equal sort keys
    between the corner groups) the input goes to _sort_pnts_np, so both agree on every input (tests/test_textline.py fuzzes that)."""
    pts = np.asarray(pts)
    if pts.shape != (4, 2):
        raise ValueError(f"sort_pnts expects 4 points, got shape {pts.shape}")
    p = pts.tolist()
    vec = [(p[i][0] - p[j][0], p[i][1] - p[j][1]) for i in range(4) for j in range(4)]
    nrm = [math.sqrt(float(dx) * float(dx) + float(dy) * float(dy)) for dx, dy in vec]
    order = sorted(range(16), key=nrm.__getitem__)

    def tie_class_parallel(k):
        a, j = vec[order[k]], k
        while j > 0 and nrm[order[j - 1]] == nrm[order[k]]:
            j -= 1
        while j < 16 and nrm[order[j]] == nrm[order[k]]:
            b = vec[order[j]]
            if a[0] * b[1] - a[1] * b[0] != 0:
                return False
            j += 1
        return True

    if not (tie_class_parallel(8) and tie_class_parallel(10)):
        return _sort_pnts_np(pts)
    a, b = vec[order[8]], vec[order[10]]
    inner = a[0] * b[0] + a[1] * b[1]
    if inner == 0:
        return _sort_pnts_np(pts)
    if inner < 0:
        a = (-a[0], -a[1])
    is_vertical = bool(abs((a[0] + b[0]) / 2) <= abs((a[1] + b[1]) / 2))
    if is_vertical:
        o = sorted(range(4), key=lambda i: p[i][1])
        if p[o[1]][1] == p[o[2]][1] or (p[o[0]][0] == p[o[1]][0] and p[o[0]] != p[o[1]]) or (p[o[2]][0] == p[o[3]][0] and p[o[2]] != p[o[3]]):
            return _sort_pnts_np(pts)
        top = sorted(o[:2], key=lambda i: p[i][0])
        bot = sorted(o[2:], key=lambda i: p[i][0])[::-1]
        return pts[[top[0], top[1], bot[0], bot[1]]], is_vertical
    o = sorted(range(4), key=lambda i: p[i][0])
    if p[o[1]][0] == p[o[2]][0]:
        return _sort_pnts_np(pts)
    le = sorted(o[:2], key=lambda i: p[i][1])
    ri = sorted(o[2:], key=lambda i: p[i][1])
    return pts[[le[0], ri[0], ri[1], le[1]]], is_vertical


@dataclass
class BBox:
    x: int
    y: int
    w: int
    h: int


class Quadrilateral:
    """Text line: 4 corner points + recognised text / colours (generic.py:356-443, the fields the dense path touches)."""

    def __init__(self, pts: np.ndarray, text: str = "", prob: float = 0.0, fg_r: int = 0, fg_g: int = 0, fg_b: int = 0,
                 bg_r: int = 0, bg_g: int = 0, bg_b: int = 0):
        self.pts, is_vertical = sort_pnts(pts)
        self.direction = "v" if is_vertical else "h"
        self.text, self.prob = text, prob
        self.fg_r, self.fg_g, self.fg_b = fg_r, fg_g, fg_b
        self.bg_r, self.bg_g, self.bg_b = bg_r, bg_g, bg_b
        self.assigned_direction = None
        self.textlines: List["Quadrilateral"] = []

    @functools.cached_property
    def structure(self) -> List[np.ndarray]:
        p1 = ((self.pts[0] + self.pts[1]) / 2).astype(int)
        p2 = ((self.pts[2] + self.pts[3]) / 2).astype(int)
        p3 = ((self.pts[1] + self.pts[2]) / 2).astype(int)
        p4 = ((self.pts[3] + self.pts[0]) / 2).astype(int)
        return [p1, p2, p3, p4]

    def _vecs(self):
        l1a, l1b, l2a, l2b = [a.astype(np.float32) for a in self.structure]
        return l1b - l1a, l2b - l2a

    @functools.cached_property
    def valid(self) -> bool:
        v1, v2 = self._vecs()
        u1, u2 = v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2)
        return bool(abs(np.arccos(np.dot(u1, u2)) * 180 / np.pi - 90) < 10)

    @functools.cached_property
    def aspect_ratio(self) -> float:
        v1, v2 = self._vecs()
        return float(np.linalg.norm(v2) / np.linalg.norm(v1))

    @functools.cached_property
    def font_size(self) -> float:
        v1, v2 = self._vecs()
        return float(min(np.linalg.norm(v2), np.linalg.norm(v1)))

    @functools.cached_property
    def aabb(self) -> BBox:
        mx, mn = np.max(self.pts, axis=0), np.min(self.pts, axis=0)
        return BBox(int(mn[0]), int(mn[1]), int(mx[0] - mn[0]), int(mx[1] - mn[1]))

    @functools.cached_property
    def extent(self) -> Tuple[float, float, float, float]:
        """(xmin, ymin, xmax, ymax) of the points as floats (``aabb`` truncates to int)."""
        p = np.asarray(self.pts, dtype=np.float64)
        return float(p[:, 0].min()), float(p[:, 1].min()), float(p[:, 0].max()), float(p[:, 1].max())

    @functools.cached_property
    def area(self) -> float:
        x, y = self.pts[:, 0].astype(np.float64), self.pts[:, 1].astype(np.float64)
        return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)  # convex quad: hull area = shoelace

    def clip(self, width: int, height: int) -> None:
        self.pts[:, 0] = np.clip(np.round(self.pts[:, 0]), 0, width)
        self.pts[:, 1] = np.clip(np.round(self.pts[:, 1]), 0, height)

    @functools.cached_property
    def xyxy(self):
        return self.aabb.x, self.aabb.y, self.aabb.x + self.aabb.w, self.aabb.y + self.aabb.h

    def _unit_vecs(self):
        v1, v2 = self._vecs()
        return v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2)

    @functools.cached_property
    def is_axis_aligned(self) -> bool:  # generic.py:483-494
        u1, _ = self._unit_vecs()
        return bool(abs(u1[1]) < 1e-2 or abs(u1[0]) < 1e-2)

    @functools.cached_property
    def is_approximate_axis_aligned(self) -> bool:  # generic.py:496-507
        u1, u2 = self._unit_vecs()
        return bool(abs(u1[1]) < 0.05 or abs(u1[0]) < 0.05 or abs(u2[1]) < 0.05 or abs(u2[0]) < 0.05)

    @functools.cached_property
    def cosangle(self) -> float:  # generic.py:509-515
        u1, _ = self._unit_vecs()
        return float(np.dot(u1, np.array([1, 0])))

    @functools.cached_property
    def angle(self) -> float:  # generic.py:517-519
        return float(np.fmod(np.arccos(self.cosangle) + np.pi, np.pi))

    @functools.cached_property
    def centroid(self) -> np.ndarray:
        return np.average(self.pts, axis=0)

    def poly_distance(self, other: "Quadrilateral") -> float:
        """shapely ``self.polygon.distance(other.polygon)`` with polygon = convex hull of the 4 points (generic.py:533-541)."""
        return polygon_distance(_convex_hull(self.pts), _convex_hull(other.pts))


def _convex_hull(pts: np.ndarray) -> np.ndarray:
    p = sorted(set(map(tuple, np.asarray(pts, dtype=np.float64))))
    if len(p) < 3:
        return np.array(p, dtype=np.float64)
    cross = lambda o, a, b: (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo, up = [], []
    for q in p:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], q) <= 0:
            lo.pop()
        lo.append(q)
    for q in reversed(p):
        while len(up) >= 2 and cross(up[-2], up[-1], q) <= 0:
            up.pop()
        up.append(q)
    return np.array(lo[:-1] + up[:-1], dtype=np.float64)


def _seg_point_dist(p, a, b) -> float:
    """Distance of point p from segment ab (plain Python floats: this runs O(lines^2 x 32) times per page)."""
    abx, aby, apx, apy = b[0] - a[0], b[1] - a[1], p[0] - a[0], p[1] - a[1]
    den = abx * abx + aby * aby
    t = 0.0 if den == 0 else min(1.0, max(0.0, (apx * abx + apy * aby) / den))
    dx, dy = apx - t * abx, apy - t * aby
    return math.sqrt(dx * dx + dy * dy)


def _segs_intersect(a, b, c, d) -> bool:
    o = lambda p, q, r: (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
    on = lambda p, q, r: min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])
    o1, o2, o3, o4 = o(a, b, c), o(a, b, d), o(c, d, a), o(c, d, b)
    if ((o1 > 0) != (o2 > 0)) and ((o3 > 0) != (o4 > 0)) and o1 * o2 != 0 and o3 * o4 != 0:
        return True
    return (o1 == 0 and on(a, b, c)) or (o2 == 0 and on(a, b, d)) or (o3 == 0 and on(c, d, a)) or (o4 == 0 and on(c, d, b))


def _point_in_polygon(p, poly) -> bool:
    inside = False
    n = len(poly)
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        if (a[1] > p[1]) != (b[1] > p[1]) and p[0] < (b[0] - a[0]) * (p[1] - a[1]) / (b[1] - a[1]) + a[0]:
            inside = not inside
    return inside


def polygon_distance(pa: np.ndarray, pb: np.ndarray) -> float:
    """Minimum distance between two simple polygons given as vertex rings (0 when they touch or overlap) — what
    shapely's ``Polygon.distance`` returns for the quads of utils/generic.py:660-662."""
    pa, pb = np.asarray(pa, dtype=np.float64).tolist(), np.asarray(pb, dtype=np.float64).tolist()  # lists of Python floats
    na, nb = len(pa), len(pb)
    for i in range(na):
        for j in range(nb):
            if _segs_intersect(pa[i], pa[(i + 1) % na], pb[j], pb[(j + 1) % nb]):
                return 0.0
    if (na >= 3 and _point_in_polygon(pb[0], pa)) or (nb >= 3 and _point_in_polygon(pa[0], pb)):
        return 0.0
    d = 1e300
    for i in range(na):
        for j in range(nb):
            d = min(d, _seg_point_dist(pa[i], pb[j], pb[(j + 1) % nb]), _seg_point_dist(pb[j], pa[i], pa[(i + 1) % na]))
    return d


def quad_pair_distances(quads: Sequence["Quadrilateral"], pairs: Sequence[Tuple[int, int]]) -> List[float]:
    """``polygon_distance(quads[u].pts, quads[v].pts)`` for every pair at once, in native host code (``mit_quad_pair_distances``: the
    same double arithmetic, operation for operation — tests/test_textline.py compares them bit for bit): a page's direction vote and
    merge graph ask for ~60 such distances, 75 us each in the interpreter."""
    if not pairs:
        return []
    import ctypes as C

    from . import lib as _lib

    pts = np.ascontiguousarray([q.pts for q in quads], dtype=np.float64)
    if pts.ndim != 3 or pts.shape[1:] != (4, 2):
        return [polygon_distance(quads[u].pts, quads[v].pts) for u, v in pairs]
    pr = np.ascontiguousarray(pairs, dtype=np.int32)
    out = np.empty(len(pr), dtype=np.float64)
    _lib.check(_lib.load().mit_quad_pair_distances(pts.ctypes.data, len(pts), pr.ctypes.data, len(pr), out.ctypes.data), "mit_quad_pair_distances")
    return out.tolist()


def quadrilateral_can_merge_region(a: Quadrilateral, b: Quadrilateral, ratio=1.9, discard_connection_gap=2, char_gap_tolerance=0.6,
                                   char_gap_tolerance2=1.5, font_size_ratio_tol=1.5, aspect_ratio_tol=2, dist: Optional[float] = None) -> bool:
    """utils/generic.py:653-698, line for line (shapely's polygon distance replaced by ``polygon_distance``).  ``dist``: that distance
    when the caller already has it (``quad_pair_distances`` over all near pairs of a page)."""
    b1, b2 = a.aabb, b.aabb
    char_size = min(a.font_size, b.font_size)
    x1, y1, w1, h1 = b1.x, b1.y, b1.w, b1.h
    x2, y2, w2, h2 = b2.x, b2.y, b2.w, b2.h
    # the gap between the axis-aligned boxes is a lower bound of the polygon distance: far-apart pairs (most of the O(K^2)
    # pairs of a page) fail the first test of the reference without the exact distance being needed — same decisions, always
    # (the gap is taken from the float extents of the points, not from ``aabb``: its int() truncation can fall short of the true
    # extent for non-integer points, and the bound must never exceed the real distance)
    ea, eb = a.extent, b.extent
    gx = max(0.0, max(ea[0], eb[0]) - min(ea[2], eb[2]))
    gy = max(0.0, max(ea[1], eb[1]) - min(ea[3], eb[3]))
    if gx * gx + gy * gy > (discard_connection_gap * char_size) ** 2:
        return False
    if dist is None:
        dist = polygon_distance(a.pts, b.pts)  # Polygon(a.pts).distance(Polygon(b.pts))
    if dist > discard_connection_gap * char_size:
        return False
    if max(a.font_size, b.font_size) / char_size > font_size_ratio_tol:
        return False
    if a.aspect_ratio > aspect_ratio_tol and b.aspect_ratio < 1. / aspect_ratio_tol:
        return False
    if b.aspect_ratio > aspect_ratio_tol and a.aspect_ratio < 1. / aspect_ratio_tol:
        return False
    if a.is_approximate_axis_aligned and b.is_approximate_axis_aligned:
        if dist < char_size * char_gap_tolerance:
            if abs(x1 + w1 // 2 - (x2 + w2 // 2)) < char_gap_tolerance2:
                return True
            if w1 > h1 * ratio and h2 > w2 * ratio:
                return False
            if w2 > h2 * ratio and h1 > w1 * ratio:
                return False
            if w1 > h1 * ratio or w2 > h2 * ratio:
                return abs(x1 - x2) < char_size * char_gap_tolerance2 or abs(x1 + w1 - (x2 + w2)) < char_size * char_gap_tolerance2
            elif h1 > w1 * ratio or h2 > w2 * ratio:
                return abs(y1 - y2) < char_size * char_gap_tolerance2 or abs(y1 + h1 - (y2 + h2)) < char_size * char_gap_tolerance2
            return False
        return False
    if abs(a.angle - b.angle) < 15 * np.pi / 180:
        fs_a, fs_b = a.font_size, b.font_size
        fs = min(fs_a, fs_b)
        if a.poly_distance(b) > fs * char_gap_tolerance2:
            return False
        if abs(fs_a - fs_b) / fs > 0.25:
            return False
        return True
    return False


def prefill_geometry(quads: Sequence[Quadrilateral]) -> bool:
    """Fills the cached geometry of all quads of a page in ONE vectorised pass: structure, font_size, aspect_ratio, aabb, extent,
    is_approximate_axis_aligned — what the direction vote and the merge graph read from every line (per quad these are ~25 numpy calls
    on 2-vectors, 70 us a line in the interpreter).  The same numpy operations on [n, ...] arrays, elementwise identical; the one
    operation that is not elementwise — np.linalg.norm's float32 dot product, which a BLAS may fuse — is exact whenever the structure
    vectors stay below 2048 (squares and their sum fit the 24-bit significand), so the fill is limited to such pages and to uniform
    integer / float point arrays; anything else keeps the lazy per-quad properties.  Returns whether the fill was applied."""
    todo = [q for q in quads if "font_size" not in q.__dict__]
    if len(todo) < 2:
        return False
    dt = todo[0].pts.dtype
    if any(q.pts.dtype != dt or q.pts.shape != (4, 2) or "structure" in q.__dict__ for q in todo):
        return False
    P = np.stack([q.pts for q in todo])
    p1 = ((P[:, 0] + P[:, 1]) / 2).astype(int)
    p2 = ((P[:, 2] + P[:, 3]) / 2).astype(int)
    p3 = ((P[:, 1] + P[:, 2]) / 2).astype(int)
    p4 = ((P[:, 3] + P[:, 0]) / 2).astype(int)
    v1 = p2.astype(np.float32) - p1.astype(np.float32)
    v2 = p4.astype(np.float32) - p3.astype(np.float32)
    if max(float(np.abs(v1).max()), float(np.abs(v2).max())) > 2048:
        return False
    n1 = np.sqrt(v1[:, 0] * v1[:, 0] + v1[:, 1] * v1[:, 1])
    n2 = np.sqrt(v2[:, 0] * v2[:, 0] + v2[:, 1] * v2[:, 1])
    if not (n1.min() > 0 and n2.min() > 0):
        return False
    ar, fs = n2 / n1, np.minimum(n2, n1)
    u1, u2 = v1 / n1[:, None], v2 / n2[:, None]
    approx = (np.abs(u1[:, 1]) < 0.05) | (np.abs(u1[:, 0]) < 0.05) | (np.abs(u2[:, 1]) < 0.05) | (np.abs(u2[:, 0]) < 0.05)
    mx, mn = P.max(axis=1), P.min(axis=1)
    Pf = P.astype(np.float64)
    fmn, fmx = Pf.min(axis=1), Pf.max(axis=1)
    for i, q in enumerate(todo):
        d = q.__dict__
        d["structure"] = [p1[i], p2[i], p3[i], p4[i]]
        d["font_size"] = float(fs[i])
        d["aspect_ratio"] = float(ar[i])
        d["is_approximate_axis_aligned"] = bool(approx[i])
        d["aabb"] = BBox(int(mn[i, 0]), int(mn[i, 1]), int(mx[i, 0] - mn[i, 0]), int(mx[i, 1] - mn[i, 1]))
        d["extent"] = (float(fmn[i, 0]), float(fmn[i, 1]), float(fmx[i, 0]), float(fmx[i, 1]))
    return True


def near_pairs(quads: Sequence[Quadrilateral], discard_connection_gap: float = 2) -> List[Tuple[int, int]]:
    """The pairs (u < v, in itertools.combinations order) that pass ``quadrilateral_can_merge_region``'s own first test — bounding-box
    gap against ``discard_connection_gap`` x the smaller font size — evaluated for all pairs at once: of the O(n^2) pairs of a page
    only the handful of neighbours has to go through the Python predicate (which repeats the test)."""
    n = len(quads)
    if n < 2:
        return []
    ext = np.array([q.extent for q in quads], dtype=np.float64)
    fs = np.array([q.font_size for q in quads], dtype=np.float64)
    gx = np.maximum(0.0, np.maximum(ext[:, None, 0], ext[None, :, 0]) - np.minimum(ext[:, None, 2], ext[None, :, 2]))
    gy = np.maximum(0.0, np.maximum(ext[:, None, 1], ext[None, :, 1]) - np.minimum(ext[:, None, 3], ext[None, :, 3]))
    near = np.triu(gx * gx + gy * gy <= (discard_connection_gap * np.minimum(fs[:, None], fs[None, :])) ** 2, 1)
    us, vs = np.nonzero(near)
    return list(zip(us.tolist(), vs.tolist()))


def generate_text_direction(quads: Sequence[Quadrilateral]):
    """CommonOCR._generate_text_direction (ocr/common.py:12-39): connected components of the merge graph
    (aspect_ratio_tol = 1), majority direction per component, lines ordered top-to-bottom ('h') or right-to-left ('v').
    Yields (quad, direction)."""
    n = len(quads)
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    prefill_geometry(quads)
    pairs = near_pairs(quads)
    for (u, v), d in zip(pairs, quad_pair_distances(quads, pairs)):
        if quadrilateral_can_merge_region(quads[u], quads[v], aspect_ratio_tol=1, dist=d):
            ru, rv = find(u), find(v)
            if ru != rv:
                parent[max(ru, rv)] = min(ru, rv)
    comps = {}
    for i in range(n):  # components in order of their first node, nodes ascending (networkx + CPython set order for small ints)
        comps.setdefault(find(i), []).append(i)
    for root in sorted(comps):
        nodes = comps[root]
        dirs = [quads[i].direction for i in nodes]
        counts = {}
        for d in dirs:  # Counter.most_common(1): highest count, first encountered on ties
            counts[d] = counts.get(d, 0) + 1
        majority = max(counts, key=lambda d: (counts[d], -dirs.index(d)))
        if majority == "h":
            nodes = sorted(nodes, key=lambda i: quads[i].aabb.y + quads[i].aabb.h // 2)
        else:
            nodes = sorted(nodes, key=lambda i: -(quads[i].aabb.x + quads[i].aabb.w))
        for i in nodes:
            yield quads[i], majority


def homography_4pt(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """3x3 H (h33 = 1) with H @ [src,1] ~ [dst,1] for exactly four correspondences.

    Stands in for ``cv2.findHomography(src, dst, cv2.RANSAC, 5.0)`` (generic.py:471,478): with four points RANSAC has a
    single candidate model and every point is an inlier, so the result is the exact solution of the 8x8 system."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    A = np.zeros((8, 8), dtype=np.float64)
    b = np.zeros(8, dtype=np.float64)
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        A[2 * i] = [x, y, 1, 0, 0, 0, -u * x, -u * y]
        A[2 * i + 1] = [0, 0, 0, x, y, 1, -v * x, -v * y]
        b[2 * i], b[2 * i + 1] = u, v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)


@dataclass
class WarpPlan:
    """Everything ``mit_ocr_warp_lines`` needs for one line; ``width`` is the crop's width inside the OCR chunk."""
    x1: int
    y1: int
    cw: int
    ch: int
    dw: int
    dh: int
    vertical: bool
    minv: np.ndarray  # [3,3] float64, destination -> crop coordinates

    @property
    def width(self) -> int:
        return self.dh if self.vertical else self.dw


def warp_plan(quad: Quadrilateral, direction: str, im_h: int, im_w: int, textheight: int = 48) -> WarpPlan:
    """Geometry of get_transformed_region (generic.py:445-481) without touching pixels."""
    l1a, l1b, l2a, l2b = [a.astype(np.float32) for a in quad.structure]
    v_vec, h_vec = l1b - l1a, l2b - l2a
    ratio = np.linalg.norm(v_vec) / np.linalg.norm(h_vec)
    src_pts = quad.pts.astype(np.int64).copy()
    x1, y1, x2, y2 = src_pts[:, 0].min(), src_pts[:, 1].min(), src_pts[:, 0].max(), src_pts[:, 1].max()
    x1, x2 = int(np.clip(x1, 0, im_w)), int(np.clip(x2, 0, im_w))
    y1, y2 = int(np.clip(y1, 0, im_h)), int(np.clip(y2, 0, im_h))
    src_pts[:, 0] -= x1
    src_pts[:, 1] -= y1
    quad.assigned_direction = direction
    if direction == "h":
        h = max(int(textheight), 2)
        w = max(int(round(textheight / ratio)), 2)
    elif direction == "v":
        w = max(int(textheight), 2)
        h = max(int(round(textheight * ratio)), 2)
    else:
        raise ValueError(f"direction must be 'h' or 'v', got {direction!r}")
    dst_pts = np.array([[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], dtype=np.float32)
    M = homography_4pt(src_pts, dst_pts)
    return WarpPlan(x1, y1, x2 - x1, y2 - y1, w, h, direction == "v", np.linalg.inv(M))


def chunk_plan(widths: Sequence[int], max_chunk_size: int = 16) -> List[Tuple[List[int], List[int], int]]:
    """Model48pxOCR._infer's batching (model_48px.py:79-86): lines sorted by crop width, consecutive groups of 16,
    each padded to ``4 * (max(widths) + 7) // 4`` (== max + 7).  Returns [(indices, widths, padded_width)]."""
    perm = sorted(range(len(widths)), key=lambda i: widths[i])
    out = []
    for c in range(0, len(perm), max_chunk_size):
        idx = perm[c:c + max_chunk_size]
        ws = [int(widths[i]) for i in idx]
        out.append((idx, ws, 4 * (max(ws) + 7) // 4))
    return out


# numpy mirror of the C struct MitWarpLine (include/mit_hip.h): 9 doubles + 10 int32 = 112 bytes
WARP_LINE_DTYPE = np.dtype([("minv", "<f8", (9,)), ("page", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("cw", "<i4"), ("ch", "<i4"),
                            ("dw", "<i4"), ("dh", "<i4"), ("vertical", "<i4"), ("out_row", "<i4"), ("_pad", "<i4")])


def warp_plans(quads: Sequence[Quadrilateral], directions: Sequence[str], im_h: int, im_w: int, textheight: int = 48) -> np.ndarray:
    """``warp_plan`` for all lines of a page at once (same arithmetic, batched): returns WARP_LINE_DTYPE records with
    page / out_row left at 0.  One batched 8x8 solve + 3x3 inverse instead of 2K tiny LAPACK calls."""
    K = len(quads)
    rec = np.zeros(K, dtype=WARP_LINE_DTYPE)
    if K == 0:
        return rec
    pts = np.stack([np.asarray(q.pts) for q in quads])                      # [K,4,2] canonical order
    mid = lambda a, b: ((pts[:, a] + pts[:, b]) / 2).astype(int)            # Quadrilateral.structure
    p1, p2, p3, p4 = mid(0, 1), mid(2, 3), mid(1, 2), mid(3, 0)
    v_vec = (p2 - p1).astype(np.float32)
    h_vec = (p4 - p3).astype(np.float32)
    ratio = np.linalg.norm(v_vec, axis=1) / np.linalg.norm(h_vec, axis=1)   # float32, like the reference
    src = pts.astype(np.int64)
    x1 = np.clip(src[:, :, 0].min(1), 0, im_w)
    x2 = np.clip(src[:, :, 0].max(1), 0, im_w)
    y1 = np.clip(src[:, :, 1].min(1), 0, im_h)
    y2 = np.clip(src[:, :, 1].max(1), 0, im_h)
    src = src - np.stack([x1, y1], axis=1)[:, None, :]
    vert = np.array([d == "v" for d in directions])
    if not all(d in ("h", "v") for d in directions):
        raise ValueError("direction must be 'h' or 'v'")
    th = max(int(textheight), 2)
    long_side = np.where(vert, [max(int(round(float(np.float32(textheight) * r))), 2) for r in ratio],
                         [max(int(round(float(np.float32(textheight) / r))), 2) for r in ratio]).astype(np.int64)
    dw = np.where(vert, th, long_side)
    dh = np.where(vert, long_side, th)
    dst = np.zeros((K, 4, 2), dtype=np.float64)
    dst[:, 1, 0] = dst[:, 2, 0] = (dw - 1).astype(np.float32)
    dst[:, 2, 1] = dst[:, 3, 1] = (dh - 1).astype(np.float32)
    A = np.zeros((K, 8, 8), dtype=np.float64)
    b = np.zeros((K, 8), dtype=np.float64)
    x, y = src[:, :, 0].astype(np.float64), src[:, :, 1].astype(np.float64)
    u, v = dst[:, :, 0], dst[:, :, 1]
    A[:, 0::2, 0], A[:, 0::2, 1], A[:, 0::2, 2] = x, y, 1.0
    A[:, 0::2, 6], A[:, 0::2, 7] = -u * x, -u * y
    A[:, 1::2, 3], A[:, 1::2, 4], A[:, 1::2, 5] = x, y, 1.0
    A[:, 1::2, 6], A[:, 1::2, 7] = -v * x, -v * y
    b[:, 0::2], b[:, 1::2] = u, v
    h8 = np.linalg.solve(A, b[:, :, None])[:, :, 0]
    M = np.concatenate([h8, np.ones((K, 1))], axis=1).reshape(K, 3, 3)
    rec["minv"] = np.linalg.inv(M).reshape(K, 9)
    rec["x1"], rec["y1"], rec["cw"], rec["ch"] = x1, y1, x2 - x1, y2 - y1
    rec["dw"], rec["dh"], rec["vertical"] = dw, dh, vert
    for q, d in zip(quads, directions):
        q.assigned_direction = d
    return rec


def is_ignore(region_img: np.ndarray, ignore_bubble: int = 0) -> bool:
    """``--ignore-bubble`` filter on a rectified line crop (utils/bubble.py:28-84): share of dark (<= 127) values in the 2-pixel frame of
    the crop; inside [ignore_bubble, 100 - ignore_bubble] % the line is taken for text on artwork and skipped, and so is any crop with
    more than 10 coloured pixels (squared distance to its own luma > 100).  Values outside 1..50 switch the filter off."""
    if ignore_bubble < 1 or ignore_bubble > 50:
        return False
    dark = np.asarray(region_img) <= 127                       # cv2.threshold(img, 127, 255, THRESH_BINARY) == 0
    h, w = dark.shape[:2]
    frame = [dark[0:2, 0:w], dark[h - 2:h, 0:w], dark[2:h - 2, 0:2], dark[2:h - 2, w - 2:w]]
    val0 = sum(int(f.sum()) for f in frame)
    total = sum(f.size for f in frame)
    ratio = round(val0 / total, 6) * 100
    if ignore_bubble <= ratio <= 100 - ignore_bubble:
        return True
    img = np.asarray(region_img)
    gray = np.dot(img[..., :3], [0.299, 0.587, 0.114])[..., np.newaxis]
    return bool(np.sum(np.sum((img - gray) ** 2, axis=-1) > 100) > 10)
