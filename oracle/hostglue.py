"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the detectors' box extraction, independent of csrc/hostglue.hip.

Restates ``SegDetectorRepresenter.boxes_from_bitmap`` (+ get_mini_boxes / box_score_fast / unclip) of
/root/reference/manga_translator/detection/ctd_utils/utils/db_utils.py:127-216 and default_utils/dbnet_utils.py:97-190
WITHOUT border following: contours are characterised through connected components (scipy.ndimage) —
  outer border of an 8-connected component  -> its pixels (same hull), filled = component with holes filled,
  hole border (4-connected background hole) -> the foreground pixels 4-adjacent to the hole, filled = ring + hole,
ordered by where a raster scan would start tracing them, reversed (OpenCV's list order).  Min-area rectangles by brute
force over hull edges; the pyclipper round offset follows ClipperOffset (DoOffset / OffsetPoint / DoRound).

Parity status: the PRIMITIVES are **unpinned** — OpenCV, pyclipper and shapely are not installed anywhere this can run.  The
control flow around them is pinned: oracle/ref_import.segdet runs the reference's own SegDetectorRepresenter Python of both
detectors with stand-ins built from this file and oracle/contours.py, oracle/make_golden.golden_boxes records its output
(tests/golden/boxes.npz), and the native routine reproduces it row for row (tests/test_hostglue.py).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
from scipy import ndimage


def _hull(points: np.ndarray) -> np.ndarray:
    pts = sorted(set(map(tuple, np.asarray(points, dtype=np.float64))))
    if len(pts) < 3:
        return np.array(pts, dtype=np.float64)
    cross = lambda o, a, b: (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lower: List = []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper: List = []
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1], dtype=np.float64)


def min_area_rect(points: np.ndarray) -> Tuple[np.ndarray, float]:
    """cv2.minAreaRect + cv2.boxPoints -> (4 corners float32 in get_mini_boxes order, short side)."""
    h = _hull(points)
    if len(h) == 1:
        return np.repeat(h.astype(np.float32), 4, axis=0), 0.0
    best = None
    n = len(h)
    for e in range(1 if n == 2 else n):
        a, b = h[e], h[(e + 1) % n]
        u = (b - a) / np.linalg.norm(b - a)
        v = np.array([-u[1], u[0]])
        pu, pv = h @ u, h @ v
        area = (pu.max() - pu.min()) * (pv.max() - pv.min())
        if best is None or area < best[0]:
            best = (area, u, v, pu.min(), pu.max(), pv.min(), pv.max())
    _, u, v, u0, u1, v0, v1 = best
    corners = np.array([u0 * u + v0 * v, u1 * u + v0 * v, u1 * u + v1 * v, u0 * u + v1 * v]).astype(np.float32)
    pts = sorted(list(corners), key=lambda p: p[0])
    i1, i4 = (0, 1) if pts[1][1] > pts[0][1] else (1, 0)
    i2, i3 = (2, 3) if pts[3][1] > pts[2][1] else (3, 2)
    return np.array([pts[i1], pts[i2], pts[i3], pts[i4]], dtype=np.float32), float(min(u1 - u0, v1 - v0))


def clipper_offset_round(box: np.ndarray, delta: float) -> np.ndarray:
    """pyclipper.PyclipperOffset().AddPath(box, JT_ROUND, ET_CLOSEDPOLYGON); Execute(delta) — vertex set of the result."""
    rnd = lambda v: int(v - 0.5) if v < 0 else int(v + 0.5)
    path = []
    for x, y in box:
        p = (int(x), int(y))  # truncation, like pyclipper's conversion of float coordinates
        if not path or path[-1] != p:
            path.append(p)
    while len(path) > 1 and path[0] == path[-1]:
        path.pop()
    n = len(path)
    if n < 3 or delta <= 0:
        return np.zeros((0, 2))
    area = sum(path[i][0] * path[(i + 1) % n][1] - path[(i + 1) % n][0] * path[i][1] for i in range(n))
    if area < 0:
        path.reverse()
    y = min(0.25, abs(delta) * 0.25)
    steps = math.pi / math.acos(1 - y / abs(delta))
    steps = min(steps, abs(delta) * math.pi)
    m_sin, m_cos, spr = math.sin(2 * math.pi / steps), math.cos(2 * math.pi / steps), steps / (2 * math.pi)
    normals = []
    for i in range(n):
        dx, dy = path[(i + 1) % n][0] - path[i][0], path[(i + 1) % n][1] - path[i][1]
        f = 1.0 / math.hypot(dx, dy)
        normals.append((dy * f, -dx * f))
    out = []
    k = n - 1
    for j in range(n):
        px, py = path[j]
        add = lambda nx, ny: out.append((rnd(px + nx * delta), rnd(py + ny * delta)))
        sin_a = normals[k][0] * normals[j][1] - normals[j][0] * normals[k][1]
        cos_a = normals[k][0] * normals[j][0] + normals[k][1] * normals[j][1]
        if abs(sin_a * delta) < 1.0 and cos_a > 0:
            add(*normals[k])
        else:
            if abs(sin_a * delta) >= 1.0:
                sin_a = max(-1.0, min(1.0, sin_a))
            if sin_a * delta < 0:
                add(*normals[k])
                out.append((px, py))
                add(*normals[j])
            else:
                a = math.atan2(sin_a, cos_a)
                st = max(rnd(spr * abs(a)), 1)
                X, Y = normals[k]
                for _ in range(st):
                    add(X, Y)
                    X, Y = X * m_cos - m_sin * Y, X * m_sin + Y * m_cos
                add(*normals[j])
        k = j
    return np.array(out, dtype=np.float64)


def _contours_as_regions(bitmap: np.ndarray):
    """[(start key, border points [n,2] (x,y), filled mask)] for every border cv2.findContours(RETR_LIST) would trace."""
    fg = bitmap.astype(bool)
    H, W = fg.shape
    regions = []
    lab, n = ndimage.label(fg, structure=np.ones((3, 3)))
    for k in range(1, n + 1):
        comp = lab == k
        ys, xs = np.nonzero(comp)
        regions.append(((ys[0], xs[0]), np.stack([xs, ys], 1), ndimage.binary_fill_holes(comp)))
    blab, bn = ndimage.label(~fg)  # 4-connected background
    edge = set(np.unique(np.concatenate([blab[0], blab[-1], blab[:, 0], blab[:, -1]]))) - {0}
    cross4 = ndimage.generate_binary_structure(2, 1)
    for k in range(1, bn + 1):
        if k in edge:
            continue
        hole = blab == k
        ring = ndimage.binary_dilation(hole, structure=cross4) & fg
        ys, xs = np.nonzero(ring)
        hy, hx = np.nonzero(hole)
        # a raster scan meets the hole border at the foreground pixel just left of the hole's first pixel
        filled = ndimage.binary_fill_holes(ring | hole)
        regions.append(((hy[0], hx[0] - 1), np.stack([xs, ys], 1), filled))
    regions.sort(key=lambda r: r[0])
    return regions[::-1]


def boxes_from_bitmap(pred: np.ndarray, thresh: float, dest_width: int, dest_height: int, *, unclip_ratio: float, min_sside: float,
                      box_thresh: float = 0.0, min_sside_out: float = 0.0, roll_start: bool = False, max_candidates: int = 1000):
    bitmap = pred > thresh
    H, W = bitmap.shape
    regions = _contours_as_regions(bitmap)[:max_candidates]
    boxes = np.zeros((len(regions), 4, 2), dtype=np.int64)
    scores = np.zeros(len(regions), dtype=np.float32)
    for idx, (_, pts, filled) in enumerate(regions):
        box, sside = min_area_rect(pts)
        if sside < min_sside:
            continue
        score = float(pred[filled].astype(np.float64).mean())
        if box_thresh > score:
            continue
        b = box.astype(np.float64)
        area = abs(sum(b[i][0] * b[(i + 1) % 4][1] - b[(i + 1) % 4][0] * b[i][1] for i in range(4))) / 2  # shapely Polygon.area
        length = sum(math.hypot(*(b[(i + 1) % 4] - b[i])) for i in range(4))                              # .length
        exp = clipper_offset_round(box, area * unclip_ratio / length)
        if len(exp) == 0:
            continue
        ebox, esside = min_area_rect(exp)
        if esside < min_sside_out:
            continue
        ebox[:, 0] = np.clip(np.round(ebox[:, 0] / W * dest_width), 0, dest_width)
        ebox[:, 1] = np.clip(np.round(ebox[:, 1] / H * dest_height), 0, dest_height)
        if roll_start:
            ebox = np.roll(ebox, 4 - int(ebox.sum(axis=1).argmin()), 0)
        boxes[idx] = ebox.astype(np.int64)
        scores[idx] = score
    return boxes, scores
