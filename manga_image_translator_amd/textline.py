"""Text-line geometry on the host: the ``Quadrilateral`` type that crosses every stage boundary and the
plan for rectifying a line into a 48-px-high crop.

Mirror of the reference's ``sort_pnts`` / ``Quadrilateral`` (/root/reference/manga_translator/utils/generic.py:324-443)
and of the geometry half of ``Quadrilateral.get_transformed_region`` (:445-481).  The pixel half of that function
(cv2.warpPerspective + cv2.rotate) runs on the GPU: ``warp_plan`` only produces the crop rectangle, the destination
size and the inverse homography that ``mit_ocr_warp_lines`` (csrc/ocr_warp.hip) consumes.
"""
from __future__ import annotations

import functools
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np


def sort_pnts(pts: np.ndarray) -> Tuple[np.ndarray, bool]:
    """Canonical corner order [tl, tr, br, bl] and the vertical flag (generic.py:324-354)."""
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
    def area(self) -> float:
        x, y = self.pts[:, 0].astype(np.float64), self.pts[:, 1].astype(np.float64)
        return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)  # convex quad: hull area = shoelace

    def clip(self, width: int, height: int) -> None:
        self.pts[:, 0] = np.clip(np.round(self.pts[:, 0]), 0, width)
        self.pts[:, 1] = np.clip(np.round(self.pts[:, 1]), 0, height)


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
