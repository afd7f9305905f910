"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the text-line rectification that feeds the OCR stage.

Restates ``Quadrilateral.get_transformed_region`` (/root/reference/manga_translator/utils/generic.py:445-481) and the
OpenCV routines it calls, in numpy.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: **unpinned** for the OpenCV half.  cv2 is installed neither here nor on the GPU box, so
``cv2.findHomography`` / ``cv2.warpPerspective`` / ``cv2.rotate`` are restated from OpenCV 4.x's imgproc sources
(imgwarp.cpp: WarpPerspectiveInvoker, remapBilinear<FixedPtCast<int, uchar, 15>>, initInterTab2D): coordinates are
evaluated in double per 128-column block, scaled by INTER_TAB_SIZE = 32 and rounded half-to-even; 5 fractional bits pick
bilinear weights that are exact integers on the 1/32 grid; BORDER_CONSTANT (0).  The geometry half (sort order,
structure vectors, ratio, destination size) is checked against the reference's own ``sort_pnts``/``Quadrilateral``
code imported from /root/reference in tests/test_textline.py when that tree is present.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def sort_pnts(pts: np.ndarray) -> Tuple[np.ndarray, bool]:
    """generic.py:324-354, restated with explicit loops instead of the fancy-index one-liners."""
    pts = np.asarray(pts)
    assert pts.shape == (4, 2)
    vecs = np.array([pts[i] - pts[j] for i in range(4) for j in range(4)])
    order = np.argsort(np.linalg.norm(vecs, axis=1))
    a, b = vecs[order[8]].copy(), vecs[order[10]].copy()
    if (a * b).sum() < 0:
        a = -a
    struc = np.abs((a + b) / 2)
    vertical = bool(struc[0] <= struc[1])
    if vertical:
        p = pts[np.argsort(pts[:, 1])]
        top = p[:2][np.argsort(p[:2, 0])]
        bot = p[2:][np.argsort(p[2:, 0])[::-1]]
        return np.concatenate([top, bot]), vertical
    p = pts[np.argsort(pts[:, 0])]
    left = sorted(p[:2], key=lambda q: q[1])
    right = sorted(p[2:], key=lambda q: q[1])
    return np.array([left[0], right[0], right[1], left[1]]), vertical


def structure(pts: np.ndarray):
    """Quadrilateral.structure (generic.py:379-385)."""
    return [((pts[0] + pts[1]) / 2).astype(int), ((pts[2] + pts[3]) / 2).astype(int),
            ((pts[1] + pts[2]) / 2).astype(int), ((pts[3] + pts[0]) / 2).astype(int)]


def find_homography_4pt(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.findHomography with four correspondences: the unique H with h33 = 1, from the 8x8 linear system
    (the formulation of cv2.getPerspectiveTransform; with 4 points RANSAC has one candidate and all points are inliers).

    Pixels whose source coordinate falls exactly on a 1/32 rounding tie (e.g. the last column of an axis-aligned box,
    which maps exactly onto the crop's right edge) depend on the last bits of H: another solver (SVD/DLT, OpenCV's
    normalised LM refinement) moves ~1e-13 and can flip them.  tests/test_textline.py bounds that sensitivity."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        (x, y), (u, v) = src[i], dst[i]
        A[2 * i] = [x, y, 1, 0, 0, 0, -u * x, -u * y]
        A[2 * i + 1] = [0, 0, 0, x, y, 1, -v * x, -v * y]
        b[2 * i], b[2 * i + 1] = u, v
    return np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)


def find_homography_4pt_dlt(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """Same H through the DLT null vector (SVD) — used only to measure the tie sensitivity described above."""
    rows = []
    for (x, y), (u, v) in zip(np.asarray(src, np.float64), np.asarray(dst, np.float64)):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, vt = np.linalg.svd(np.array(rows))
    h = vt[-1]
    return (h / h[8]).reshape(3, 3)


def warp_perspective_u8(src: np.ndarray, M: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.warpPerspective(src, M, (w, h)) for 8-bit images: INTER_LINEAR, BORDER_CONSTANT 0 (the defaults)."""
    w, h = dsize
    ch, cw = src.shape[:2]
    Mi = np.linalg.inv(np.asarray(M, np.float64)).reshape(-1)
    bh0 = min(32, h)
    bw0 = min(4096 // bh0, w)
    ys, xs = np.mgrid[0:h, 0:w]
    xb = (xs // bw0) * bw0
    x1 = xs - xb
    X0 = Mi[0] * xb + Mi[1] * ys + Mi[2]
    Y0 = Mi[3] * xb + Mi[4] * ys + Mi[5]
    W0 = Mi[6] * xb + Mi[7] * ys + Mi[8]
    Wd = W0 + Mi[6] * x1
    with np.errstate(divide="ignore"):
        Wd = np.where(Wd != 0, INTER_TAB_SIZE / Wd, 0.0)
    lim = lambda v: np.maximum(-2147483648.0, np.minimum(2147483647.0, v))
    X = np.rint(lim((X0 + Mi[0] * x1) * Wd)).astype(np.int64)
    Y = np.rint(lim((Y0 + Mi[3] * x1) * Wd)).astype(np.int64)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    ax, ay = X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)
    wts = [(32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32]
    acc = np.zeros((h, w, src.shape[2]), dtype=np.int64)
    for (dy, dx), wt in zip(((0, 0), (0, 1), (1, 0), (1, 1)), wts):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < ch) & (xx >= 0) & (xx < cw)
        v = src[np.clip(yy, 0, ch - 1), np.clip(xx, 0, cw - 1)].astype(np.int64)
        acc += np.where(ok[..., None], v, 0) * wt[..., None]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def get_transformed_region(img: np.ndarray, quad_pts: np.ndarray, direction: str, textheight: int = 48) -> np.ndarray:
    """Quadrilateral.get_transformed_region (generic.py:445-481); ``quad_pts`` are the already-sorted ``self.pts``."""
    l1a, l1b, l2a, l2b = [a.astype(np.float32) for a in structure(quad_pts)]
    ratio = np.linalg.norm(l1b - l1a) / np.linalg.norm(l2b - l2a)
    src_pts = quad_pts.astype(np.int64).copy()
    im_h, im_w = img.shape[:2]
    x1, y1, x2, y2 = src_pts[:, 0].min(), src_pts[:, 1].min(), src_pts[:, 0].max(), src_pts[:, 1].max()
    x1, x2 = np.clip(x1, 0, im_w), np.clip(x2, 0, im_w)
    y1, y2 = np.clip(y1, 0, im_h), np.clip(y2, 0, im_h)
    crop = img[y1:y2, x1:x2]
    src_pts[:, 0] -= x1
    src_pts[:, 1] -= y1
    if direction == "h":
        h = max(int(textheight), 2)
        w = max(int(round(textheight / ratio)), 2)
    else:
        w = max(int(textheight), 2)
        h = max(int(round(textheight * ratio)), 2)
    dst_pts = np.array([[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], dtype=np.float32)
    M = find_homography_4pt(src_pts, dst_pts)
    region = warp_perspective_u8(crop, M, (w, h))
    if direction == "v":
        region = np.rot90(region, 1)  # cv2.ROTATE_90_COUNTERCLOCKWISE
    return np.ascontiguousarray(region)
