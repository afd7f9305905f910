"""TEST INFRASTRUCTURE (oracle) — border following and polygon fill, restated so that the reference's OWN box-extraction Python
(SegDetectorRepresenter, detection/ctd_utils/utils/db_utils.py:127-216 and detection/default_utils/dbnet_utils.py:97-190) can be
executed here with stand-ins for cv2 / pyclipper / shapely (oracle/ref_import.segdet).  Independent of csrc/hostglue.hip.

* ``find_contours_list`` — cv2.findContours(img, RETR_LIST, CHAIN_APPROX_NONE-equivalent): Suzuki & Abe's border following
  (CVGIP 1985, Algorithm 1) over the zero-padded image: every outer border of an 8-connected component and every hole border, as
  the sequence of border pixels, listed last-found first like OpenCV's contour list.  (CHAIN_APPROX_SIMPLE only drops the interior
  points of straight runs: hull, min-area rectangle and fill of the polygon are unchanged.)
* ``fill_poly`` — cv2.fillPoly of such a contour: the border pixels plus every pixel whose centre the closed polygon encloses
  (even-odd rule).
Parity with the real OpenCV is unpinned (it is installed nowhere this runs); scripts/make_cv2_pins.py records what would pin it."""
from __future__ import annotations

from typing import List

import numpy as np

# 8-neighbourhood in clockwise order starting at "west" (row, col offsets)
_CW = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]


def find_contours_list(bitmap: np.ndarray) -> List[np.ndarray]:
    """-> list of int32 arrays [n, 1, 2] (x, y), OpenCV's order (the border found last comes first)."""
    H, W = bitmap.shape
    f = np.zeros((H + 2, W + 2), dtype=np.int64)
    f[1:-1, 1:-1] = (np.asarray(bitmap) != 0).astype(np.int64)
    nbd = 1
    out = []
    for i in range(1, H + 1):
        for j in range(1, W + 1):
            if f[i, j] == 0:
                continue
            if f[i, j] == 1 and f[i, j - 1] == 0:      # outer border
                start = 0                               # neighbour index of (i, j - 1) in _CW
            elif f[i, j] >= 1 and f[i, j + 1] == 0:    # hole border
                start = 4
            else:
                continue
            nbd += 1
            # (3.1) clockwise from the start neighbour: first non-zero pixel
            k1 = None
            for s in range(8):
                k = (start + s) % 8
                if f[i + _CW[k][0], j + _CW[k][1]] != 0:
                    k1 = k
                    break
            if k1 is None:                              # isolated pixel
                f[i, j] = -nbd
                out.append(np.array([[[j - 1, i - 1]]], dtype=np.int32))
                continue
            i1, j1 = i + _CW[k1][0], j + _CW[k1][1]
            i2, j2, i3, j3 = i1, j1, i, j
            pts = []
            while True:
                # (3.3) counter-clockwise around (i3, j3), starting after (i2, j2)
                k2 = _CW.index((i2 - i3, j2 - j3))
                east_zero = False
                for s in range(1, 9):
                    k = (k2 - s) % 8
                    y, x = i3 + _CW[k][0], j3 + _CW[k][1]
                    if f[y, x] != 0:
                        i4, j4 = y, x
                        break
                    if k == 4:                          # (i3, j3 + 1) examined and found zero
                        east_zero = True
                # (3.4)
                if east_zero:
                    f[i3, j3] = -nbd
                elif f[i3, j3] == 1:
                    f[i3, j3] = nbd
                pts.append((j3 - 1, i3 - 1))
                # (3.5)
                if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
                    break
                i2, j2, i3, j3 = i3, j3, i4, j4
            out.append(np.array(pts, dtype=np.int32).reshape(-1, 1, 2))
    return out[::-1]


def fill_poly(mask: np.ndarray, pts: np.ndarray, value=1) -> np.ndarray:
    """cv2.fillPoly(mask, [pts], value) for a closed pixel contour: border pixels + enclosed pixel centres (even-odd)."""
    p = np.asarray(pts).reshape(-1, 2).astype(np.int64)
    H, W = mask.shape
    n = len(p)
    # border: the segments between consecutive vertices (8-connected steps for traced contours; straight runs otherwise)
    for a in range(n):
        (x0, y0), (x1, y1) = p[a], p[(a + 1) % n]
        steps = int(max(abs(x1 - x0), abs(y1 - y0), 1))
        for t in range(steps + 1):
            x = int(round(x0 + (x1 - x0) * t / steps))
            y = int(round(y0 + (y1 - y0) * t / steps))
            if 0 <= x < W and 0 <= y < H:
                mask[y, x] = value
    if n >= 3:
        ys = np.arange(H)[:, None] + 0.0
        xs = np.arange(W)[None, :] + 0.0
        inside = np.zeros((H, W), dtype=bool)
        for a in range(n):
            (x0, y0), (x1, y1) = p[a].astype(np.float64), p[(a + 1) % n].astype(np.float64)
            if y0 == y1:
                continue
            cond = (y0 > ys) != (y1 > ys)
            xint = (x1 - x0) * (ys - y0) / (y1 - y0) + x0
            inside ^= cond & (xs < xint)
        mask[inside] = value
    return mask
