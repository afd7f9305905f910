"""TEST INFRASTRUCTURE — CPU restatement of cv2.resize(..., INTER_LINEAR_EXACT) on 8-bit images (parity unpinned against the real
OpenCV, which is installed nowhere this runs; what it pins is the product's table-driven kernel / numpy twin against an
independent statement of the same published algorithm).

OpenCV imgproc/src/resize.cpp, ``resize_bitExact<uchar, interpolationLinear<uchar>>`` with fixed-point type ufixedpoint16
(8 fractional bits, fixedpoint.inl.hpp):
  * position of destination sample d along an axis:  f = (1 / (n_dst / n_src)) * (d + 0.5) - 0.5  in IEEE double (softdouble)
  * i = floor(f); if i < 0 the sample copies source 0, if i >= n_src - 1 (or n_src == 1) it copies source n_src - 1; otherwise
    weights  w1 = cvRound((f - i) * 256)  (round half to even),  w0 = 256 - w1
  * horizontal pass in 16-bit 8.8 fixed point:  h = w0 * S[i] + w1 * S[i + 1]         (copies are S << 8)
  * vertical pass in 32-bit 16.16 fixed point:  v = (u0 * h0 + u1 * h1 + 32768) >> 16, saturated to 255
  * an exact 2x shrink in both directions (non 2-channel) is routed to the INTER_AREA box mean (a + b + c + d + 2) >> 2.
Written with plain Python integers, one sample at a time, on purpose: no code shared with manga_image_translator_amd/imgproc.py.
Reference call site: resize_keep_aspect, /root/reference/manga_translator/utils/generic.py:251-255.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np


def _round_half_even(x: float) -> int:
    fl = math.floor(x)
    d = x - fl
    if d > 0.5 or (d == 0.5 and fl % 2 == 1):
        return fl + 1
    return fl


def _axis(n_src: int, n_dst: int) -> List[Tuple[int, int, int]]:
    """(first source index, w0, w1) per destination index; a copy is (index, 256, 0)."""
    inv_scale = n_dst / n_src
    scale = 1.0 / inv_scale
    out = []
    for d in range(n_dst):
        f = scale * (d + 0.5) - 0.5
        i = math.floor(f)
        if i < 0 or n_src <= 1:
            out.append((0, 256, 0))
        elif i >= n_src - 1:
            out.append((n_src - 1, 256, 0))
        else:
            w1 = _round_half_even((f - i) * 256.0)
            out.append((i, 256 - w1, w1))
    return out


def resize_linear_exact_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    dw, dh = int(dsize[0]), int(dsize[1])
    squeeze = src.ndim == 2
    s = src[..., None] if squeeze else src
    sh, sw, cn = s.shape
    if sh == 2 * dh and sw == 2 * dw and cn != 2:
        t = s.astype(np.int64)
        out = ((t[0::2, 0::2] + t[0::2, 1::2] + t[1::2, 0::2] + t[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        return out[..., 0] if squeeze else out
    xs, ys = _axis(sw, dw), _axis(sh, dh)
    # horizontal pass, row by row (8.8 fixed point)
    hbuf = np.zeros((sh, dw, cn), dtype=np.int64)
    t = s.astype(np.int64)
    for x, (i, w0, w1) in enumerate(xs):
        hbuf[:, x] = w0 * t[:, i] + (w1 * t[:, min(i + 1, sw - 1)] if w1 else 0)
    out = np.zeros((dh, dw, cn), dtype=np.uint8)
    for y, (i, u0, u1) in enumerate(ys):
        v = u0 * hbuf[i] + (u1 * hbuf[min(i + 1, sh - 1)] if u1 else 0)
        out[y] = np.minimum((v + 32768) >> 16, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def resize_keep_aspect(img: np.ndarray, size: int) -> np.ndarray:
    """utils/generic.py:251-255 with the restated INTER_LINEAR_EXACT."""
    ratio = float(size) / max(img.shape[0], img.shape[1])
    return resize_linear_exact_u8(img, (round(img.shape[1] * ratio), round(img.shape[0] * ratio)))
