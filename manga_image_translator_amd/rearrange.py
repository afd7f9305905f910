"""Tall / wide page tiling for the detectors ("rearrange"): what the reference's ``det_rearrange_forward`` does
(/root/reference/manga_translator/utils/generic.py:876-997, called from detection/ctd.py:137 and detection/default.py:60).

A webtoon strip letterboxed to the detect size would lose its text, so when ``long side / tgt_size > 2.5`` and the aspect ratio
exceeds 3 the reference cuts the strip into overlapping bands of ``pw_num`` strip-widths each, lays ``pw_num`` bands side by
side into a square, pads / shrinks that square to ``tgt_size`` (``square_pad_resize``, :848-874), runs the detector network on
batches of at most four squares, and stitches the output maps back (bands averaged where consecutive ones overlap).

This module re-derives that geometry once (``plan``) and applies it in two vectorised passes (``squares`` before the network,
``stitch`` after it); the network itself is whatever batched callable the plugin hands in (the HIP engine).  The arithmetic of the
stitch (accumulate, halve the overlap of every band after the first) is kept in float32 numpy in the reference's order, so the
stitched maps are bit-identical to the reference's for the same network output (tests/golden/rearrange.npz pins that against the
reference function itself, with a deterministic stand-in for the network).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Plan:
    transpose: bool      # the strip is wide, not tall: work on the transposed page
    h: int               # strip length (after the transpose)
    w: int               # strip width
    pw_num: int          # bands laid side by side in one square
    patch: int           # band length = square side before padding = pw_num * w
    ph_num: int          # number of bands
    ph_step: int         # band pitch along the strip (consecutive bands overlap by patch - ph_step)
    rel_steps: List[float]  # band start / strip length
    p_num: int           # number of squares
    pad_num: int         # empty bands appended to fill the last square


def plan(height: int, width: int, tgt_size: int) -> Optional[Plan]:
    """The tiling of an [height, width] page for a detector running at ``tgt_size``, or None when the page is processed whole
    (generic.py:941-949: rearrange only when long/tgt > 2.5 and long/short > 3)."""
    transpose = height < width
    h, w = (width, height) if transpose else (height, width)
    if not (h / tgt_size > 2.5 and h / w > 3):
        return None
    pw_num = max(int(np.floor(2 * tgt_size / w)), 2)
    patch = pw_num * w
    ph_num = int(np.ceil(h / patch))
    ph_step = int((h - patch) / (ph_num - 1)) if ph_num > 1 else 0
    rel = [(i * ph_step) / h for i in range(ph_num)]
    p_num = int(np.ceil(ph_num / pw_num))
    return Plan(transpose, h, w, pw_num, patch, ph_num, ph_step, rel, p_num, p_num * pw_num - ph_num)


def squares(img: np.ndarray, pl: Plan, tgt_size: int, resize: Optional[Callable[[np.ndarray, Tuple[int, int]], np.ndarray]] = None):
    """u8 page [H,W,3] -> (squares u8 [p_num, tgt, tgt, 3], pad_size): bands of ``patch`` rows cut at pitch ph_step (the last one
    ends at or before the strip end by construction of ph_step; empty bands fill the last square), ``pw_num`` bands side by side
    per square, zero padding at the bottom / right up to ``tgt_size`` or — for squares larger than ``tgt_size`` —
    an INTER_LINEAR shrink (``resize(img, (w, h))``; with ``resize=None`` the squares are returned unshrunk, [p_num, side, side, 3],
    for a caller that shrinks them on the GPU)."""
    strip = np.transpose(img, (1, 0, 2)) if pl.transpose else img
    C = strip.shape[2]
    side = pl.patch
    out = []
    pad_size = None
    for s in range(pl.p_num):
        sq = np.zeros((side, side, C), dtype=strip.dtype)
        for j in range(pl.pw_num):
            b = s * pl.pw_num + j
            if b >= pl.ph_num:
                break
            t = b * pl.ph_step
            band = strip[t:t + side]
            assert band.shape[0] == side, "ph_step keeps every band inside the strip"
            if pl.transpose:   # '(p pw_num) ph pw c -> p (pw_num pw) ph c': bands stacked along rows, each one transposed
                sq[j * pl.w:(j + 1) * pl.w, :, :] = np.transpose(band, (1, 0, 2))
            else:              # '(p pw_num) ph pw c -> p ph (pw_num pw) c': bands side by side
                sq[:, j * pl.w:(j + 1) * pl.w, :] = band
        pad = max(tgt_size - side, 0)   # square_pad_resize: the input is already square, so pad_h == pad_w == tgt - side
        if pad:
            sq = np.pad(sq, ((0, pad), (0, pad), (0, 0)))
        if sq.shape[0] > tgt_size and resize is not None:
            sq = resize(sq, (tgt_size, tgt_size))
        if pad_size is None:
            pad_size = pad
        out.append(sq)
    return np.stack(out), int(pad_size)


def stitch(maps: Sequence[np.ndarray], pl: Plan, channel: int) -> np.ndarray:
    """Network output squares [c, s, s] (already cropped of the padding) -> the strip's map [1, c, H', W'] (float32).

    Every band is added at its relative position; from the second band on, the rows it shares with its predecessor (the first
    ``s - step`` rows of the band) are halved after the add — the reference's running average (generic.py:897-916)."""
    psize = maps[0].shape[-1]
    step = int(pl.ph_step * psize / pl.patch)
    pw = int(psize / pl.pw_num)
    hh = int(pw / pl.w * pl.h)
    tgt = np.zeros((channel, hh, pw), dtype=np.float32)
    last = len(maps) * pl.pw_num - pl.pad_num - 1
    done = False
    for ii, p in enumerate(maps):
        if done:
            break
        if pl.transpose:
            p = np.transpose(p, (0, 2, 1))
        for jj in range(pl.pw_num):
            pidx = ii * pl.pw_num + jj
            t = int(round(pl.rel_steps[pidx] * hh))
            b = min(t + psize, hh)
            tgt[..., t:b, :] += p[..., :b - t, jj * pw:(jj + 1) * pw]
            if pidx > 0:
                tgt[..., t:t + (psize - step), :] /= 2.0
            if pidx >= last:
                done = True
                break
    if pl.transpose:
        tgt = np.transpose(tgt, (0, 2, 1))
    return tgt[None]


def forward(img: np.ndarray, batch_forward: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]], tgt_size: int,
            resize: Optional[Callable[[np.ndarray, Tuple[int, int]], np.ndarray]] = None, max_batch_size: int = 4):
    """det_rearrange_forward: (db [1,2,H',W'], mask [1,1,H'',W'']) or (None, None) when the page needs no tiling.
    ``batch_forward(u8 [n<=4, s, s, 3]) -> (db [n,2,m,m], mask [n,1,m',m'])`` float32 numpy, s = tgt_size (or the unshrunk
    square side when ``resize`` is None and the squares exceed tgt_size: the callable then shrinks them itself)."""
    pl = plan(img.shape[0], img.shape[1], tgt_size)
    if pl is None:
        return None, None
    sq, pad_size = squares(img, pl, tgt_size, resize)
    dbs, masks = [], []
    for i in range(0, len(sq), max_batch_size):
        db, mask = batch_forward(sq[i:i + max_batch_size])
        for d, m in zip(db, mask):
            if pad_size > 0:
                pd, pm = int(db.shape[-1] / tgt_size * pad_size), int(mask.shape[-1] / tgt_size * pad_size)
                d, m = d[..., :-pd, :-pd], m[..., :-pm, :-pm]
            dbs.append(d)
            masks.append(m)
    return stitch(dbs, pl, 2), stitch(masks, pl, 1)
