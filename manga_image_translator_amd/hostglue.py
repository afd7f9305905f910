"""Glue between the detector networks and the text lines, on the native routines of libmit_hip.so: the GPU chain
(``boxes_from_bitmap_gpu`` / ``ctd_boxes_gpu`` / ``dbnet_boxes_gpu`` -> csrc/ctd_boxes.hip, what the plugins and the coupled engine call
for maps that are on the device) and its host form (``boxes_from_bitmap`` -> csrc/hostglue.hip: injected or host-assembled maps, and the
fallback for a border that does not fit a workgroup's LDS).

``boxes_from_bitmap`` stands in for ``SegDetectorRepresenter.__call__`` of the reference
(/root/reference/manga_translator/detection/ctd_utils/utils/db_utils.py:40-171 for ``ctd``,
detection/default_utils/dbnet_utils.py:16-144 for ``default``), which runs on OpenCV + pyclipper + shapely.  The native
code restates those libraries' published algorithms (csrc/hostglue.hip); parity with the real libraries is unpinned
because none of them is installed where this repo is built or tested.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
from scipy import ndimage

from . import lib as _lib


def boxes_from_bitmap(pred: np.ndarray, thresh: float, dest_width: int, dest_height: int, *, unclip_ratio: float,
                      min_sside: float, box_thresh: float = 0.0, min_sside_out: float = 0.0, roll_start: bool = False,
                      max_candidates: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    """pred f32 [H,W] -> (boxes int64 [n,4,2], scores f32 [n]), one slot per contour like the reference (zeros = skipped)."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    if pred.ndim != 2:
        raise ValueError(f"boxes_from_bitmap expects a 2-D map, got shape {pred.shape}")
    bitmap = np.ascontiguousarray(pred > thresh, dtype=np.uint8)  # binarize (db_utils.py:75)
    H, W = pred.shape
    boxes = np.zeros((max_candidates, 4, 2), dtype=np.int64)
    scores = np.zeros(max_candidates, dtype=np.float32)
    n = C.c_int(0)
    lib = _lib.load()
    _lib.check(lib.mit_boxes_from_bitmap(pred.ctypes.data, bitmap.ctypes.data, H, W, int(dest_width), int(dest_height), max_candidates,
                                         float(unclip_ratio), float(min_sside), float(box_thresh), float(min_sside_out), int(roll_start),
                                         boxes.ctypes.data, scores.ctypes.data, C.byref(n)), "mit_boxes_from_bitmap")
    return boxes[:n.value], scores[:n.value]


def ctd_boxes(lines_map: np.ndarray, im_h: int, im_w: int) -> Tuple[np.ndarray, np.ndarray]:
    """ComicTextDetector's call: SegDetectorRepresenter(thresh=0.3) on lines_map[:, 0] (ctd.py:102,156; unclip 1.5, sside >= 2)."""
    return boxes_from_bitmap(lines_map[0, 0], 0.3, im_w, im_h, unclip_ratio=1.5, min_sside=2.0)


def dbnet_boxes(db: np.ndarray, h: int, w: int, text_threshold: float, box_threshold: float, unclip_ratio: float):
    """DefaultDetector's call: dbnet_utils.SegDetectorRepresenter(text_threshold, box_threshold, unclip_ratio) on db[:, 0]
    (default.py:73-77; min_size 3, expanded short side >= 5, corners rolled to start at the smallest x + y)."""
    return boxes_from_bitmap(db[0, 0], text_threshold, w, h, unclip_ratio=unclip_ratio, min_sside=3.0, box_thresh=box_threshold,
                             min_sside_out=5.0, roll_start=True)


def boxes_from_bitmap_gpu_launch(pred, thresh: float, dest_width: int, dest_height: int, *, unclip_ratio: float, min_sside: float, box_thresh: float = 0.0,
                                 min_sside_out: float = 0.0, roll_start: bool = False, max_candidates: int = 1000):
    """Enqueue ``boxes_from_bitmap`` for a batch of maps that are ON THE DEVICE (csrc/ctd_boxes.hip: labelling, border walk, minAreaRect,
    score, round-join offset — all on the GPU) on the current stream; nothing is waited for.  pred f32 [B,H,W] (any page stride, rows
    dense: ``lines[:, 0]`` of an NCHW map as it is).  -> a handle for ``boxes_from_bitmap_gpu_collect``."""
    import torch

    from . import ops

    if pred.dim() != 3 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError(f"boxes_from_bitmap_gpu expects a float32 CUDA tensor [B,H,W], got {pred.dtype} {tuple(pred.shape)} on {pred.device}")
    B, H, W = pred.shape
    if pred.stride(2) != 1 or pred.stride(1) != W:
        pred = pred.contiguous()
    lib = _lib.load()
    dev = pred.device
    ws_bytes = int(lib.mit_boxes_from_bitmap_dev_workspace_bytes(B, H, W, max_candidates))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    boxes = torch.empty(B, max_candidates, 4, 2, dtype=torch.int64, device=dev)
    scores = torch.empty(B, max_candidates, dtype=torch.float32, device=dev)
    meta = torch.empty(2, B, dtype=torch.int32, device=dev)   # counts, overflow flags
    _lib.check(lib.mit_boxes_from_bitmap_dev(pred.data_ptr(), pred.stride(0), None, 0, float(thresh), B, H, W, int(dest_width), int(dest_height),
                                             max_candidates, float(unclip_ratio), float(min_sside), float(box_thresh), float(min_sside_out), int(roll_start),
                                             ws.data_ptr(), ws_bytes, boxes.data_ptr(), scores.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(),
                                             C.c_void_p(ops.current_stream())), "mit_boxes_from_bitmap_dev")
    meta_h = torch.empty(meta.shape, dtype=torch.int32, pin_memory=True)
    meta_h.copy_(meta, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    kw = dict(unclip_ratio=unclip_ratio, min_sside=min_sside, box_thresh=box_thresh, min_sside_out=min_sside_out, roll_start=roll_start,
              max_candidates=max_candidates)
    return dict(pred=pred, ws=ws, boxes=boxes, scores=scores, meta_h=meta_h, ev=ev, args=(thresh, dest_width, dest_height), kw=kw)


def boxes_from_bitmap_gpu_collect(h) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Wait for a launched extraction and bring the boxes over: per page (boxes int64 [n,4,2], scores f32 [n]), the arrays the host routine
    returns for that page.  A page with a border longer than a wave's LDS holds (8192 points) is computed by the host routine (same results)."""
    h["ev"].synchronize()
    meta_h, max_candidates = h["meta_h"], h["kw"]["max_candidates"]
    B = meta_h.shape[1]
    counts = np.minimum(meta_h[0].numpy(), max_candidates)
    kmax = int(counts.max()) if B else 0
    boxes_h = h["boxes"][:, :kmax].cpu().numpy() if kmax else np.zeros((B, 0, 4, 2), np.int64)
    scores_h = h["scores"][:, :kmax].cpu().numpy() if kmax else np.zeros((B, 0), np.float32)
    out = []
    for b in range(B):
        if int(meta_h[1, b]) != 0:    # a border that does not fit a wave's LDS: this page on the host routine
            out.append(boxes_from_bitmap(h["pred"][b].cpu().numpy(), *h["args"], **h["kw"]))
        else:
            n = int(counts[b])
            out.append((boxes_h[b, :n].copy(), scores_h[b, :n].copy()))
    return out


def boxes_from_bitmap_gpu(pred, thresh: float, dest_width: int, dest_height: int, **kw) -> List[Tuple[np.ndarray, np.ndarray]]:
    return boxes_from_bitmap_gpu_collect(boxes_from_bitmap_gpu_launch(pred, thresh, dest_width, dest_height, **kw))


def ctd_boxes_gpu(lines, im_h: int, im_w: int) -> List[Tuple[np.ndarray, np.ndarray]]:
    """``ctd_boxes`` for a batch on the device: lines f32 [B,2,h,w] (the network's output as it is) -> per page (boxes, scores)."""
    return boxes_from_bitmap_gpu(lines[:, 0], 0.3, im_w, im_h, unclip_ratio=1.5, min_sside=2.0)


def dbnet_boxes_gpu(db, h: int, w: int, text_threshold: float, box_threshold: float, unclip_ratio: float) -> List[Tuple[np.ndarray, np.ndarray]]:
    """``dbnet_boxes`` for a batch on the device: db f32 [B,C,H,W]."""
    return boxes_from_bitmap_gpu(db[:, 0], text_threshold, w, h, unclip_ratio=unclip_ratio, min_sside=3.0, box_thresh=box_threshold, min_sside_out=5.0,
                                 roll_start=True)


def contour_count(bitmap: np.ndarray) -> Tuple[int, int]:
    bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
    n, p = C.c_int(0), C.c_int64(0)
    _lib.check(_lib.load().mit_find_contours_count(bitmap.ctypes.data, bitmap.shape[0], bitmap.shape[1], C.byref(n), C.byref(p)))
    return n.value, p.value


# ---------------------------------------------------------------------------------------------------------------------
# Mask refinement of the ctd detector (ctd_utils/textmask.py:16-174) and the image resize around it, on numpy + scipy.
# The reference runs these through OpenCV; each primitive below states the OpenCV rule it follows (parity unpinned against
# the real library, pinned against the reference's own Python through tests/golden/refine_mask.npz).
# ---------------------------------------------------------------------------------------------------------------------

def resize_linear_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 [H,W] or [H,W,C]: an exact 2x shrink is the 2x2 box
    mean ((a+b+c+d+2)>>2); otherwise separable bilinear with 11-bit fixed-point coefficients, pixel centres aligned,
    vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2 (OpenCV resize.cpp, 8-bit path)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    squeeze = src.ndim == 2
    s = src[..., None] if squeeze else src
    sh, sw = s.shape[:2]
    if sh == 2 * dh and sw == 2 * dw:
        t = s.astype(np.int32)
        out = ((t[0::2, 0::2] + t[0::2, 1::2] + t[1::2, 0::2] + t[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        return out[..., 0] if squeeze else out

    def taps(n_src, n_dst):
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        lo, hi = i0 < 0, i0 >= n_src - 1
        i0 = np.where(lo, 0, np.where(hi, n_src - 1, i0))
        fr = np.where(lo | hi, np.float32(0), fr)
        c1 = np.rint(fr * np.float32(2048)).astype(np.int64)
        c0 = np.rint((np.float32(1) - fr) * np.float32(2048)).astype(np.int64)
        return i0, np.minimum(i0 + 1, n_src - 1), c0, c1

    y0, y1, yc0, yc1 = taps(sh, dh)
    x0, x1, xc0, xc1 = taps(sw, dw)
    t = s.astype(np.int64)
    rows = t[:, x0] * xc0[None, :, None] + t[:, x1] * xc1[None, :, None]
    out = (((yc0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((yc1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def enlarge_window(rect, im_w: int, im_h: int, ratio: float = 2.5, aspect_ratio: float = 1.0) -> List[int]:
    """ctd_utils/utils/imgproc_utils.py:134-150: grow the box so that its area is ``ratio`` x, clipped to the page."""
    x1, y1, x2, y2 = [int(v) for v in rect]
    w, h = x2 - x1, y2 - y1
    roots = np.roots([aspect_ratio, w + h * aspect_ratio, (1 - ratio) * w * h])
    roots.sort()
    delta = int(round(float(np.real(roots[-1])) / 2))
    delta_w = min(x1, im_w - x2, int(delta * aspect_ratio))
    delta = min(y1, im_h - y2, delta)
    return [x1 - delta_w, y1 - delta, x2 + delta_w, y2 + delta]


def _gray_bgr2gray(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, COLOR_BGR2GRAY) on 8 bits: (c0*1868 + c1*9617 + c2*4899 + 8192) >> 14 (the reference feeds RGB
    pages to the BGR conversion, textmask.py:57-58; the literal channel weights are kept)."""
    t = img.astype(np.int64)
    return ((t[..., 0] * 1868 + t[..., 1] * 9617 + t[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def _otsu_threshold(c: np.ndarray) -> int:
    """OpenCV getThreshVal_Otsu_8u: the grey level maximising the between-class variance (first maximum)."""
    hist = np.bincount(c.reshape(-1), minlength=256).astype(np.float64)
    scale = 1.0 / c.size
    mu = float((np.arange(256) * hist).sum()) * scale
    q1 = mu1 = 0.0
    best, best_sigma = 0, 0.0
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < 2.220446049250313e-16 or max(q1, q2) > 1.0 - 2.220446049250313e-16:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > best_sigma:
            best_sigma, best = sigma, i
    return best


_RECT3 = np.ones((3, 3), bool)
_CROSS3 = ndimage.generate_binary_structure(2, 1)  # cv2.getStructuringElement(MORPH_ELLIPSE, (3, 3))


def _erode(m: np.ndarray, fp: np.ndarray) -> np.ndarray:  # cv2.erode: outside the image counts as +inf
    return ndimage.minimum_filter(m, footprint=fp, mode="constant", cval=255)  # cv2: min over src(x + x' - anchor), anchor = k // 2 (no kernel reflection)


def _dilate(m: np.ndarray, fp: np.ndarray) -> np.ndarray:  # cv2.dilate: outside the image counts as -inf
    return ndimage.maximum_filter(m, footprint=fp, mode="constant", cval=0)


def _in_range_u8(src: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """cv2.inRange(src_u8, lo, hi) with scalar bounds: OpenCV converts the bounds to int32 with cvRound (half to even) and
    saturates them to the 8-bit range; an inverted or out-of-range interval selects nothing (core/src/arithm.cpp inRange)."""
    ilo, ihi = int(np.rint(lo)), int(np.rint(hi))
    if ilo > ihi or ilo > 255 or ihi < 0:
        return np.zeros(src.shape, np.uint8)
    ilo, ihi = max(ilo, 0), min(ihi, 255)
    return np.where((src >= ilo) & (src <= ihi), 255, 0).astype(np.uint8)


def _xor_sum(a: np.ndarray, b: np.ndarray) -> int:
    return int(np.bitwise_xor(a, b).sum(dtype=np.uint64))


def _minxor_thresh(threshed: np.ndarray, mask: np.ndarray):
    """textmask.py:29-42 (dilate=False): the candidate or its complement, whichever is closer to the predicted mask."""
    neg = 255 - threshed
    a, b = _xor_sum(neg, mask), _xor_sum(threshed, mask)
    return (neg, a) if a < b else (threshed, b)


def _topk_color(color_list, bins, k=3, color_var=10, bin_tol=0.001):
    """textmask.py:16-27."""
    idx = np.argsort(bins * -1)
    color_list, bins = color_list[idx], bins[idx]
    top = [color_list[0]]
    tol = np.sum(bins) * bin_tol
    if len(color_list) > 1:
        for color, b in zip(color_list[1:], bins[1:]):
            if np.abs(np.array(top) - color).min() > color_var:
                top.append(color)
            if len(top) >= k or b < tol:
                break
    return top


def _components(mask: np.ndarray, connectivity: int):
    """cv2.connectedComponentsWithStats(mask, connectivity): (n incl. background, labels, [(x, y, w, h, area)] per label)."""
    lab, n = ndimage.label(mask > 0, structure=np.ones((3, 3)) if connectivity == 8 else None)
    zy, zx = np.nonzero(lab == 0)  # label 0 = background: tight box of the zero pixels, like OpenCV
    stats = [(int(zx.min()), int(zy.min()), int(zx.max() - zx.min() + 1), int(zy.max() - zy.min() + 1), len(zx)) if len(zx) else (0, 0, 0, 0, 0)]
    for sl, k in zip(ndimage.find_objects(lab), range(1, n + 1)):
        ys, xs = sl
        stats.append((xs.start, ys.start, xs.stop - xs.start, ys.stop - ys.start, int((lab[sl] == k).sum())))
    return n + 1, lab, stats


def _merge_mask_list(mask_list, pred_mask: np.ndarray, inpaint_dilate: bool) -> np.ndarray:
    """textmask.py:74-132 (filter_with_lines False, pred_thresh 30) on the native routine (csrc/hostmask.hip)."""
    cands = np.ascontiguousarray(np.stack([np.where(m[0] > 0, 255, 0).astype(np.uint8) for m in mask_list]))
    scores = np.ascontiguousarray(np.array([int(m[1]) for m in mask_list], dtype=np.int64))
    pred = np.ascontiguousarray(pred_mask, dtype=np.uint8)
    h, w = pred.shape
    merged = np.empty((h, w), np.uint8)
    _lib.check(_lib.load().mit_merge_mask_list(cands.ctypes.data, scores.ctypes.data, len(mask_list), pred.ctypes.data, h, w,
                                               int(bool(inpaint_dilate)), merged.ctypes.data), "mit_merge_mask_list")
    return merged


def _merge_mask_list_numpy(mask_list, pred_mask: np.ndarray, inpaint_dilate: bool) -> np.ndarray:
    """The same in numpy / scipy (the statement the native routine is tested against; ~100x slower on a real text line)."""
    mask_list = sorted(mask_list, key=lambda x: x[1])
    pred_mask = _erode(pred_mask, _CROSS3)
    pred_mask = np.where(pred_mask > 60, 255, 0).astype(np.uint8)
    merged = np.zeros_like(pred_mask)

    def try_merge(lab, k, stat):
        x, y, w, h, _ = stat
        sl = (slice(y, y + h), slice(x, x + w))
        tmp = np.where(lab[sl] == k, 255, 0).astype(np.uint8) | merged[sl]
        if _xor_sum(tmp, pred_mask[sl]) < _xor_sum(merged[sl], pred_mask[sl]):
            merged[sl] = tmp

    for cand, _ in mask_list:
        n, lab, stats = _components(cand, 8)
        for k in range(1, n):
            if stats[k][2] * stats[k][3] < 3:
                continue
            try_merge(lab, k, stats[k])
    if inpaint_dilate:
        merged[...] = _dilate(merged, np.ones((5, 5), bool))
    # fill holes (:113-131): background components smaller than the second-largest area, when that lowers the XOR
    n, lab, stats = _components(255 - merged, 8)
    areas = np.sort(np.array([s[4] for s in stats]))
    thresh = areas[-2] if len(areas) > 1 else areas[-1]
    for k in range(n):  # label 0 (the mask itself) is visited too, like the reference; merging it is a no-op
        if stats[k][4] < thresh:
            try_merge(lab, k, stats[k])
    return merged


def refine_mask(img: np.ndarray, pred_mask: np.ndarray, quads: Sequence, refine_mode=None) -> np.ndarray:
    """ctd_utils/textmask.py:158-174: per text line, candidate masks from the 3 dominant grey levels and an Otsu split of
    the best colour channel, merged component by component where that brings the mask closer to the network's prediction.
    ``refine_mode`` 0 (REFINEMASK_INPAINT) adds the 5x5 dilation; the ctd detector passes None (ctd.py:177)."""
    out = np.zeros_like(pred_mask)
    for q in quads:
        bx1, by1, bx2, by2 = enlarge_window(q.xyxy, img.shape[1], img.shape[0])
        im = np.ascontiguousarray(img[by1:by2, bx1:bx2])
        msk = np.ascontiguousarray(pred_mask[by1:by2, bx1:bx2])
        if im.size == 0:
            continue
        grey = _gray_bgr2gray(im)
        cand = grey[_erode(msk, _RECT3) > 127]                   # get_topk_masklist :56-71
        bins, edges = np.histogram(cand, bins=255)
        masks = []
        for color in _topk_color(edges, bins, color_var=10, k=3):
            c_top = min(color + 30, 255)
            c_bottom = c_top - 60
            masks.append(list(_minxor_thresh(_in_range_u8(grey, c_bottom, c_top), msk)))
        otsu = []                                                 # get_otsuthresh_masklist :44-54 (per_channel False)
        for c in range(3):
            ch = im[..., c]
            otsu.append(list(_minxor_thresh(np.where(ch > _otsu_threshold(ch), 255, 0).astype(np.uint8), msk)))
        otsu.sort(key=lambda x: x[1])
        masks.append(otsu[0])
        out[by1:by2, bx1:bx2] |= _merge_mask_list(masks, msk, refine_mode == 0)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# refine_mask on the GPU (csrc/ctd_refine.hip): the pixel work of every text line of a page in a few batched launches; the
# per-line scalar decisions (numpy's own histogram binning, top-k colours, Otsu, candidate order) stay here, on 256-bin
# histograms and six xor sums per line.  Bit-identical to refine_mask above (tests/test_ctd_refine_gpu.py).
# ---------------------------------------------------------------------------------------------------------------------

def _in_range_bounds(lo: float, hi: float):
    """The integer interval cv2.inRange(src_u8, lo, hi) selects (see _in_range_u8), or None when it selects nothing."""
    ilo, ihi = int(np.rint(lo)), int(np.rint(hi))
    if ilo > ihi or ilo > 255 or ihi < 0:
        return None
    return max(ilo, 0), min(ihi, 255)


def refine_candidates(hist: np.ndarray):
    """Host half 1 of refine_mask_gpu: from the per-line histograms (n x 4 x 256: grey under the eroded mask, then the three
    channels) to the six raw candidates of every line as (kind, lo, hi, invert) — get_topk_masklist (textmask.py:56-71: numpy's own
    255-bin histogram of the grey levels, the top-k colours, +-30 in cv2.inRange's rounding) and the per-channel Otsu thresholds of
    get_otsuthresh_masklist (:44-54).  kind 0 = absent, 1 = inRange(grey, lo, hi), 2 + c = channel c > lo."""
    n = hist.shape[0]
    levels = np.arange(256, dtype=np.uint8)
    ch_hist = np.ascontiguousarray(hist[:, 1:4], dtype=np.int32)
    otsu = np.zeros((n, 3), dtype=np.int32)
    _lib.check(_lib.load().mit_otsu_from_hist(ch_hist.ctypes.data, 3 * n, otsu.ctypes.data), "mit_otsu_from_hist")
    out = []
    for i in range(n):
        g = hist[i, 0]
        present = g > 0
        bins_w, edges = np.histogram(levels[present], bins=255, weights=g[present].astype(np.float64))
        bins = np.rint(bins_w).astype(np.int64)   # the integer counts np.histogram(cand, bins=255) returns
        line = [(0, 0, 0, 0)] * 3
        for k, color in enumerate(_topk_color(edges, bins, color_var=10, k=3)):
            c_top = min(color + 30, 255)
            bounds = _in_range_bounds(c_top - 60, c_top)
            line[k] = (1, bounds[0], bounds[1], 0) if bounds else (1, 1, 0, 0)  # lo > hi: selects nothing
        out.append(line + [(2 + ch, int(otsu[i, ch]), 0, 0) for ch in range(3)])
    return out


def refine_merge_order(raw, sums: np.ndarray, sizes: Sequence[int]):
    """Host half 2: minxor_thresh (textmask.py:29-42) per candidate — the candidate or its complement, whichever byte-xor sum
    against the mask crop is smaller (``sums[i, k]`` = the candidate's, 255 * size - that = the complement's) — then the best Otsu
    channel and the merge order of the <= 4 survivors by score (stable, like sorted())."""
    out = []
    for i, line in enumerate(raw):
        total = 255 * int(sizes[i])
        picked = []
        for k, (kind, lo, hi, _) in enumerate(line):
            if kind == 0:
                picked.append(None)
                continue
            s_pos = int(sums[i, k])
            s_neg = total - s_pos
            picked.append((1, s_neg) if s_neg < s_pos else (0, s_pos))
        mask_list = [(k, picked[k]) for k in range(3) if picked[k] is not None]
        best_otsu = sorted([(k, picked[k]) for k in range(3, 6)], key=lambda t: t[1][1])[0]
        mask_list.append(best_otsu)
        mask_list.sort(key=lambda t: t[1][1])
        out.append([(line[k][0], line[k][1], line[k][2], inv) for k, (inv, _) in mask_list])
    return out


class _RefineWorkspace:
    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device):
        import torch

        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


class _PerThreadRefineWorkspace:
    """One scratch slab per host thread: a batch engine refines several pages at once from a thread pool (coupled.py)."""

    def __init__(self):
        import threading

        self._tls = threading.local()

    def get(self, nbytes: int, device):
        ws = getattr(self._tls, "ws", None)
        if ws is None:
            ws = self._tls.ws = _RefineWorkspace()
        return ws.get(nbytes, device)


_REFINE_WS = _PerThreadRefineWorkspace()


def refine_mask_gpu(page_dev, pred_dev, quads: Sequence, refine_mode=None):
    """refine_mask (ctd_utils/textmask.py:158-174) with the page (u8 [H,W,3]) and the network's mask at page size (u8 [H,W]) on
    the device; returns the refined mask as a device tensor u8 [H,W].  ``refine_mode`` must be None (the ctd detector's call,
    ctd.py:177): the REFINEMASK_INPAINT dilation variant is only in the host routine."""
    import ctypes as C

    import torch

    from . import ops

    if refine_mode is not None:
        raise NotImplementedError("refine_mask_gpu: only refine_mode=None (the detector's call) runs on the GPU")
    if page_dev.dtype != torch.uint8 or pred_dev.dtype != torch.uint8 or not page_dev.is_cuda or not pred_dev.is_cuda:
        raise ValueError("refine_mask_gpu expects uint8 device tensors")
    H, W = int(page_dev.shape[0]), int(page_dev.shape[1])
    if tuple(page_dev.shape) != (H, W, 3) or tuple(pred_dev.shape) != (H, W):
        raise ValueError(f"refine_mask_gpu: page {tuple(page_dev.shape)} / mask {tuple(pred_dev.shape)}")
    page_dev, pred_dev = page_dev.contiguous(), pred_dev.contiguous()
    out = torch.zeros(H, W, dtype=torch.uint8, device=page_dev.device)
    wins = []
    for q in quads:
        bx1, by1, bx2, by2 = enlarge_window(q.xyxy, W, H)
        bx1, by1, bx2, by2 = max(bx1, 0), max(by1, 0), min(bx2, W), min(by2, H)   # what the numpy slices of refine_mask keep
        if bx2 > bx1 and by2 > by1:
            wins.append((bx1, by1, bx2, by2))
    n = len(wins)
    if n == 0:
        return out
    L = _lib.load()
    warr = (_lib.MitRefineWindow * n)()
    for i, (a, b, c, d) in enumerate(wins):
        warr[i].x1, warr[i].y1, warr[i].x2, warr[i].y2 = a, b, c, d
    need = L.mit_ctd_refine_workspace_bytes(C.byref(warr), n)
    if need < 0:
        raise RuntimeError("refine_mask_gpu: bad windows or too many window pixels")
    ws = _REFINE_WS.get(need, page_dev.device)
    st = C.c_void_p(ops.current_stream())
    args = (page_dev.data_ptr(), pred_dev.data_ptr(), H, W, C.byref(warr), n)
    hist = np.zeros((n, 4, 256), dtype=np.int32)
    _lib.check(L.mit_ctd_refine_hist(*args, hist.ctypes.data, ws.data_ptr(), ws.numel(), st), "mit_ctd_refine_hist")
    raw = refine_candidates(hist)
    cands = (_lib.MitRefineCand * (6 * n))()
    for i in range(n):
        for k in range(6):
            e = cands[i * 6 + k]
            e.kind, e.lo, e.hi, e.invert = raw[i][k]
    sums = np.zeros((n, 6), dtype=np.uint64)
    _lib.check(L.mit_ctd_refine_scores(*args, C.byref(cands), sums.ctypes.data, ws.data_ptr(), ws.numel(), st), "mit_ctd_refine_scores")
    ordered = refine_merge_order(raw, sums, [(c - a) * (d - b) for a, b, c, d in wins])
    order = (_lib.MitRefineCand * (4 * n))()
    for i in range(n):
        for slot, cand in enumerate(ordered[i]):
            dst = order[i * 4 + slot]
            dst.kind, dst.lo, dst.hi, dst.invert = cand
    _lib.check(L.mit_ctd_refine_merge(*args, C.byref(order), out.data_ptr(), ws.data_ptr(), ws.numel(), st), "mit_ctd_refine_merge")
    return out
