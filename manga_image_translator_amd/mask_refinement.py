"""Mask refinement between OCR and inpainting: which connected components of the detector's raw mask belong
to which text line, per-line clean-up, dilation — the reference's ``mask_refinement.dispatch``
(/root/reference/manga_translator/mask_refinement/__init__.py:9-50) and ``complete_mask``
(mask_refinement/text_mask_utils.py:96-195).

Host-side control flow like the reference's, with the two heavy steps on the GPU.  Native on the host: the 8-bit linear resizes,
component labelling and statistics, the component -> text-line assignment (overlap ratio, distance to the line polygon), crop /
dilation-size arithmetic, elliptical dilation, the final merge.  On the GPU (default backend ``GpuMaskBackend``): the
``cv2.bilateralFilter(img, 17, 80, 80)`` of the page (``mit_bilateral_u8c3``) and the per-line DenseCRF
(``text_mask_utils.refine_mask`` -> pydensecrf), all lines of the page in ONE batched call (``mit_densecrf_refine``) on the
filtered page that never leaves the device.  There is no CPU substitute: without the HIP library / a GPU the default backend
raises.  ``refine=`` / ``bilateral=`` still inject per-crop / per-page callables (the reference's own, or the stubs the pin uses).
Pinned against the reference's Python, run with stand-ins for cv2 / shapely and the same two callables stubbed on both sides
(tests/golden/mask_refinement.npz, oracle/make_golden.py); the GPU steps are checked against oracle/densecrf.py and
oracle/imgproc.bilateral_filter_u8 (parity with pydensecrf / OpenCV themselves is unpinned: neither is installed anywhere this runs)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
from scipy import ndimage as _nd

from . import hostglue as HG
from .textline import Quadrilateral

RefineFn = Callable[[np.ndarray, np.ndarray], np.ndarray]      # (rgb crop, mask crop) -> mask crop   (refine_mask :71-94)
BilateralFn = Callable[[np.ndarray], np.ndarray]               # page -> filtered page                (cv2.bilateralFilter(img, 17, 80, 80) :159)


class CallableMaskBackend:
    """The two heavy steps as host callables: ``bilateral(page) -> page`` and ``refine(rgb crop, mask crop) -> mask crop``
    (e.g. the reference's own ``cv2.bilateralFilter`` / ``refine_mask``, or the stubs of the golden pin)."""

    def __init__(self, refine: RefineFn, bilateral: BilateralFn):
        self._refine, self._bilateral = refine, bilateral

    def resize_image(self, img: np.ndarray, size):
        """cv2.resize(img, (w, h), INTER_LINEAR) of the page (mask_refinement/__init__.py:17); may return a backend handle."""
        return HG.resize_linear_u8(img, size)

    def resize_mask(self, mask: np.ndarray, size) -> np.ndarray:
        return HG.resize_linear_u8(mask, size)

    def filter_page(self, img: np.ndarray):
        return self._bilateral(img)

    def refine(self, page, rects, masks):
        return [self._refine(np.ascontiguousarray(page[y:y + h, x:x + w]), m) for (x, y, w, h), m in zip(rects, masks)]


class GpuMaskBackend:
    """Default backend: bilateral filter + batched DenseCRF on the device (imgproc.bilateral_filter_u8, densecrf.DenseCrfRefiner)."""

    def __init__(self, device=None, gpu_tail: bool = True):
        """``gpu_tail``: the per-line dilations, their union and the closing dilation run on the device too (``refine_dilate_union``);
        False keeps them on the host (scipy) — bit-identical, for A/B runs and tests."""
        import torch

        from . import densecrf, lib

        self.gpu_tail = gpu_tail
        lib.load()  # fails loudly without the HIP library
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("mask refinement: the bilateral filter and the DenseCRF run on the GPU and no GPU is available "
                                   "(pass refine= / bilateral= callables to run them elsewhere)")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._crf = densecrf.DenseCrfRefiner(self.device)

    def _dev(self, a):
        import torch

        return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def resize_image(self, img, size):
        """The scaled page stays on the device: it only feeds the bilateral filter and the CRF crops (same tables as the host
        resize, bit-identical bytes)."""
        from . import imgproc

        return imgproc.resize_u8(self._dev(img)[None], size)[0]

    def resize_mask(self, mask: np.ndarray, size) -> np.ndarray:
        from . import imgproc

        return imgproc.resize_u8(self._dev(mask)[None], size)[0].cpu().numpy()

    def resize_binarize(self, mask_dev, size):
        """cv2.resize(mask, (w, h), INTER_LINEAR) then mask[mask > 0] = 255 (mask_refinement/__init__.py:28-29), on the device."""
        import ctypes as C

        from . import imgproc, lib as _lib, ops

        out = imgproc.resize_u8(mask_dev[None].contiguous(), size)[0]
        _lib.check(_lib.load().mit_binarize_u8(out.data_ptr(), out.numel(), C.c_void_p(ops.current_stream())), "mit_binarize_u8")
        return out

    def filter_page(self, img):
        from . import imgproc

        return imgproc.bilateral_filter_u8(self._dev(img), 17, 80.0, 80.0)

    def refine(self, page, rects, masks):
        return self._crf.refine(page, rects, masks)

    def refine_dilate_union(self, page, jobs, masks, H: int, W: int, kernel_size: int):
        """The tail of complete_mask (text_mask_utils.py:172-195) without leaving the device: batched DenseCRF of the lines' crops, each
        refined crop dilated by its own ellipse inside its window and OR-ed into the page mask, then the closing dilation.
        jobs: (crop rectangle (x, y, w, h), window rectangle, dilation size) per line; masks: the crops' component masks (host).
        Returns the u8 [H, W] mask as a device tensor."""
        import ctypes as C

        import torch

        from . import lib as _lib, ops

        L = _lib.load()
        out_dev, offs = self._crf.refine(page, [r for r, _, _ in jobs], masks, packed=True)
        n = len(jobs)
        final = torch.zeros(2, H, W, dtype=torch.uint8, device=self.device)
        scratch = torch.empty((n + 1) * C.sizeof(_lib.MitDilateJob), dtype=torch.uint8, device=self.device)
        stream = C.c_void_p(ops.current_stream())
        arr = (_lib.MitDilateJob * max(n, 1))()
        for j, ((x1, y1, w1, h1), (x2, y2, w2, h2), k) in enumerate(jobs):
            a = arr[j]
            a.sx, a.sy, a.sw, a.sh, a.dx, a.dy, a.dw, a.dh, a.k, a.spitch, a.src_off = x1, y1, w1, h1, x2, y2, w2, h2, k, w1, offs[j]
        if n:
            _lib.check(L.mit_mask_dilate_jobs(out_dev.data_ptr(), C.byref(arr), n, final[0].data_ptr(), H, W, 1, scratch.data_ptr(), stream),
                       "mit_mask_dilate_jobs")
        one = (_lib.MitDilateJob * 1)()
        a = one[0]
        a.sx, a.sy, a.sw, a.sh, a.dx, a.dy, a.dw, a.dh, a.k, a.spitch, a.src_off = 0, 0, W, H, 0, 0, W, H, kernel_size, W, 0
        _lib.check(L.mit_mask_dilate_jobs(final[0].data_ptr(), C.byref(one), 1, final[1].data_ptr(), H, W, 0,
                                          scratch.data_ptr() + n * C.sizeof(_lib.MitDilateJob), stream), "mit_mask_dilate_jobs")
        return final[1]

    def release_workspace(self):
        self._crf.release_workspace()


_DEFAULT_BACKEND = None


def default_backend():
    global _DEFAULT_BACKEND
    if _DEFAULT_BACKEND is None:
        _DEFAULT_BACKEND = GpuMaskBackend()
    return _DEFAULT_BACKEND


def _backend_for(refine: Optional[RefineFn], bilateral: Optional[BilateralFn], backend):
    if backend is not None:
        return backend
    if refine is None and bilateral is None:
        return default_backend()
    if refine is None or bilateral is None:
        raise ValueError("mask refinement: inject both refine= and bilateral= (or neither, for the GPU backend)")
    return CallableMaskBackend(refine, bilateral)


def ellipse_kernel(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)): row i covers c -+ round(c * sqrt(1 - (dy / r)^2)), r = c = k // 2."""
    r = c = k // 2
    out = np.zeros((k, k), dtype=bool)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / (r * r)))) if r else 0
            out[i, max(c - dx, 0):min(c + dx + 1, k)] = True
    return out


def dilate(img: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """cv2.dilate with the default anchor / border: dst(x) = max over kernel taps x' of src(x + x' - k // 2), pixels outside the image
    ignored.  ``maximum_filter`` has exactly these offsets; ``grey_dilation`` reflects the footprint and is one pixel off for even sizes."""
    if img.size == 0:
        return img
    return _nd.maximum_filter(img, footprint=kernel, mode="constant", cval=0)


def _clip_quad_to_rect_area(pts: np.ndarray, x0: float, y0: float, x1: float, y1: float) -> float:
    """Area of (convex quad) ∩ (axis-aligned rectangle): the quad clipped by the four sides in turn (Sutherland-Hodgman)."""
    poly = [tuple(map(float, p)) for p in pts]
    for axis, bound, keep_ge in ((0, x0, True), (0, x1, False), (1, y0, True), (1, y1, False)):
        if not poly:
            return 0.0
        nxt = []
        for i, p in enumerate(poly):
            q = poly[(i + 1) % len(poly)]
            dp, dq = (p[axis] - bound, q[axis] - bound) if keep_ge else (bound - p[axis], bound - q[axis])
            if dp >= 0:
                nxt.append(p)
            if (dp > 0 and dq < 0) or (dp < 0 and dq > 0):
                t = dp / (dp - dq)
                nxt.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        poly = nxt
    if len(poly) < 3:
        return 0.0
    a = np.asarray(poly)
    x, y = a[:, 0], a[:, 1]
    return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)


def _point_in_polygon(pts: np.ndarray, p) -> bool:
    inside = False
    n = len(pts)
    for i in range(n):
        a, b = pts[i], pts[(i + 1) % n]
        if (a[1] > p[1]) != (b[1] > p[1]) and p[0] < (b[0] - a[0]) * (p[1] - a[1]) / (b[1] - a[1]) + a[0]:
            inside = not inside
    return inside


def _polygon_point_distance(pts: np.ndarray, p) -> float:
    pts = np.asarray(pts, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    if _point_in_polygon(pts, p):
        return 0.0
    best = np.inf
    for i in range(len(pts)):
        a, b = pts[i], pts[(i + 1) % len(pts)]
        ab = b - a
        den = float(ab @ ab)
        t = 0.0 if den == 0 else min(1.0, max(0.0, float((p - a) @ ab) / den))
        best = min(best, float(np.linalg.norm(p - (a + t * ab))))
    return best


def _polygon_point_distances(polys: np.ndarray, p) -> np.ndarray:
    """``_polygon_point_distance`` of one point against M polygons of equal vertex count at once (polys float64 [M, V, 2]): the
    distance row of text_mask_utils.py:131 for one component, without M x V Python iterations."""
    px, py = float(p[0]), float(p[1])
    a = polys
    b = np.roll(polys, -1, axis=1)
    ax, ay, bx, by = a[..., 0], a[..., 1], b[..., 0], b[..., 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        cross = ((ay > py) != (by > py)) & (px < (bx - ax) * (py - ay) / (by - ay) + ax)   # the crossing test of _point_in_polygon, edge by edge
    inside = (np.count_nonzero(cross, axis=1) % 2) == 1
    abx, aby = bx - ax, by - ay
    den = abx * abx + aby * aby
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(den == 0, 0.0, np.minimum(1.0, np.maximum(0.0, ((px - ax) * abx + (py - ay) * aby) / den)))
    dx, dy = px - (ax + t * abx), py - (ay + t * aby)
    d = np.sqrt(dx * dx + dy * dy).min(axis=1)
    return np.where(inside, 0.0, d)


def _extend_rect(x, y, w, h, max_x, max_y, extend):  # text_mask_utils.py:56-61
    x1, y1 = max(x - extend, 0), max(y - extend, 0)
    return x1, y1, min(w + extend * 2, max_x - x1 - 1), min(h + extend * 2, max_y - y1 - 1)


def _xywh(q: Quadrilateral):  # BBox.xywh (utils/generic.py:319-321): int32 truncation of the axis-aligned box
    mn, mx = np.min(q.pts, axis=0), np.max(q.pts, axis=0)
    return tuple(int(v) for v in np.array([mn[0], mn[1], mx[0] - mn[0], mx[1] - mn[1]], dtype=np.int32))


def _assign_components_native(mask: np.ndarray, textlines: Sequence[Quadrilateral], keep_threshold: float):
    """The component labelling and line assignment of complete_mask in native host code (``mit_mask_assign_lines``, csrc/hostglue.hip).
    Returns (rects per line or None, crop(i, x, y, w, h) -> the line's component image inside a rectangle, crops(jobs) -> all at once)."""
    import ctypes as C

    from . import lib as _lib

    L = _lib.load()
    H, W = mask.shape
    M = len(textlines)
    boxes = np.ascontiguousarray([_xywh(t) for t in textlines], dtype=np.int32).reshape(M, 4)
    polys = np.ascontiguousarray([np.asarray(t.pts, dtype=np.float64) for t in textlines], dtype=np.float64).reshape(M, 4, 2)
    V = 4
    font = np.ascontiguousarray([float(t.font_size) for t in textlines], dtype=np.float64)
    runs = np.empty((H * ((W + 1) // 2), 4), dtype=np.int32)              # upper bounds; the pages are only touched as far as they are used
    assign = np.empty(((H + 1) // 2) * ((W + 1) // 2) + 1, dtype=np.int32)
    rects = np.empty((max(M, 1), 4), dtype=np.int32)
    n_runs, n_comp = C.c_int64(0), C.c_int32(0)
    if not mask.flags.c_contiguous or mask.dtype != np.uint8:
        raise ValueError("complete_mask: a C-contiguous uint8 mask is required")
    _lib.check(L.mit_mask_assign_lines(mask.ctypes.data, H, W, boxes.ctypes.data, polys.ctypes.data, font.ctypes.data, M, V,
                                       float(keep_threshold), runs.ctypes.data, runs.shape[0], assign.ctypes.data, assign.shape[0],
                                       rects.ctypes.data, C.byref(n_runs), C.byref(n_comp)), "mit_mask_assign_lines")
    nr = int(n_runs.value)

    def crops(jobs):  # jobs: (line, x, y, w, h)
        if not jobs:
            return []
        ja = np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 5)
        offs = np.zeros(len(jobs) + 1, dtype=np.int64)
        np.cumsum(ja[:, 3].astype(np.int64) * ja[:, 4], out=offs[1:])
        out = np.empty(int(offs[-1]), dtype=np.uint8)
        _lib.check(L.mit_mask_line_crops(runs.ctypes.data, nr, assign.ctypes.data, ja.ctypes.data, len(jobs), out.ctypes.data,
                                         offs.ctypes.data), "mit_mask_line_crops")
        return [out[offs[j]:offs[j + 1]].reshape(int(ja[j, 4]), int(ja[j, 3])) for j in range(len(jobs))]

    line_rects = [None if rects[i, 0] < 0 else [int(v) for v in rects[i]] for i in range(M)]
    return line_rects, (lambda i, x, y, w, h: crops([(i, x, y, w, h)])[0]), crops


def complete_mask(img: np.ndarray, mask: np.ndarray, textlines: Sequence[Quadrilateral], keep_threshold: float = 1e-2,
                  dilation_offset: int = 0, kernel_size: int = 3, refine: Optional[RefineFn] = None,
                  bilateral: Optional[BilateralFn] = None, backend=None, device_result: bool = False,
                  native: Optional[bool] = None) -> Optional[np.ndarray]:
    """text_mask_utils.complete_mask (:96-195).  ``mask`` is modified in place exactly like the reference's (line boxes are
    outlined with zeros before labelling).  The per-line DenseCRF calls of :172-176 are independent of each other (each line owns
    its component image, all read the same filtered page), so they are collected and refined as one batch.
    ``native``: component labelling and line assignment in C++ (``mit_mask_assign_lines``; the default whenever every line is a
    quadrilateral) or in numpy / scipy (the restatement the golden cases pin; same assignment, kept for A/B runs and as the statement of
    what the native code computes)."""
    be = _backend_for(refine, bilateral, backend)
    H, W = mask.shape
    if native is None:
        native = all(np.asarray(t.pts).shape == (4, 2) for t in textlines)
    if native:
        # a device backend starts the page's bilateral filter now: it runs while the host labels and assigns the components
        page = be.filter_page(img) if getattr(be, "gpu_tail", None) is not None else None
        rects, _line_crop, _line_crops = _assign_components_native(mask, textlines, keep_threshold)
        if not any(r is not None for r in rects):
            return None
        return _complete_mask_tail(be, img, mask, textlines, rects, _line_crop, _line_crops, dilation_offset, kernel_size, device_result,
                                   page=page)
    boxes = [_xywh(t) for t in textlines]
    polys = [np.asarray(t.pts, dtype=np.float64) for t in textlines]
    areas2 = [HG_area(p) for p in polys]
    poly_arr = np.stack(polys) if polys and all(p.shape == polys[0].shape for p in polys) else None   # [M, V, 2]: vectorised distance rows
    pmin = np.array([p.min(0) for p in polys]).reshape(-1, 2)
    pmax = np.array([p.max(0) for p in polys]).reshape(-1, 2)
    for x, y, w, h in boxes:  # cv2.rectangle(mask, (x, y), (x + w, y + h), 0, 1): one-pixel outline, inclusive corners, clipped
        xa, xb, ya, yb = max(x, 0), min(x + w, W - 1), max(y, 0), min(y + h, H - 1)
        if xa > xb or ya > yb:
            continue
        for yy in (y, y + h):
            if 0 <= yy < H:
                mask[yy, xa:xb + 1] = 0
        for xx in (x, x + w):
            if 0 <= xx < W:
                mask[ya:yb + 1, xx] = 0
    labels, n = _nd.label(mask > 0, structure=np.ones((3, 3)))  # 8-connectivity (cv2.connectedComponentsWithStats default)
    objs = _nd.find_objects(labels)
    counts = np.bincount(labels.reshape(-1), minlength=n + 1)
    M = len(textlines)
    members: List[List[int]] = [[] for _ in range(M)]  # labels of the components given to each line (the reference paints them into one
    rects: List[Optional[List[int]]] = [None] * M      # page-sized image per line; only the crop of that image is ever read: _line_crop)
    valid = False
    for label in range(1, n + 1):
        area1 = int(counts[label])
        if area1 <= 9:
            continue
        sy, sx = objs[label - 1]
        x1, y1, w1, h1 = sx.start, sy.start, sx.stop - sx.start, sy.stop - sy.start
        ratio = np.zeros(M, dtype=np.float32)
        # a line whose bounding box misses the component's has overlap exactly 0: only the others are clipped (:129-130)
        near = np.nonzero((pmin[:, 0] <= x1 + w1) & (pmax[:, 0] >= x1) & (pmin[:, 1] <= y1 + h1) & (pmax[:, 1] >= y1))[0]
        for i in near:
            ratio[i] = _clip_quad_to_rect_area(polys[i], x1, y1, x1 + w1, y1 + h1) / min(area1, areas2[i])
        avg = int(np.argmax(ratio))
        if area1 >= areas2[avg]:
            continue
        if ratio[avg] <= keep_threshold:
            # the distance matrix of :131 is only read on this branch, so it is only evaluated here
            centre = (x1 + w1 / 2.0, y1 + h1 / 2.0)
            dist = (_polygon_point_distances(poly_arr, centre).astype(np.float32) if poly_arr is not None
                    else np.array([_polygon_point_distance(polys[i], centre) for i in range(M)], dtype=np.float32))
            avg = int(np.argmin(dist))
            unit = max(min([textlines[avg].font_size, w1, h1]), 10)
            if dist[avg] >= 0.5 * unit:
                continue
        members[avg].append(label)
        r = rects[avg]
        rects[avg] = [x1, y1, x1 + w1, y1 + h1] if r is None else [min(r[0], x1), min(r[1], y1), max(r[2], x1 + w1), max(r[3], y1 + h1)]
        valid = True
    if not valid:
        return None
    lut = np.zeros(n + 1, dtype=np.uint8)

    def _line_crop(i, x, y, w, h):
        """The crop [y:y+h, x:x+w] of line i's component image (255 on the pixels of the components assigned to it)."""
        lut[members[i]] = 255
        out = lut[labels[y:y + h, x:x + w]]
        lut[members[i]] = 0
        return np.ascontiguousarray(out)

    return _complete_mask_tail(be, img, mask, textlines, rects, _line_crop, lambda jobs: [_line_crop(*j) for j in jobs], dilation_offset,
                               kernel_size, device_result)


def _complete_mask_tail(be, img, mask, textlines, rects, _line_crop, _line_crops, dilation_offset, kernel_size, device_result, page=None):
    """complete_mask from the per-line rectangles on (:172-195): crop, DenseCRF, per-line dilation, union, closing dilation."""
    H, W = mask.shape
    M = len(textlines)
    final = np.zeros_like(mask)
    if page is None:
        page = be.filter_page(img)
    jobs = []  # (line index, crop rectangle, dilation size)
    for i in range(M):
        if rects[i] is None:  # the reference's sentinel rectangle slices to an empty crop and is skipped (:172-173)
            continue
        x1, y1, w1, h1 = rects[i][0], rects[i][1], rects[i][2] - rects[i][0], rects[i][3] - rects[i][1]
        text_size = min(w1, h1, textlines[i].font_size)
        x1, y1, w1, h1 = _extend_rect(x1, y1, w1, h1, W, H, int(text_size * 0.1))
        dilate_size = max((int((text_size + dilation_offset) * 0.3) // 2) * 2 + 1, 3)
        if w1 <= 0 or h1 <= 0 or x1 >= W or y1 >= H:   # an empty slice
            continue
        jobs.append((i, (x1, y1, w1, h1), dilate_size))
    crops = _line_crops([(i, x, y, w, h) for i, (x, y, w, h), _ in jobs])
    # (the device dilation takes ellipse sizes up to 255 — a text size of ~830 px at the scaled page; anything larger stays on the host form)
    if getattr(be, "gpu_tail", False) and kernel_size % 2 == 1 and kernel_size <= 255 and all(k <= 255 for _, _, k in jobs):
        # device tail: the windows are the host form's own rectangles; inside a window every non-zero pixel of the line's component
        # image lies in its crop rectangle (the crop is the components' bounding box, extended), so the refined crop is all the
        # dilation has to read
        dev_jobs = [(r, _extend_rect(*r, W, H, -(-k // 2)), k) for _, r, k in jobs]
        out = be.refine_dilate_union(page, dev_jobs, crops, H, W, kernel_size)
        return out if device_result else out.cpu().numpy()   # device_result: the caller (dispatch) resizes it on the device
    refined = be.refine(page, [r for _, r, _ in jobs], crops)
    for (i, (x1, y1, w1, h1), dilate_size), region in zip(jobs, refined):
        x2, y2, w2, h2 = _extend_rect(x1, y1, w1, h1, W, H, -(-dilate_size // 2))
        # the line's component image inside the dilation window: its components (all inside the window or outside it: see the note on
        # the device tail), with the crop replaced by its refined version
        cc = _line_crop(i, x2, y2, w2, h2)
        ya, yb, xa, xb = max(y1, y2), min(y1 + h1, y2 + h2), max(x1, x2), min(x1 + w1, x2 + w2)
        cc[ya - y2:yb - y2, xa - x2:xb - x2] = region[ya - y1:yb - y1, xa - x1:xb - x1]
        final[y2:y2 + h2, x2:x2 + w2] |= dilate(cc, ellipse_kernel(dilate_size))
    return dilate(final, ellipse_kernel(kernel_size))


def HG_area(pts: np.ndarray) -> float:
    x, y = pts[:, 0], pts[:, 1]
    return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)


def bubble_filter(final_mask: np.ndarray, raw_image: np.ndarray, ignore_bubble: int) -> np.ndarray:
    """The ``--ignore-bubble`` stage of mask_refinement.dispatch (:34-50): dilate the mask by a square of 2.5 % of the longer page side,
    and for every outer contour of the result look at the page inside the contour's bounding rectangle (one pixel larger to the right /
    bottom: ``cv2.rectangle`` corners are inclusive) through ``is_ignore`` — the 2-pixel frame of the WHOLE page with everything outside
    the rectangle black, then the colour test — and erase the contour's filled polygon (component + enclosed holes) when it says so."""
    H, W = final_mask.shape
    k = int(max(H, W) * 0.025)
    out = dilate(final_mask, np.ones((k, k), np.uint8)) if k > 0 else final_mask.copy()
    filled = _nd.binary_fill_holes(out > 0)                                    # RETR_EXTERNAL: components nested in holes do not get their own contour
    labels, n = _nd.label(filled, structure=np.ones((3, 3)))
    if n == 0:
        return out
    # is_ignore() of a page that is black outside a rectangle, without building that page per contour: the frame statistics only need
    # the bright frame values inside the rectangle, the colour test only the rectangle's pixels (black pixels have no colour distance)
    img = np.asarray(raw_image)
    bright = img > 127
    frame = np.zeros((H, W), bool)
    frame[:2] = frame[H - 2:] = True
    frame[:, :2] = frame[:, W - 2:] = True
    total = int(frame.sum()) * (img.shape[2] if img.ndim == 3 else 1)
    gray = np.dot(img[..., :3], [0.299, 0.587, 0.114])[..., np.newaxis]
    coloured = np.sum((img - gray) ** 2, axis=-1) > 100
    for lab, sl in enumerate(_nd.find_objects(labels), start=1):
        y0, y1, x0, x1 = sl[0].start, min(sl[0].stop + 1, H), sl[1].start, min(sl[1].stop + 1, W)
        fr = frame[y0:y1, x0:x1]
        val0 = total - int((bright[y0:y1, x0:x1] & (fr[..., None] if img.ndim == 3 else fr)).sum())
        ratio = round(val0 / total, 6) * 100
        if ignore_bubble <= ratio <= 100 - ignore_bubble or int(coloured[y0:y1, x0:x1].sum()) > 10:
            out[labels == lab] = 0
    return out


def dispatch_sync(text_regions, raw_image: np.ndarray, raw_mask: np.ndarray, method: str = "fit_text", dilation_offset: int = 0,
                  ignore_bubble: int = 0, verbose: bool = False, kernel_size: int = 3, refine: Optional[RefineFn] = None,
                  bilateral: Optional[BilateralFn] = None, backend=None, device_result: bool = False) -> np.ndarray:
    """mask_refinement.dispatch (:9-50) for ``method='fit_text'``; ``ignore_bubble`` in 1..50 adds the bubble stage (``bubble_filter``).
    With the GPU backend ``raw_image`` / ``raw_mask`` may be device tensors (a batch engine keeps pages and masks resident), and
    ``device_result`` returns the final mask as a device tensor (no bubble stage then: it needs the page on the host)."""
    if device_result and 1 <= ignore_bubble <= 50:
        raise NotImplementedError("mask refinement: the --ignore-bubble stage runs on the host (device_result=False)")
    if method != "fit_text":
        raise NotImplementedError("mask refinement: only method='fit_text' is native (the reference's 'fill' path references an unset variable)")
    h, w = raw_image.shape[:2]
    scale = max(min((raw_mask.shape[0] - h / 3) / raw_mask.shape[0], 1), 0.5)
    size = (int(w * scale), int(h * scale))
    be = _backend_for(refine, bilateral, backend)
    img_small = be.resize_image(raw_image, size)  # a device tensor with the GPU backend: it never comes back to the host
    if hasattr(be, "resize_binarize"):  # device backend: resize and "> 0 -> 255" there, one download
        mask_small = be.resize_binarize(be._dev(raw_mask), size).cpu().numpy()
    else:
        mask_small = be.resize_mask(raw_mask, size).copy()
        mask_small[mask_small > 0] = 255
    lines = [Quadrilateral(np.asarray(l) * scale, "", 0) for region in text_regions for l in region.lines]
    final = complete_mask(img_small, mask_small, lines, dilation_offset=dilation_offset, kernel_size=kernel_size, backend=be,
                          device_result=True)
    if final is None:
        final = np.zeros((h, w), dtype=np.uint8)
    elif isinstance(final, np.ndarray):
        final = be.resize_mask(final, (w, h)).copy()
        final[final > 0] = 255
    else:  # device tensor from the GPU tail: resize and binarise there, one download
        final = be.resize_binarize(final, (w, h))
        if device_result:
            return final
        final = final.cpu().numpy()
    if device_result:
        import torch

        return torch.from_numpy(final).to(be.device)
    if 1 <= ignore_bubble <= 50:
        final = bubble_filter(final, np.asarray(raw_image), ignore_bubble)
    return final


async def dispatch(text_regions, raw_image: np.ndarray, raw_mask: np.ndarray, method: str = "fit_text", dilation_offset: int = 0,
                   ignore_bubble: int = 0, verbose: bool = False, kernel_size: int = 3, **kw) -> np.ndarray:
    return dispatch_sync(text_regions, raw_image, raw_mask, method, dilation_offset, ignore_bubble, verbose, kernel_size, **kw)


def dispatch_device(text_regions, page_dev, mask_dev, dilation_offset: int = 0, kernel_size: int = 3, backend=None):
    """``dispatch`` for a device-resident page (u8 [H,W,3]) and raw mask (u8 [H,W]); returns the refined mask on the device.  Only the
    down-scaled raw mask (for the component labelling) and the per-line component crops cross PCIe."""
    return dispatch_sync(text_regions, page_dev, mask_dev, "fit_text", dilation_offset, 0, False, kernel_size, backend=backend or default_backend(),
                         device_result=True)
