"""TEST INFRASTRUCTURE — loads the reference's own nn.Module files from /root/reference.

Only usable where /root/reference exists (the build container).  It is used by
``oracle/make_golden.py`` to generate the committed fixtures under ``tests/golden/`` and by
the ``not gpu`` tests that validate the oracle restatements (``oracle/*.py``) against the
real reference code.  Nothing in the product imports this.

The reference package itself does not import here (colorama, cv2, the Rust wheel … are not
installed), so the dense model files are loaded by path with stub modules for the missing
third-party imports.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from unittest import mock

sys.dont_write_bytecode = True  # /root/reference is read-only for this repo: importing from it must not leave __pycache__ there

REF_ROOT = "/root/reference"
PKG = os.path.join(REF_ROOT, "manga_translator")

_loaded = {}


def available() -> bool:
    return os.path.isdir(PKG)


def _stub_third_party():
    for name in ["cv2", "colorama", "dotenv", "shapely", "shapely.geometry", "shapely.affinity", "pyclipper",
                 "py3langid", "langcodes", "omegaconf", "torchvision", "torchvision.models", "torchvision.ops",
                 "skimage", "kornia", "timm", "freetype", "pydensecrf", "pydensecrf.utils", "pydensecrf.densecrf",
                 "requests"]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = mock.MagicMock()


def _pkg(name: str):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    return sys.modules[name]


def _load(dotted: str, relpath: str):
    if dotted in _loaded:
        return _loaded[dotted]
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(PKG, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    _loaded[dotted] = mod
    return mod


def _prepare():
    if not available():
        raise RuntimeError("/root/reference is not present on this machine")
    _stub_third_party()
    for p in ["manga_translator", "manga_translator.ocr", "manga_translator.inpainting", "manga_translator.detection",
              "manga_translator.detection.ctd_utils", "manga_translator.detection.ctd_utils.utils",
              "manga_translator.detection.ctd_utils.yolov5"]:
        _pkg(p)
    for name in ["manga_translator.config", "manga_translator.utils", "manga_translator.utils.generic",
                 "manga_translator.utils.bubble"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    # tiny stand-ins for the plugin base classes so the plugin classes in the same files define
    for dotted, names in [("manga_translator.ocr.common", ["OfflineOCR", "CommonOCR"]),
                          ("manga_translator.inpainting.common", ["OfflineInpainter", "CommonInpainter"]),
                          ("manga_translator.detection.common", ["OfflineDetector", "CommonDetector"])]:
        if dotted not in sys.modules:
            m = types.ModuleType(dotted)
            for n in names:
                setattr(m, n, type(n, (), {}))
            sys.modules[dotted] = m


def lama():
    """reference module manga_translator/inpainting/inpainting_lama_mpe.py"""
    _prepare()
    return _load("manga_translator.inpainting.inpainting_lama_mpe", "inpainting/inpainting_lama_mpe.py")


def ocr48():
    """reference module manga_translator/ocr/model_48px.py"""
    _prepare()
    _load("manga_translator.ocr.xpos_relative_position", "ocr/xpos_relative_position.py")
    return _load("manga_translator.ocr.model_48px", "ocr/model_48px.py")


def ocr_ctc():
    """reference module manga_translator/ocr/model_48px_ctc.py"""
    _prepare()
    return _load("manga_translator.ocr.model_48px_ctc", "ocr/model_48px_ctc.py")


def ctd():
    """reference modules manga_translator/detection/ctd_utils/{basemodel,yolov5/*}.py -> (basemodel, yolo)"""
    _prepare()
    base = "manga_translator.detection.ctd_utils"
    _load(base + ".utils.yolov5_utils", "detection/ctd_utils/utils/yolov5_utils.py")
    _load(base + ".utils.weight_init", "detection/ctd_utils/utils/weight_init.py")
    _load(base + ".yolov5.common", "detection/ctd_utils/yolov5/common.py")
    yolo = _load(base + ".yolov5.yolo", "detection/ctd_utils/yolov5/yolo.py")
    bm = _load(base + ".basemodel", "detection/ctd_utils/basemodel.py")
    return bm, yolo


def generic():
    """reference module manga_translator/utils/generic.py (sort_pnts, Quadrilateral, ...), loaded under a private
    package name so the MagicMock standing in for ``manga_translator.utils`` elsewhere is untouched."""
    _prepare()
    if "_ref_utils" not in sys.modules:
        pk = types.ModuleType("_ref_utils")
        pk.__path__ = []
        sys.modules["_ref_utils"] = pk
    _load("_ref_utils.generic2", "utils/generic2.py")
    return _load("_ref_utils.generic", "utils/generic.py")


def cv2_shim():
    """A stand-in ``cv2`` namespace built from the oracle's restatements of the few OpenCV calls on the path, so the
    reference's OWN Python around those calls (letterbox, load_masked_position_encoding's ring loop,
    get_transformed_region) can be executed here.  The primitives themselves stay "parity unpinned" (no real cv2)."""
    import numpy as np

    from . import ctd as OC, lama as OL, textline as OT

    ns = types.SimpleNamespace()
    ns.INTER_NEAREST, ns.INTER_LINEAR, ns.INTER_AREA, ns.INTER_LINEAR_EXACT = 0, 1, 3, 5
    ns.BORDER_CONSTANT, ns.RANSAC, ns.ROTATE_90_COUNTERCLOCKWISE = 0, 8, 2
    ns.COLOR_BGR2RGB, ns.COLOR_RGB2BGR = 4, 4

    def resize(src, dsize, interpolation=1, **_):
        if interpolation == ns.INTER_AREA:
            return OL.resize_area_u8(src, dsize)
        if interpolation == ns.INTER_NEAREST:
            return OL.resize_nearest(src, dsize)
        if interpolation == ns.INTER_LINEAR:
            return OC.resize_linear_u8(src[..., None], dsize)[..., 0] if src.ndim == 2 else OC.resize_linear_u8(src, dsize)
        if interpolation == ns.INTER_LINEAR_EXACT:
            from . import imgproc as OI

            return OI.resize_linear_exact_u8(src, dsize)
        raise NotImplementedError(interpolation)

    def filter2D(src, ddepth, kernel):
        p = np.pad(src, 1, mode="reflect")
        h, w = src.shape
        out = np.zeros_like(src)
        for ky in range(3):
            for kx in range(3):
                out = out + kernel[ky, kx] * p[ky:ky + h, kx:kx + w]
        return out

    def copyMakeBorder(src, top, bottom, left, right, borderType, value=(0, 0, 0)):
        out = np.zeros((src.shape[0] + top + bottom, src.shape[1] + left + right) + src.shape[2:], dtype=src.dtype)
        out[...] = np.asarray(value, dtype=src.dtype)[:src.shape[2]] if src.ndim == 3 else value
        out[top:top + src.shape[0], left:left + src.shape[1]] = src
        return out

    # ---- primitives used by ctd_utils/textmask.py (refine_mask) ----
    from scipy import ndimage as _nd

    ns.MORPH_RECT, ns.MORPH_ELLIPSE, ns.THRESH_BINARY, ns.THRESH_OTSU, ns.CV_16U, ns.COLOR_BGR2GRAY = 0, 2, 0, 8, 2, 6

    def getStructuringElement(shape, ksize, anchor=None):
        w, h = int(ksize[0]), int(ksize[1])
        if shape == ns.MORPH_ELLIPSE:  # OpenCV morph.cpp: row i spans c -+ round(c * sqrt((r^2 - dy^2) / r^2)), r = h / 2, c = w / 2
            r, c = h // 2, w // 2
            inv_r2 = 1.0 / (r * r) if r else 0.0
            k = np.zeros((h, w), np.uint8)
            for i in range(h):
                dy = i - r
                if abs(dy) <= r:
                    dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
                    k[i, max(c - dx, 0):min(c + dx + 1, w)] = 1
            return k
        return np.ones((h, w), np.uint8)

    def _morph(src, kernel, iterations, dil):
        out = src
        for _ in range(iterations):
            fn = _nd.maximum_filter if dil else _nd.minimum_filter  # OpenCV: extreme of src(x + x' - anchor), anchor = k // 2, kernel not reflected
            out = fn(out, footprint=np.asarray(kernel, bool), mode="constant", cval=0 if dil else 255)
        return out

    def threshold(src, thresh, maxval, typ):
        if typ & ns.THRESH_OTSU:
            hist = np.bincount(src.reshape(-1), minlength=256).astype(np.float64) / src.size
            omega = np.cumsum(hist)
            mu = np.cumsum(hist * np.arange(256))
            with np.errstate(divide="ignore", invalid="ignore"):
                sigma = (mu[-1] * omega - mu) ** 2 / (omega * (1 - omega))
            sigma[~np.isfinite(sigma)] = 0
            thresh = int(np.argmax(sigma))
        return thresh, np.where(src > thresh, maxval, 0).astype(np.uint8)

    ns.CC_STAT_LEFT, ns.CC_STAT_TOP, ns.CC_STAT_WIDTH, ns.CC_STAT_HEIGHT, ns.CC_STAT_AREA, ns.CV_32S = 0, 1, 2, 3, 4, 4

    def connectedComponentsWithStats(img, connectivity=8, ltype=4):
        lab, n = _nd.label(img > 0, structure=np.ones((3, 3)) if connectivity == 8 else None)
        stats = np.zeros((n + 1, 5), dtype=np.int32)
        for k in range(n + 1):
            ys, xs = np.nonzero(lab == k)
            if len(ys):
                stats[k] = [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys)]
        return n + 1, lab.astype(np.uint16 if ltype == ns.CV_16U else np.int32), stats, np.zeros((n + 1, 2))

    def rectangle(img, pt1, pt2, color, thickness=1):
        """Axis-aligned rectangle with inclusive corners, clipped to the image; thickness 1 = outline, negative = filled."""
        (x1, x2), (y1, y2) = sorted((int(pt1[0]), int(pt2[0]))), sorted((int(pt1[1]), int(pt2[1])))
        c = color[0] if isinstance(color, (tuple, list)) else color
        H, W = img.shape[:2]
        xa, xb, ya, yb = max(x1, 0), min(x2, W - 1), max(y1, 0), min(y2, H - 1)
        if xa > xb or ya > yb:
            return img
        if thickness < 0:
            img[ya:yb + 1, xa:xb + 1] = c
            return img
        if thickness != 1:
            raise NotImplementedError("rectangle: thickness 1 or filled only")
        for y in (y1, y2):
            if 0 <= y < H:
                img[y, xa:xb + 1] = c
        for x in (x1, x2):
            if 0 <= x < W:
                img[ya:yb + 1, x] = c
        return img

    ns.rectangle = rectangle
    ns.bitwise_not = lambda a: np.bitwise_not(a)
    ns.bilateralFilter = lambda img, d, sigma_color, sigma_space: img  # stand-in; replaced by the caller's stub where it matters

    def cvtColor(src, code):
        if code == ns.COLOR_BGR2GRAY:
            t = src.astype(np.int64)
            return ((t[..., 0] * 1868 + t[..., 1] * 9617 + t[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)
        return src[..., ::-1]

    ns.getStructuringElement = getStructuringElement
    ns.dilate = lambda src, kernel, iterations=1: _morph(src, kernel, iterations, True)
    ns.erode = lambda src, kernel, iterations=1: _morph(src, kernel, iterations, False)
    ns.bitwise_xor, ns.bitwise_or = np.bitwise_xor, np.bitwise_or
    ns.threshold, ns.connectedComponentsWithStats = threshold, connectedComponentsWithStats
    def inRange(src, lo, hi):
        """scalar bounds on an 8-bit image: cvRound to int32 (half to even), empty when inverted / out of range, saturate"""
        ilo, ihi = int(np.rint(lo)), int(np.rint(hi))
        if ilo > ihi or ilo > 255 or ihi < 0:
            return np.zeros(src.shape, np.uint8)
        return np.where((src >= max(ilo, 0)) & (src <= min(ihi, 255)), 255, 0).astype(np.uint8)

    ns.inRange = inRange
    ns.resize, ns.filter2D, ns.copyMakeBorder = resize, filter2D, copyMakeBorder
    ns.cvtColor = cvtColor
    ns.findHomography = lambda s, d, *a, **k: (OT.find_homography_4pt(s, d), None)
    ns.warpPerspective = lambda img, M, dsize, **k: OT.warp_perspective_u8(img, M, dsize)
    ns.rotate = lambda img, code: np.ascontiguousarray(np.rot90(img, 1))
    return ns


def dbnet():
    """reference module manga_translator/detection/default_utils/DBNet_resnet34.py (TextDetection) with the oracle's
    restated ResNet-34 standing in for the absent ``torchvision.models.resnet34``."""
    _prepare()
    from . import dbnet as _odb

    tv = types.ModuleType("torchvision.models")
    tv.resnet34 = _odb.resnet34
    sys.modules["torchvision.models"] = tv
    _pkg("manga_translator.detection.default_utils")
    _load("manga_translator.detection.default_utils.DBHead", "detection/default_utils/DBHead.py")
    return _load("manga_translator.detection.default_utils.DBNet_resnet34", "detection/default_utils/DBNet_resnet34.py")


def esrgan():
    """reference module manga_translator/upscaling/esrgan_pytorch.py (RRDBNet)"""
    _prepare()
    _pkg("manga_translator.upscaling")
    if "manga_translator.upscaling.common" not in sys.modules:
        m = types.ModuleType("manga_translator.upscaling.common")
        m.OfflineUpscaler = type("OfflineUpscaler", (), {})
        sys.modules["manga_translator.upscaling.common"] = m
    return _load("manga_translator.upscaling.esrgan_pytorch", "upscaling/esrgan_pytorch.py")


def textmask():
    """reference module manga_translator/detection/ctd_utils/textmask.py (refine_mask) with the cv2 stand-in."""
    _prepare()
    base = "manga_translator.detection.ctd_utils"
    ip = _load(base + ".utils.imgproc_utils", "detection/ctd_utils/utils/imgproc_utils.py")
    tm = _load(base + ".textmask", "detection/ctd_utils/textmask.py")
    ip.cv2 = tm.cv2 = cv2_shim()
    return tm


def bubble():
    """reference module manga_translator/utils/bubble.py (is_ignore, check_color) with the cv2 stand-in (cv2.threshold only)."""
    _prepare()
    m = _load("manga_translator.utils.bubble", "utils/bubble.py")
    m.cv2 = cv2_shim()
    return m


def mask_refinement(refine_stub=None, bilateral_stub=None, with_bubble=False):
    """reference modules manga_translator/mask_refinement/{text_mask_utils,__init__}.py with the cv2 / shapely stand-ins and the
    reference's own Quadrilateral.  ``pydensecrf`` exists nowhere this can run: the DenseCRF call (text_mask_utils.refine_mask) and
    cv2.bilateralFilter are replaced by the caller's deterministic stubs, so what gets pinned is everything AROUND them — component
    -> text-line assignment, crops, dilation sizes, the resizes of dispatch()."""
    _prepare()
    G = generic()
    shp = shapely_shim()
    G.Polygon, G.MultiPoint = shp.Polygon, shp.MultiPoint
    utils = types.ModuleType("manga_translator.utils")
    utils.Quadrilateral, utils.TextBlock = G.Quadrilateral, type("TextBlock", (), {})
    utils.image_resize = lambda *a, **k: None
    bubble = types.ModuleType("manga_translator.utils.bubble")
    bubble.is_ignore = globals()["bubble"]().is_ignore if with_bubble else (lambda *a, **k: False)   # the reference's own function on request
    saved = {k: sys.modules.get(k) for k in ("manga_translator.utils", "manga_translator.utils.bubble")}
    sys.modules["manga_translator.utils"], sys.modules["manga_translator.utils.bubble"] = utils, bubble
    try:
        _pkg("manga_translator.mask_refinement")
        for k in ("manga_translator.mask_refinement.text_mask_utils", "manga_translator.mask_refinement"):
            _loaded.pop(k, None)
        tmu = _load("manga_translator.mask_refinement.text_mask_utils", "mask_refinement/text_mask_utils.py")
        spec = importlib.util.spec_from_file_location("manga_translator.mask_refinement._init", os.path.join(PKG, "mask_refinement", "__init__.py"),
                                                      submodule_search_locations=[])
        mr = importlib.util.module_from_spec(spec)
        mr.__package__ = "manga_translator.mask_refinement"
        spec.loader.exec_module(mr)
    finally:
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    cv = cv2_shim()
    if bilateral_stub is not None:
        cv.bilateralFilter = bilateral_stub
    if with_bubble:  # the calls of the --ignore-bubble stage of dispatch (:34-50), restated on scipy + oracle/contours.py
        import numpy as np
        from scipy import ndimage as _nd

        from . import contours as OCt

        cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE = 0, 2

        def find_external(img, mode, method):
            assert mode == cv.RETR_EXTERNAL
            filled = _nd.binary_fill_holes(np.asarray(img) > 0)          # a component inside a hole has no external contour
            lab, n = _nd.label(filled, structure=np.ones((3, 3)))
            out = []
            for k in range(1, n + 1):
                cs = OCt.find_contours_list((lab == k).astype(np.uint8))  # hole-free component: exactly its outer border
                assert len(cs) == 1
                out.append(cs[0])
            return out[::-1], None

        def bounding_rect(cnt):
            p = np.asarray(cnt).reshape(-1, 2)
            x0, y0 = p.min(axis=0)
            x1, y1 = p.max(axis=0)
            return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)

        def bitwise_and(a, b, mask=None):
            r = np.bitwise_and(a, b)
            if mask is not None:
                r = np.where((np.asarray(mask) != 0)[..., None] if r.ndim == 3 else (np.asarray(mask) != 0), r, 0).astype(a.dtype)
            return r

        def draw_contours(img, cnts, idx, color, thickness):
            assert thickness < 0 and idx == -1
            for c in cnts:
                OCt.fill_poly(img, np.asarray(c).reshape(-1, 2), color)
            return img

        cv.findContours, cv.boundingRect, cv.bitwise_and, cv.drawContours = find_external, bounding_rect, bitwise_and, draw_contours
    tmu.cv2 = mr.cv2 = cv
    tmu.Polygon = shp.Polygon
    tmu.tqdm = lambda it, *a, **k: it
    if refine_stub is not None:
        tmu.refine_mask = refine_stub
    return mr, tmu, G


def box_shims():
    """(cv2, pyclipper, Polygon) stand-ins for the reference's SegDetectorRepresenter, built from the oracle's restatements:
    cv2.findContours = a Suzuki-Abe border follower, minAreaRect / boxPoints, fillPoly, mean (oracle/contours.py,
    oracle/hostglue.py); pyclipper's round offset; shapely's polygon area / length."""
    import numpy as np

    from . import contours as OCt, hostglue as OH

    cv = types.SimpleNamespace(RETR_LIST=1, CHAIN_APPROX_SIMPLE=2)

    def min_area_rect(contour):
        pts = np.asarray(contour, dtype=np.float64).reshape(-1, 2)
        box, _ = OH.min_area_rect(pts)
        b = box.astype(np.float64)
        w, h = float(np.hypot(*(b[1] - b[0]))), float(np.hypot(*(b[2] - b[1])))
        return (tuple(b.mean(0)), (w, h), 0.0, box)

    def mean(arr, mask):
        sel = np.asarray(mask) > 0
        return (float(np.asarray(arr)[sel].astype(np.float64).mean()) if sel.any() else 0.0, 0, 0, 0)

    cv.findContours = lambda img, mode, method: (OCt.find_contours_list(img), None)
    cv.minAreaRect = min_area_rect
    cv.boxPoints = lambda rect: np.asarray(rect[3], dtype=np.float32)
    cv.fillPoly = lambda mask, pts, color: OCt.fill_poly(mask, np.asarray(pts)[0], color)
    cv.mean = mean

    class _Offset:
        def __init__(self):
            self.path = None

        def AddPath(self, path, jt, et):
            self.path = np.asarray(path)

        def Execute(self, delta):
            out = OH.clipper_offset_round(self.path, float(delta))
            return [out.astype(np.int64).tolist()] if len(out) else []

    clip = types.SimpleNamespace(PyclipperOffset=_Offset, JT_ROUND=1, ET_CLOSEDPOLYGON=0)

    class _Poly:
        def __init__(self, pts):
            self.p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)

        @property
        def area(self):
            x, y = self.p[:, 0], self.p[:, 1]
            return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)

        @property
        def length(self):
            return float(np.hypot(*(np.roll(self.p, -1, 0) - self.p).T).sum())

    return cv, clip, _Poly


def segdet(kind: str = "ctd"):
    """The reference's box extraction, ``SegDetectorRepresenter`` of detection/ctd_utils/utils/db_utils.py (kind "ctd") or
    detection/default_utils/dbnet_utils.py ("default"), with the stand-ins of ``box_shims`` for the three libraries it drives.
    What this pins is the reference's own control flow around them: get_mini_boxes' corner order, box_score_fast's window
    arithmetic, the unclip distance, the size filters, rounding, clipping and scaling to the destination size."""
    _prepare()
    if kind == "ctd":
        mod = _load("manga_translator.detection.ctd_utils.utils.db_utils", "detection/ctd_utils/utils/db_utils.py")
    else:
        _pkg("manga_translator.detection.default_utils")
        mod = _load("manga_translator.detection.default_utils.dbnet_utils", "detection/default_utils/dbnet_utils.py")
    mod.cv2, mod.pyclipper, mod.Polygon = box_shims()
    return mod


def shapely_shim():
    """Minimal ``shapely.geometry`` stand-in (Polygon / MultiPoint with area, length, distance, convex_hull) so the reference's
    own ``quadrilateral_can_merge_region`` / ``Quadrilateral.polygon`` (utils/generic.py) can be executed here.  Geometry by
    straightforward formulas (shoelace, pairwise segment distances) — an independent restatement from the product's."""
    import numpy as np

    class Polygon:
        def __init__(self, pts):
            self.pts = np.asarray([tuple(p) for p in pts], dtype=np.float64)

        @property
        def area(self):
            x, y = self.pts[:, 0], self.pts[:, 1]
            return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)

        @property
        def length(self):
            return float(np.linalg.norm(np.roll(self.pts, -1, 0) - self.pts, axis=1).sum())

        def _edges(self):
            return [(self.pts[i], self.pts[(i + 1) % len(self.pts)]) for i in range(len(self.pts))]

        def _contains(self, p):
            inside = False
            for a, b in self._edges():
                if (a[1] > p[1]) != (b[1] > p[1]) and p[0] < (b[0] - a[0]) * (p[1] - a[1]) / (b[1] - a[1]) + a[0]:
                    inside = not inside
            return inside

        def distance(self, other):
            def seg_seg(a, b, c, d):
                def pt_seg(p, s, e):
                    se = e - s
                    den = float(se @ se)
                    t = 0.0 if den == 0 else min(1.0, max(0.0, float((p - s) @ se) / den))
                    return float(np.linalg.norm(p - (s + t * se)))
                def orient(p, q, r):
                    return np.sign((q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0]))
                if orient(a, b, c) != orient(a, b, d) and orient(c, d, a) != orient(c, d, b):
                    return 0.0
                return min(pt_seg(a, c, d), pt_seg(b, c, d), pt_seg(c, a, b), pt_seg(d, a, b))
            if self._contains(other.pts[0]) or other._contains(self.pts[0]):
                return 0.0
            return min(seg_seg(a, b, c, d) for a, b in self._edges() for c, d in other._edges())

    class Point:
        def __init__(self, x, y):
            self.x, self.y = float(x), float(y)

    def _poly_centroid(self):
        x, y = self.pts[:, 0], self.pts[:, 1]
        xn, yn = np.roll(x, -1), np.roll(y, -1)
        cr = x * yn - xn * y
        a = cr.sum() / 2
        return Point(((x + xn) * cr).sum() / (6 * a), ((y + yn) * cr).sum() / (6 * a))

    def _poly_intersection(self, other):
        """Convex-convex: ``other`` clipped by the half-plane of every edge of ``self`` (orientation-independent)."""
        sgn = 1.0 if np.dot(self.pts[:, 0], np.roll(self.pts[:, 1], -1)) - np.dot(self.pts[:, 1], np.roll(self.pts[:, 0], -1)) > 0 else -1.0
        out = [tuple(p) for p in other.pts]
        for a, b in self._edges():
            side = lambda p: sgn * ((b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]))
            src, out = out, []
            for i, p in enumerate(src):
                q = src[(i + 1) % len(src)]
                sp, sq = side(p), side(q)
                if sp >= 0:
                    out.append(p)
                if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                    t = sp / (sp - sq)
                    out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
            if len(out) < 3:
                return Polygon([(0, 0), (0, 0), (0, 0)])
        return Polygon(out)

    _poly_dist = Polygon.distance

    def _poly_distance(self, other):
        if isinstance(other, Point):
            p = np.array([other.x, other.y])
            if self._contains(p):
                return 0.0
            best = np.inf
            for a, b in self._edges():
                ab = b - a
                den = float(ab @ ab)
                t = 0.0 if den == 0 else min(1.0, max(0.0, float((p - a) @ ab) / den))
                best = min(best, float(np.linalg.norm(p - (a + t * ab))))
            return best
        return _poly_dist(self, other)

    Polygon.centroid = property(_poly_centroid)
    Polygon.intersection = _poly_intersection
    Polygon.distance = _poly_distance

    class MultiPoint:
        def __init__(self, pts):
            self.pts = [tuple(map(float, p)) for p in pts]

        @property
        def convex_hull(self):
            p = sorted(set(self.pts))
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
            return Polygon(lo[:-1] + up[:-1])

    return types.SimpleNamespace(Polygon=Polygon, MultiPoint=MultiPoint, Point=Point)


class RefConfig:
    """Stand-in for ``manga_translator.Config`` (a pydantic model the reference's client only pickles): plain attributes, picklable
    under the module name the reference's own class has, so that the worker's restricted unpickler sees what it would see in production."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __eq__(self, other):
        return isinstance(other, RefConfig) and self.__dict__ == other.__dict__


RefConfig.__module__ = "manga_translator"
RefConfig.__qualname__ = RefConfig.__name__ = "Config"


def server_client():
    """reference module server/sent_data_internal.py (fetch_data_stream, process_stream, handle_buffer, extract_header): the client the
    reference's front server drives a shared-mode worker with.  Its only package import is ``from manga_translator import Config``."""
    if not os.path.isdir(os.path.join(REF_ROOT, "server")):
        raise RuntimeError("/root/reference/server is not present on this machine")
    if "_ref_server.sent_data_internal" in _loaded:
        return _loaded["_ref_server.sent_data_internal"]
    mt = _pkg("manga_translator")
    had = getattr(mt, "Config", None)
    mt.Config = RefConfig
    try:
        spec = importlib.util.spec_from_file_location("_ref_server.sent_data_internal", os.path.join(REF_ROOT, "server", "sent_data_internal.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if had is not None and had is not RefConfig:
            mt.Config = had
    _loaded["_ref_server.sent_data_internal"] = mod
    return mod

