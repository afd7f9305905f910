"""Mask refinement against the reference's own code.

tests/golden/mask_refinement.npz = mask_refinement.dispatch / complete_mask of the reference executed by oracle/make_golden.py
with stand-ins for cv2 / shapely; pydensecrf exists nowhere this can run, so the DenseCRF call and cv2.bilateralFilter are
replaced by the same deterministic stubs on both sides — the pin covers everything around them."""
import asyncio
import os

import numpy as np
import pytest

from manga_image_translator_amd import mask_refinement as MR
from manga_image_translator_amd.textline import Quadrilateral
from oracle import make_golden as MG

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_refinement.npz"))
REFINE, BILATERAL = MG.mask_refinement_stubs()


@pytest.mark.parametrize("native", [True, False], ids=["native", "numpy"])
@pytest.mark.parametrize("tag,offset,ksize", [("a", 0, 3), ("b", 6, 5)])
def test_complete_mask_matches_reference(tag, offset, ksize, native):
    """Both forms of the component labelling / line assignment (C++ ``mit_mask_assign_lines``, the default; numpy / scipy) against the
    reference's own complete_mask."""
    quads = [Quadrilateral(l.astype(np.float64), "", 0) for l in G["lines"]]
    m = G["mask"].copy()
    got = MR.complete_mask(G["img"].copy(), m, quads, dilation_offset=offset, kernel_size=ksize, refine=REFINE, bilateral=BILATERAL,
                           native=native)
    assert np.array_equal(m, G[f"complete_{tag}_mask_after"])      # the outlined working mask, modified in place like the reference's
    assert np.array_equal(got, G[f"complete_{tag}"])


@pytest.mark.parametrize("tag,offset,ksize", [("a", 0, 3), ("b", 6, 5)])
def test_dispatch_matches_reference(tag, offset, ksize):
    region = type("Region", (), {"lines": G["lines"]})()
    got = asyncio.run(MR.dispatch([region], G["img"].copy(), G["mask"].copy(), "fit_text", offset, 0, False, ksize, refine=REFINE,
                                  bilateral=BILATERAL))
    assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
    assert np.array_equal(got, G[f"dispatch_{tag}"])


@pytest.mark.parametrize("level", [10, 40])
def test_dispatch_with_the_bubble_filter_matches_reference(level):
    """--ignore-bubble (dispatch :34-50 + utils/bubble.py, the reference's own Python over the cv2 stand-ins): a grey page with one
    coloured patch and a bright band on the bottom frame; the two levels keep different components."""
    region = type("Region", (), {"lines": G["lines"]})()
    got = MR.dispatch_sync([region], G["img_bubble"].copy(), G["mask"].copy(), "fit_text", 0, level, False, 3, refine=REFINE, bilateral=BILATERAL)
    assert np.array_equal(got, G[f"dispatch_bubble{level}"])
    assert not np.array_equal(G["dispatch_bubble10"], G["dispatch_bubble40"])
    plain = MR.dispatch_sync([region], G["img_bubble"].copy(), G["mask"].copy(), "fit_text", 0, 0, False, 3, refine=REFINE, bilateral=BILATERAL)
    k = int(max(plain.shape) * 0.025)
    assert (got > 0).sum() < (MR.dilate(plain, np.ones((k, k), np.uint8)) > 0).sum()          # something was erased


def test_bubble_filter_equals_is_ignore_on_blocked_pages():
    """bubble_filter evaluates is_ignore() of "the page, black outside the contour's rectangle" without building that page; the literal
    form (textline.is_ignore on the blocked page) must agree, also for even dilation sizes and rectangles on the page frame."""
    from scipy import ndimage as nd

    from manga_image_translator_amd import textline as TL

    rng = np.random.default_rng(3)
    for trial in range(24):
        H, W = (120, 160) if trial % 2 else (90, 200)
        raw = np.repeat(rng.integers(0, 256, (H, W, 1)).astype(np.uint8), 3, 2)
        if trial % 3 == 0:
            raw[40:60, 50:90] = rng.integers(0, 256, (20, 40, 3))
        if trial % 4 == 0:
            raw[:] = 250
        mask = np.zeros((H, W), np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            y, x = int(rng.integers(0, H - 10)), int(rng.integers(0, W - 10))
            mask[y:y + int(rng.integers(2, 12)), x:x + int(rng.integers(2, 14))] = 255
        level = int(rng.integers(1, 51))
        got = MR.bubble_filter(mask, raw, level)
        k = int(max(H, W) * 0.025)
        want = MR.dilate(mask, np.ones((k, k), np.uint8))
        labels, n = nd.label(nd.binary_fill_holes(want > 0), structure=np.ones((3, 3)))
        for lab, sl in enumerate(nd.find_objects(labels), start=1):
            y0, y1, x0, x1 = sl[0].start, min(sl[0].stop + 1, H), sl[1].start, min(sl[1].stop + 1, W)
            block = np.zeros_like(raw)
            block[y0:y1, x0:x1] = raw[y0:y1, x0:x1]
            if TL.is_ignore(block, level):
                want[labels == lab] = 0
        assert np.array_equal(got, want), trial


def test_dilate_uses_opencv_anchor_for_even_kernels():
    """cv2.dilate: dst(x) = max over taps x' of src(x + x' - k // 2) — for even k the window reaches one pixel further up / left."""
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    out = MR.dilate(img, np.ones((4, 4), np.uint8))
    ys, xs = np.nonzero(out)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (3, 6, 3, 6)       # x + x' - 2 == 4 for x' in 0..3  ->  x in 3..6
    out = MR.dilate(img, np.ones((3, 3), np.uint8))
    assert np.nonzero(out)[0].min() == 3 and np.nonzero(out)[0].max() == 5


def test_nothing_to_keep_gives_an_empty_mask():
    region = type("Region", (), {"lines": np.zeros((0, 4, 2), np.int32)})()
    got = MR.dispatch_sync([region], G["img"], np.zeros_like(G["mask"]), refine=REFINE, bilateral=BILATERAL)
    assert np.array_equal(got, G["dispatch_none"]) and got.shape == G["mask"].shape


def test_scene_exercises_the_assignment_rules():
    """The synthetic page reaches the overlap rule, the distance rule (a stray 4.5 px off line 0 is adopted, one 9 px off is not),
    the speck rule and the component-larger-than-its-line rule."""
    quads = [Quadrilateral(l.astype(np.float64), "", 0) for l in G["lines"]]
    calls = []
    MR.complete_mask(G["img"].copy(), G["mask"].copy(), quads, refine=lambda rgb, m: calls.append(m.shape) or REFINE(rgb, m),
                     bilateral=BILATERAL)
    assert 1 <= len(calls) <= len(quads)
    out = G["complete_a"]
    assert out[60:68, 232:237].any()            # adopted through the distance rule
    assert not out[44:53, 262:271].any()        # too far from every line
    assert not out[250:257, 5:12].any()


def test_no_silent_substitute_for_the_crf():
    quads = [Quadrilateral(l.astype(np.float64), "", 0) for l in G["lines"]]
    with pytest.raises(RuntimeError, match="DenseCRF|bilateralFilter"):
        MR.complete_mask(G["img"].copy(), G["mask"].copy(), quads)


def test_ellipse_kernels():
    assert MR.ellipse_kernel(3).astype(int).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert MR.ellipse_kernel(5).astype(int).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    assert MR.ellipse_kernel(1).astype(int).tolist() == [[1]]


def _random_scene(rng, H, W, n_lines):
    """Blobs of every kind the assignment distinguishes: glyph-sized ones inside rotated line quads, a few far away (nearest-line branch
    and its distance cut-off), tiny ones (<= 9 pixels), blobs larger than their line, blobs touching diagonally, blobs on the frame."""
    mask = np.zeros((H, W), np.uint8)
    quads = []
    for _ in range(n_lines):
        cx, cy = rng.uniform(30, W - 30), rng.uniform(30, H - 30)
        hw, hh = rng.uniform(20, 70), rng.uniform(6, 14)
        if rng.random() < 0.4:
            hw, hh = hh, hw
        a = rng.uniform(-0.5, 0.5) if rng.random() < 0.5 else 0.0
        c, s_ = np.cos(a), np.sin(a)
        pts = np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]]) @ np.array([[c, s_], [-s_, c]]) + [cx, cy]
        quads.append(Quadrilateral(pts.astype(np.float64), "", 0))
        for _ in range(int(rng.integers(2, 9))):   # glyphs along the line
            t = rng.uniform(-0.9, 0.9)
            gx, gy = (np.array([t * hw, rng.uniform(-0.5, 0.5) * hh]) @ np.array([[c, s_], [-s_, c]]) + [cx, cy]).astype(int)
            g = int(rng.integers(2, 7))
            mask[max(gy - g, 0):gy + g, max(gx - g, 0):gx + g] = 255
    for _ in range(12):   # strays
        x, y, w, h = int(rng.integers(0, W - 4)), int(rng.integers(0, H - 4)), int(rng.integers(1, 30)), int(rng.integers(1, 30))
        mask[y:y + h, x:x + w] = 255
    for _ in range(40):   # specks and diagonal chains
        x, y = int(rng.integers(1, W - 6)), int(rng.integers(1, H - 6))
        for d in range(int(rng.integers(1, 6))):
            mask[y + d, x + d] = 255
    mask[rng.random((H, W)) < 0.002] = 255
    return mask, quads


@pytest.mark.parametrize("seed", range(12))
def test_native_component_assignment_equals_the_numpy_form(seed):
    """``mit_mask_assign_lines`` + ``mit_mask_line_crops`` against the numpy / scipy restatement (itself pinned to the reference above) on
    random scenes: the outlined mask, the final refined mask and — through a recording refine stub — every line's crop must be identical."""
    rng = np.random.default_rng(100 + seed)
    H, W = int(rng.integers(120, 260)), int(rng.integers(150, 330))
    mask, quads = _random_scene(rng, H, W, int(rng.integers(1, 9)))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    res = {}
    for native in (True, False):
        seen = []
        m = mask.copy()
        out = MR.complete_mask(img.copy(), m, quads, dilation_offset=int(seed % 3) * 4, kernel_size=3, bilateral=BILATERAL,
                               refine=lambda rgb, cm: seen.append(cm.copy()) or REFINE(rgb, cm), native=native)
        res[native] = (m, out, seen)
    assert np.array_equal(res[True][0], res[False][0])
    assert (res[True][1] is None) == (res[False][1] is None)
    if res[True][1] is not None:
        assert np.array_equal(res[True][1], res[False][1])
    assert len(res[True][2]) == len(res[False][2]) and all(np.array_equal(a, b) for a, b in zip(res[True][2], res[False][2]))


def test_native_component_assignment_edge_cases():
    """An empty mask, a full mask, a one-pixel-wide page: the native form agrees with the numpy form (None or the same mask); no lines: None."""
    q = [Quadrilateral(np.array([[10, 10], [60, 10], [60, 30], [10, 30]], np.float64), "", 0)]
    for mask, quads in [(np.zeros((40, 80), np.uint8), q), (np.full((40, 80), 255, np.uint8), q), (np.full((40, 80), 255, np.uint8), []),
                        (np.full((64, 1), 255, np.uint8), q), (np.full((1, 64), 255, np.uint8), q)]:
        img = np.zeros(mask.shape + (3,), np.uint8)
        a_m, b_m = mask.copy(), mask.copy()
        a = MR.complete_mask(img, a_m, quads, refine=REFINE, bilateral=BILATERAL, native=True)
        if not quads:   # the numpy form (like the reference) takes an argmax over zero lines and raises; the native form keeps nothing
            assert a is None
            continue
        b = MR.complete_mask(img, b_m, quads, refine=REFINE, bilateral=BILATERAL, native=False)
        assert np.array_equal(a_m, b_m)
        assert (a is None and b is None) or np.array_equal(a, b)


def test_native_assignment_reports_capacity_and_shape_errors():
    """The C-ABI entry points of the host half refuse bad shapes and too small output buffers with a message (no writes past the caller's
    buffers), and two crop jobs of the same line both get painted."""
    import ctypes as C

    from manga_image_translator_amd import lib as L

    lib = L.load()
    H, W = 8, 16
    mask = np.zeros((H, W), np.uint8)
    mask[1, 1:5] = 255          # 4 pixels: a speck (<= 9 pixels is never assigned)
    mask[3:7, 2:12] = 255       # 40 pixels inside the line below
    polys = np.array([[[0, 2], [15, 2], [15, 7], [0, 7]]], np.float64)
    boxes = np.array([[-2, -2, 0, 0]], np.int32)   # outline off the page: leaves the mask alone
    font = np.array([5.0])
    runs = np.empty((64, 4), np.int32)
    assign = np.empty(16, np.int32)
    rects = np.empty((1, 4), np.int32)
    n_runs, n_comp = C.c_int64(), C.c_int32()
    args = lambda m, rc, ac: (m.ctypes.data, H, W, boxes.ctypes.data, polys.ctypes.data, font.ctypes.data, 1, 4, 1e-2, runs.ctypes.data, rc,
                              assign.ctypes.data, ac, rects.ctypes.data, C.byref(n_runs), C.byref(n_comp))
    assert lib.mit_mask_assign_lines(*args(mask.copy(), 64, 16)) == 0
    assert (n_runs.value, n_comp.value) == (5, 2) and assign[:3].tolist() == [-1, -1, 0] and rects[0].tolist() == [2, 3, 12, 7]
    assert lib.mit_mask_assign_lines(*args(mask.copy(), 3, 16)) != 0 and b"runs" in lib.mit_last_error()
    assert lib.mit_mask_assign_lines(*args(mask.copy(), 64, 2)) != 0 and b"components" in lib.mit_last_error()
    assert lib.mit_mask_assign_lines(mask.ctypes.data, H, W, boxes.ctypes.data, polys.ctypes.data, font.ctypes.data, 1, 2, 1e-2, runs.ctypes.data, 64,
                                     assign.ctypes.data, 16, rects.ctypes.data, C.byref(n_runs), C.byref(n_comp)) != 0
    assert lib.mit_mask_assign_lines(*args(mask.copy(), 64, 16)) == 0
    jobs = np.array([[0, 2, 3, 10, 4], [0, 0, 0, 16, 8], [3, 0, 0, 4, 4]], np.int32)   # the line's rectangle, the whole page, a line without components
    offs = np.array([0, 40, 40 + 128], np.int64)
    out = np.full(40 + 128 + 16, 7, np.uint8)
    assert lib.mit_mask_line_crops(runs.ctypes.data, n_runs.value, assign.ctypes.data, jobs.ctypes.data, 3, out.ctypes.data, offs.ctypes.data) == 0
    assert (out[:40] == 255).all()
    page = out[40:168].reshape(8, 16)
    assert (page[3:7, 2:12] == 255).all() and page.sum() == 40 * 255      # the speck belongs to no line
    assert not out[168:].any()
    bad = np.array([[0, 0, 0, -1, 4]], np.int32)
    assert lib.mit_mask_line_crops(runs.ctypes.data, n_runs.value, assign.ctypes.data, bad.ctypes.data, 1, out.ctypes.data, offs.ctypes.data) != 0
