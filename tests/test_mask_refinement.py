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


@pytest.mark.parametrize("tag,offset,ksize", [("a", 0, 3), ("b", 6, 5)])
def test_complete_mask_matches_reference(tag, offset, ksize):
    quads = [Quadrilateral(l.astype(np.float64), "", 0) for l in G["lines"]]
    m = G["mask"].copy()
    got = MR.complete_mask(G["img"].copy(), m, quads, dilation_offset=offset, kernel_size=ksize, refine=REFINE, bilateral=BILATERAL)
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
