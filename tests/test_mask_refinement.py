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
