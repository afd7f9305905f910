"""GPU halves of the mask refinement: cv2.bilateralFilter and the batched per-line DenseCRF, through the C-ABI,
against the CPU oracle (oracle/imgproc.bilateral_filter_u8, oracle/densecrf.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(seed, H, W, n_glyphs=6):
    rng = np.random.default_rng(seed)
    img = np.full((H, W, 3), 225, np.uint8) + rng.integers(0, 16, (H, W, 3)).astype(np.uint8)
    img[:, W // 2:] = (img[:, W // 2:].astype(np.int32) * 3 // 4).astype(np.uint8)  # a tone step under the text
    mask = np.zeros((H, W), np.uint8)
    gw = max(W // (2 * n_glyphs), 3)
    for k in range(n_glyphs):
        x0 = 4 + k * 2 * gw
        img[H // 4:3 * H // 4, x0:x0 + gw] = rng.integers(10, 60)
        mask[max(H // 4 - 2, 0):3 * H // 4 + 2, max(x0 - 2, 0):x0 + gw + 2] = 255
    mask[rng.random((H, W)) < 0.02] ^= 255
    return img, mask


@pytest.mark.parametrize("H,W", [(67, 93), (32, 32), (5, 70), (130, 9), (256, 300)])
def test_bilateral_filter_bit_exact(cuda, H, W):
    """mit_bilateral_u8c3 == the restated cv2.bilateralFilter(img, 17, 80, 80): same tables, same fp32 sums in tap order, same
    rounding -> identical bytes, including the reflect-101 border on images narrower than the radius."""
    from manga_image_translator_amd import imgproc
    from oracle import imgproc as OI

    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    img[H // 3:2 * H // 3, W // 4:3 * W // 4] //= 6  # an edge the range kernel has to preserve
    got = imgproc.bilateral_filter_u8(torch.from_numpy(img).to(cuda)).cpu().numpy()
    assert np.array_equal(got, OI.bilateral_filter_u8(img, 17, 80.0, 80.0))
    # other parameters (d = 9, different sigmas) and the batched form
    got2 = imgproc.bilateral_filter_u8(torch.from_numpy(np.stack([img, img[::-1].copy()])).to(cuda), 9, 25.0, 3.0).cpu().numpy()
    assert np.array_equal(got2[0], OI.bilateral_filter_u8(img, 9, 25.0, 3.0))
    assert np.array_equal(got2[1], OI.bilateral_filter_u8(img[::-1].copy(), 9, 25.0, 3.0))


def test_bilateral_rejects_bad_arguments(cuda):
    from manga_image_translator_amd import imgproc, lib as L

    with pytest.raises(ValueError):
        imgproc.bilateral_filter_u8(torch.zeros(4, 4, 3, device=cuda))
    with pytest.raises(RuntimeError, match="radius"):
        imgproc.bilateral_filter_u8(torch.zeros(4, 4, 3, dtype=torch.uint8, device=cuda), 41, 10.0, 10.0)
    assert L.load().mit_bilateral_u8c3(None, None, 1, 4, 4, 8, 197, None, None, None, None) != 0


def _check_against_oracle(rgb, mask, got_mask, got_q, iterations=5):
    from oracle import densecrf as OD

    ref_mask, q = OD.refine_mask(rgb, mask, n_iterations=iterations, return_q=True)
    h, w = mask.shape
    ref_q = q.T.reshape(h, w, 2)
    err = np.abs(got_q - ref_q).max()
    assert err < 2e-4, err
    margin = np.abs(ref_q[..., 1] - ref_q[..., 0]) < 1e-3
    assert np.array_equal(got_mask[~margin], ref_mask[~margin])
    return err, int(margin.sum())


@pytest.mark.parametrize("H,W", [(40, 120), (33, 57), (1, 64), (90, 14), (1500, 24)])   # the last: a page-tall crop (a line that was given a panel border)
def test_densecrf_matches_oracle_single_crop(cuda, H, W):
    """mit_densecrf_refine on one crop == the restated DenseCRF2D: final marginals within 2e-4 (the device splat sums exactly in
    fixed point, the library sequentially in fp32), argmax identical outside a 1e-3 margin."""
    from manga_image_translator_amd import densecrf

    rgb, mask = _scene(H + W, H, W)
    ref = densecrf.DenseCrfRefiner(cuda)
    masks, qs = ref.refine(torch.from_numpy(rgb).to(cuda), [(0, 0, W, H)], [mask], return_q=True)
    err, near = _check_against_oracle(rgb, mask, masks[0], qs[0])
    print(f"densecrf {H}x{W}: max |dQ| {err:.2e}, {near} px inside the argmax margin")
    assert set(np.unique(masks[0])) <= {0, 255}


@pytest.mark.parametrize("iterations", [1, 2, 5])
def test_densecrf_unsaturated_marginals_match_oracle(cuda, iterations):
    """A low-contrast noisy crop with a coin-flip mask keeps a good share of the marginals away from 0 / 1 (the glyph scenes
    saturate them), so the comparison exercises the lattice arithmetic itself: after 1, 2 and 5 mean-field steps."""
    from manga_image_translator_amd import densecrf

    rng = np.random.default_rng(3)
    H, W = 40, 120
    rgb = rng.integers(90, 170, (H, W, 3)).astype(np.uint8)
    mask = ((rng.random((H, W)) < 0.5) * 255).astype(np.uint8)
    ref = densecrf.DenseCrfRefiner(cuda)
    masks, qs = ref.refine(torch.from_numpy(rgb).to(cuda), [(0, 0, W, H)], [mask], return_q=True, iterations=iterations)
    mid = ((qs[0][..., 1] > 1e-3) & (qs[0][..., 1] < 1 - 1e-3)).mean()
    assert mid > 0.02, mid
    err, near = _check_against_oracle(rgb, mask, masks[0], qs[0], iterations)
    print(f"densecrf noisy crop, {iterations} iterations: {mid:.1%} of marginals unsaturated, max |dQ| {err:.2e}, {near} px inside the margin")


def test_densecrf_batch_of_crops_inside_a_page(cuda):
    """Several crops of one page in one call: each equals the oracle on its own crop (crops cannot alias in the shared hash table),
    the batch is bit-identical to one-crop-at-a-time calls, and a second run reproduces it exactly (integer atomics)."""
    from manga_image_translator_amd import densecrf

    page, pmask = _scene(7, 200, 320, n_glyphs=10)
    rects = [(10, 20, 150, 60), (100, 50, 200, 120), (0, 0, 40, 200), (300, 180, 20, 20), (10, 20, 150, 60)]
    masks = [np.ascontiguousarray(pmask[y:y + h, x:x + w]) for x, y, w, h in rects]
    ref = densecrf.DenseCrfRefiner(cuda)
    pd = torch.from_numpy(page).to(cuda)
    got, qs = ref.refine(pd, rects, masks, return_q=True)
    for (x, y, w, h), m, gm, gq in zip(rects, masks, got, qs):
        _check_against_oracle(np.ascontiguousarray(page[y:y + h, x:x + w]), m, gm, gq)
    assert np.array_equal(got[0], got[4]) and np.array_equal(qs[0], qs[4])  # same crop twice in the batch
    again, qs2 = ref.refine(pd, rects, masks, return_q=True)
    for a, b, qa, qb in zip(got, again, qs, qs2):
        assert np.array_equal(a, b) and np.array_equal(qa, qb)
    for r, m, g, q in zip(rects, masks, got, qs):
        one, q1 = ref.refine(pd, [r], [m], return_q=True)
        assert np.array_equal(one[0], g) and np.array_equal(q1[0], q)


def test_densecrf_rejects_bad_crops(cuda):
    from manga_image_translator_amd import densecrf

    ref = densecrf.DenseCrfRefiner(cuda)
    page = torch.zeros(50, 60, 3, dtype=torch.uint8, device=cuda)
    with pytest.raises(RuntimeError, match="outside|bad crop"):
        ref.refine(page, [(40, 40, 30, 30)], [np.zeros((30, 30), np.uint8)])
    with pytest.raises(ValueError):
        ref.refine(page, [(0, 0, 10, 10)], [np.zeros((5, 5), np.uint8)])
    assert ref.refine(page, [], []) == []


def test_complete_mask_default_backend_matches_oracle_backend(cuda):
    """mask_refinement.complete_mask / dispatch with the default (GPU) backend on the golden scene == the same host code driven by
    the CPU oracle's bilateral filter + DenseCRF: the refined page mask is identical except where a marginal sits on the argmax margin."""
    from manga_image_translator_amd import mask_refinement as MR
    from manga_image_translator_amd.textline import Quadrilateral
    from oracle import densecrf as OD, imgproc as OI

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_refinement.npz"))
    quads = [Quadrilateral(l.astype(np.float64), "", 0) for l in G["lines"]]
    got = MR.complete_mask(G["img"].copy(), G["mask"].copy(), quads, backend=MR.GpuMaskBackend(cuda))
    ref = MR.complete_mask(G["img"].copy(), G["mask"].copy(), quads, refine=OD.refine_mask, bilateral=OI.bilateral_filter_u8)
    assert got is not None and got.shape == ref.shape and set(np.unique(got)) <= {0, 255}
    assert (got != ref).mean() < 2e-3, (got != ref).mean()
    region = type("Region", (), {"lines": G["lines"]})()
    d1 = MR.dispatch_sync([region], G["img"].copy(), G["mask"].copy())  # default backend
    d2 = MR.dispatch_sync([region], G["img"].copy(), G["mask"].copy(), refine=OD.refine_mask, bilateral=OI.bilateral_filter_u8)
    assert d1.shape == G["mask"].shape and (d1 != d2).mean() < 2e-3


def test_gpu_tail_of_complete_mask_is_bit_identical_to_the_host_tail(cuda):
    """The per-line elliptical dilations, their union, the closing dilation and the resize + binarisation back to page size on the
    device (mit_mask_dilate_jobs, mit_binarize_u8) against the same steps on the host (scipy maximum_filter with
    cv2.getStructuringElement's ellipse rows, pinned to the reference's Python by tests/golden/mask_refinement.npz): same bytes —
    on the golden scene and on a bench-like page (32 lines, dilation sizes 13..23, windows overlapping and touching the borders)."""
    from manga_image_translator_amd import mask_refinement as MR, synth
    from manga_image_translator_amd.textline import Quadrilateral

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_refinement.npz"))
    dev_be, host_be = MR.GpuMaskBackend(cuda), MR.GpuMaskBackend(cuda, gpu_tail=False)
    quads = [Quadrilateral(l.astype(np.float64), "", 0) for l in G["lines"]]
    for off, ks in ((0, 3), (20, 3), (7, 5)):
        a = MR.complete_mask(G["img"].copy(), G["mask"].copy(), quads, dilation_offset=off, kernel_size=ks, backend=dev_be)
        b = MR.complete_mask(G["img"].copy(), G["mask"].copy(), quads, dilation_offset=off, kernel_size=ks, backend=host_be)
        assert a is not None and a.dtype == np.uint8 and np.array_equal(a, b), (off, ks, int((a != b).sum()))
    H, W = 1024, 728
    page, boxes, mask = synth.synth_page(3, H, W, n_boxes=32)
    boxes = np.asarray(boxes).copy()
    boxes[0][:, 0] = np.clip(boxes[0][:, 0] + (W - boxes[0][:, 0].max()), 0, W)     # one line flush with the right border
    boxes[1][:, 1] = np.clip(boxes[1][:, 1] + (H - boxes[1][:, 1].max()), 0, H)     # one flush with the bottom
    rng = np.random.default_rng(5)
    raw = np.zeros((H, W), np.uint8)
    for q in boxes:                                                                   # glyph-like blobs inside every line box
        x0, y0, x1, y1 = int(q[:, 0].min()), int(q[:, 1].min()), int(q[:, 0].max()), int(q[:, 1].max())
        for _ in range(max((x1 - x0) * (y1 - y0) // 300, 6)):
            sx, sy = int(rng.integers(x0, max(x1 - 6, x0 + 1))), int(rng.integers(y0, max(y1 - 8, y0 + 1)))
            raw[sy:sy + int(rng.integers(3, 9)), sx:sx + int(rng.integers(2, 7))] = 255
    region = type("Region", (), {"lines": boxes.astype(np.int64)})()
    for off in (0, 20):
        d1 = MR.dispatch_sync([region], page, raw.copy(), dilation_offset=off, backend=dev_be)
        d2 = MR.dispatch_sync([region], page, raw.copy(), dilation_offset=off, backend=host_be)
        assert d1.shape == (H, W) and set(np.unique(d1)) <= {0, 255} and d1.any() and np.array_equal(d1, d2), int((d1 != d2).sum())
