"""refine_mask of the ctd detector on the GPU (csrc/ctd_refine.hip + hostglue.refine_mask_gpu) against the host routine
hostglue.refine_mask, which is itself pinned to the reference's ctd_utils/textmask.py (tests/golden/refine_mask.npz).
Integer work: the two must agree byte for byte."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _text_page(seed, H, W, n_lines):
    """A page with stroke-like glyphs of varying tone inside text boxes, speckle noise, and a network-like soft mask."""
    from manga_image_translator_amd import textline as TL

    rng = np.random.default_rng(seed)
    page = np.full((H, W, 3), 235, np.uint8) - rng.integers(0, 25, (H, W, 3)).astype(np.uint8)
    page[:, W // 2:] = (page[:, W // 2:] * 0.8).astype(np.uint8)
    pred = np.zeros((H, W), np.float32)
    quads = []
    for i in range(n_lines):
        bw, bh = int(rng.integers(30, max(W // 2, 40))), int(rng.integers(12, max(H // 6, 16)))
        x0, y0 = int(rng.integers(0, W - bw)), int(rng.integers(0, H - bh))
        tone = int(rng.integers(0, 90))
        for _ in range(max(bw // 10, 2)):       # strokes
            sx, sy = x0 + int(rng.integers(0, bw - 3)), y0 + int(rng.integers(0, bh - 3))
            ex, ey = min(sx + int(rng.integers(2, 9)), x0 + bw), min(sy + int(rng.integers(2, 12)), y0 + bh)
            page[sy:ey, sx:ex] = tone + rng.integers(0, 12)
            if rng.random() < 0.2:              # a hole inside a stroke
                page[sy + 1:ey - 1, sx + 1:ex - 1] = 230
        pred[y0:y0 + bh, x0:x0 + bw] = np.maximum(pred[y0:y0 + bh, x0:x0 + bw], rng.uniform(0.55, 1.0))
        quads.append(TL.Quadrilateral(np.array([[x0, y0], [x0 + bw, y0], [x0 + bw, y0 + bh], [x0, y0 + bh]], np.float64)))
    page[rng.random((H, W)) < 0.01] = 20        # specks: one- and two-pixel components
    pred = np.clip(pred + rng.normal(0, 0.08, (H, W)), 0, 1)
    return page, (pred * 255).astype(np.uint8), quads


@pytest.mark.parametrize("seed,H,W,n", [(0, 300, 400, 6), (1, 512, 360, 12), (2, 64, 80, 2), (3, 1024, 728, 24)])
def test_refine_mask_gpu_equals_host(cuda, seed, H, W, n):
    from manga_image_translator_amd import hostglue as HG

    page, pred, quads = _text_page(seed, H, W, n)
    ref = HG.refine_mask(page, pred, quads, None)
    got = HG.refine_mask_gpu(torch.from_numpy(page).to(cuda), torch.from_numpy(pred).to(cuda), quads, None).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert ref.any(), "scene produced an empty mask: the comparison would be vacuous"
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {ref.size} bytes differ"
    again = HG.refine_mask_gpu(torch.from_numpy(page).to(cuda), torch.from_numpy(pred).to(cuda), quads, None).cpu().numpy()
    assert np.array_equal(again, got)


def test_refine_mask_gpu_edge_cases(cuda):
    from manga_image_translator_amd import hostglue as HG, textline as TL

    page, pred, quads = _text_page(5, 200, 260, 4)
    pd, md = torch.from_numpy(page).to(cuda), torch.from_numpy(pred).to(cuda)
    assert not HG.refine_mask_gpu(pd, md, [], None).any()                       # no lines: empty mask
    q = lambda x0, y0, x1, y1: TL.Quadrilateral(np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]], np.float64))
    edge = [q(0, 0, 40, 20), q(220, 180, 260, 200), q(100, 90, 103, 92), quads[0], quads[0]]   # page corners, a 3x2 box, a repeated line
    assert np.array_equal(HG.refine_mask_gpu(pd, md, edge, None).cpu().numpy(), HG.refine_mask(page, pred, edge, None))
    flat = np.full_like(page, 128)                                               # constant crop: single-level histograms, Otsu = 0
    zero = np.zeros_like(pred)
    for pg, pm in ((flat, pred), (page, zero), (flat, zero), (page, np.full_like(pred, 255))):
        got = HG.refine_mask_gpu(torch.from_numpy(pg).to(cuda), torch.from_numpy(pm).to(cuda), quads, None).cpu().numpy()
        assert np.array_equal(got, HG.refine_mask(pg, pm, quads, None))
    with pytest.raises(NotImplementedError):
        HG.refine_mask_gpu(pd, md, quads, 0)


def test_ctd_plugin_uses_the_gpu_refine(cuda):
    """HipComicTextDetector end to end: the refined mask it returns equals the host routine run on the same intermediate maps."""
    import asyncio

    from manga_image_translator_amd import hostglue as HG, imgproc, pipeline, plugins as P, synth

    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    w = pipeline.synthetic_weights(dict_size=64)
    page = synth.synth_page(4, 512, 384, n_boxes=5)[0]
    det = P.HipComicTextDetector(weights=w)
    run(det.load("cuda"))
    tls, mask, _ = run(det.infer(page, 1024, 0.3, 0.6, 1.5))
    mask_u8, lines, _ = det.engine.forward(torch.from_numpy(page).to(cuda)[None])
    full = imgproc.resize_u8(mask_u8[:1].contiguous(), (384, 512))[0].cpu().numpy()
    assert np.array_equal(mask, HG.refine_mask(page, full, tls, None))
    run(det.unload())
