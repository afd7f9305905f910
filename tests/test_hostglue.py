"""Detector box extraction on the host (native C++ in libmit_hip.so, no GPU): against closed-form cases and against the
independent scipy-based oracle.  The OpenCV / pyclipper / shapely routines the reference calls are not installed
anywhere this can run, so this pins the two restatements to each other and to geometry that can be worked out by hand."""
import numpy as np
import pytest

from manga_image_translator_amd import hostglue as HG
from oracle import hostglue as OH


def _blobs(rng, H, W, n):
    pred = np.zeros((H, W), np.float32)
    for _ in range(n):
        h, w = int(rng.integers(3, 24)), int(rng.integers(3, 40))
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        yy, xx = np.mgrid[0:h, 0:w]
        kind = rng.integers(0, 3)
        if kind == 0:
            m = np.ones((h, w), bool)
        elif kind == 1:
            m = ((yy - h / 2) / (h / 2)) ** 2 + ((xx - w / 2) / (w / 2)) ** 2 <= 1.0
        else:
            m = (np.abs(yy - h / 2) * w + np.abs(xx - w / 2) * h) <= h * w / 2  # diamond
        pred[y:y + h, x:x + w][m] = rng.uniform(0.35, 1.0)
        if h > 8 and w > 8 and rng.random() < 0.5:
            pred[y + h // 3:y + h // 2 + 1, x + w // 3:x + w // 2 + 1] = rng.uniform(0.0, 0.25)  # a hole
    return pred + rng.uniform(0, 0.02, size=pred.shape).astype(np.float32)


def test_axis_aligned_rectangle_closed_form():
    pred = np.zeros((100, 120), np.float32)
    pred[20:40, 30:90] = 0.9            # pixel centres x 30..89, y 20..39 -> 59 x 19 box
    boxes, scores = HG.ctd_boxes(pred[None, None], 100, 120)
    assert len(boxes) == 1 and scores[0] == pytest.approx(0.9)
    d = 59 * 19 * 1.5 / (2 * (59 + 19))  # area * ratio / perimeter
    x0, y0, x1, y1 = 30 - d, 20 - d, 89 + d, 39 + d
    exp = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
    assert np.abs(boxes[0] - exp).max() <= 1.0  # integer polygon offset + rounding
    # destination scaling (boxes are mapped to the page size)
    b2, _ = HG.boxes_from_bitmap(pred, 0.3, 240, 300, unclip_ratio=1.5, min_sside=2.0)
    assert np.abs(b2[0] - exp * np.array([2.0, 3.0])).max() <= 3.0


def test_counts_holes_and_order():
    pred = np.zeros((50, 60), np.float32)
    pred[5:20, 5:30] = 0.8
    pred[9:15, 10:20] = 0.0   # hole -> its own contour
    pred[30:45, 35:55] = 0.6
    pred[2, 50] = 0.9          # isolated pixel: sside 0 -> skipped slot
    assert HG.contour_count(pred > 0.3)[0] == OH_count(pred > 0.3) == 4
    boxes, scores = HG.ctd_boxes(pred[None, None], 50, 60)
    assert len(boxes) == 4
    assert boxes[0].sum() > 0 and boxes[0][:, 1].min() >= 20          # last found first: the lower rectangle
    assert not boxes[3].any() and scores[3] == 0                      # the isolated pixel (found first) comes last, skipped
    assert scores[2] == pytest.approx((15 * 25 - 6 * 10) * 0.8 / (15 * 25), rel=1e-5)  # outer border: hole filled in the mean


def OH_count(bitmap):
    regs = OH._contours_as_regions(bitmap)
    return len(regs)


def test_contour_count_matches_oracle():
    rng = np.random.default_rng(1)
    for _ in range(5):
        pred = _blobs(rng, 90, 130, 12)
        n, pts = HG.contour_count(pred > 0.3)
        assert n == OH_count(pred > 0.3) and pts >= n


@pytest.mark.parametrize("mode", ["ctd", "default"])
def test_native_matches_oracle_on_random_blobs(mode):
    rng = np.random.default_rng(7)
    for trial in range(6):
        pred = _blobs(rng, 96, 128, 14)
        kw = dict(unclip_ratio=1.5, min_sside=2.0) if mode == "ctd" else dict(unclip_ratio=2.3, min_sside=3.0, box_thresh=0.5,
                                                                              min_sside_out=5.0, roll_start=True)
        nb, ns = HG.boxes_from_bitmap(pred, 0.3, 256, 192, **kw)
        ob, os_ = OH.boxes_from_bitmap(pred, 0.3, 256, 192, **kw)
        assert nb.shape == ob.shape and nb.shape[0] >= 8
        assert np.array_equal(ns == 0, os_ == 0), "different contours skipped"
        assert np.allclose(ns, os_, atol=1e-5)
        keep = np.nonzero(ns != 0)[0]
        same = 0
        for i in keep:
            d = np.abs(nb[i] - ob[i]).max()
            if d <= 2:  # float32 vs float64 calipers + rounding
                same += 1
                continue
            # mirror-symmetric blobs (ellipses, diamonds) have two minimal rectangles of equal area; either is a valid
            # cv2.minAreaRect answer, so only the rectangle's centre and side lengths must agree
            side = lambda b: sorted([np.linalg.norm(b[1] - b[0]), np.linalg.norm(b[2] - b[1])])
            assert np.abs(nb[i].mean(0) - ob[i].mean(0)).max() <= 2 and np.allclose(side(nb[i]), side(ob[i]), atol=3), (i, nb[i], ob[i])
        assert same >= 0.8 * len(keep)


def test_bad_input():
    with pytest.raises(ValueError):
        HG.boxes_from_bitmap(np.zeros((2, 3, 4), np.float32), 0.3, 10, 10, unclip_ratio=1.5, min_sside=2)
    b, s = HG.ctd_boxes(np.zeros((1, 2, 16, 16), np.float32), 16, 16)
    assert b.shape == (0, 4, 2) and s.shape == (0,)


def test_refine_mask_matches_reference_fixture():
    """hostglue.refine_mask / enlarge_window vs the reference's own textmask.py executed with the cv2 stand-in
    (tests/golden/refine_mask.npz): bit-identical uint8 masks in both refine modes."""
    import os

    from manga_image_translator_amd import textline as TL

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refine_mask.npz"))
    quads = [TL.Quadrilateral(q) for q in g["quads"]]
    for mode, key in ((None, "out_none"), (0, "out_inpaint")):
        got = HG.refine_mask(g["page"], g["pred"].copy(), quads, mode)
        assert got.dtype == np.uint8 and np.array_equal(got, g[key]), key
    assert 0 < (g["out_none"] > 0).sum() < (g["out_inpaint"] > 0).sum() < (g["pred"] > 60).sum()
    assert HG.enlarge_window([10, 10, 50, 30], 320, 384) == [2, 2, 58, 38]  # d^2 + 60 d - 1200 = 0 -> d = 15.8, delta = round(d / 2) = 8
    assert HG.enlarge_window([0, 5, 40, 25], 320, 384)[0] == 0  # clipped at the page border


def test_resize_linear_u8_matches_oracle_and_box_case():
    from oracle import ctd as OC

    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    for dsize in ((106, 74), (80, 20), (53, 37), (19, 91)):
        assert np.array_equal(HG.resize_linear_u8(src, dsize), OC.resize_linear_u8(src[..., None], dsize)[..., 0]), dsize
    rgb = rng.integers(0, 256, size=(40, 60, 3), dtype=np.uint8)
    assert np.array_equal(HG.resize_linear_u8(rgb, (30, 20)), OC.resize_linear_u8(rgb, (30, 20)))  # exact 2x: box mean
    assert np.array_equal(HG.resize_linear_u8(src, (53, 37)), src)


def test_in_range_follows_opencv_scalar_bounds():
    """cv2.inRange on 8-bit data with float scalar bounds (textmask.py:68 passes np.histogram edges): the bounds become
    int32 through cvRound (round half to even) and saturate; inverted / out-of-range intervals select nothing."""
    from oracle import ref_import as R

    g = np.arange(256, dtype=np.uint8).reshape(16, 16)
    shim = R.cv2_shim().inRange if R.available() else None
    for lo, hi, first, last in [(99.4, 159.4, 99, 159), (99.6, 159.6, 100, 160), (98.5, 160.5, 98, 160), (99.5, 161.5, 100, 162),
                                (-31.2, 28.8, 0, 29), (200.0, 255.0, 200, 255), (225.7, 285.7, 226, 255)]:
        m = HG._in_range_u8(g, lo, hi)
        sel = np.flatnonzero(m.reshape(-1))
        assert sel[0] == first and sel[-1] == last and len(sel) == last - first + 1, (lo, hi, sel[0], sel[-1])
        if shim is not None:
            assert np.array_equal(shim(g, lo, hi), m)
    assert not HG._in_range_u8(g, 120.0, 60.0).any() and not HG._in_range_u8(g, 256.2, 300.0).any() and not HG._in_range_u8(g, -50.0, -0.6).any()


def test_native_otsu_from_histograms_matches_the_python_recurrence():
    """mit_otsu_from_hist (the GPU refine_mask's host half) == hostglue._otsu_threshold on the image the histogram came from,
    including constant and two-level images."""
    from manga_image_translator_amd import hostglue as HG, lib

    L = lib.load()
    rng = np.random.default_rng(0)
    imgs = [np.full((9, 9), 77, np.uint8), np.where(rng.random((20, 30)) < 0.3, 10, 240).astype(np.uint8)]
    imgs += [rng.normal(rng.integers(40, 200), rng.integers(5, 60), (50, 70)).clip(0, 255).astype(np.uint8) for _ in range(12)]
    hist = np.stack([np.bincount(i.reshape(-1), minlength=256) for i in imgs]).astype(np.int32)
    out = np.zeros(len(imgs), np.int32)
    lib.check(L.mit_otsu_from_hist(hist.ctypes.data, len(imgs), out.ctypes.data), "mit_otsu_from_hist")
    assert out.tolist() == [HG._otsu_threshold(i) for i in imgs]
    assert L.mit_otsu_from_hist(None, 1, None) != 0


def test_refine_mask_gpu_host_logic_against_refine_mask():
    """The host halves of the GPU refine_mask (hostglue.refine_candidates / refine_merge_order) with the three device phases
    emulated in numpy — per-line histograms, candidate xor sums, merge_mask_list over the ordered candidates — reproduce
    hostglue.refine_mask (the routine pinned to the reference's textmask.py) byte for byte.  The device kernels themselves are
    checked against the same routine in tests/test_ctd_refine_gpu.py."""
    from manga_image_translator_amd import hostglue as HG, textline as TL

    rng = np.random.default_rng(3)
    H, W = 220, 300
    page = np.full((H, W, 3), 235, np.uint8) - rng.integers(0, 25, (H, W, 3)).astype(np.uint8)
    pred = np.zeros((H, W), np.float32)
    quads = []
    for (x0, y0, bw, bh) in ((20, 15, 120, 40), (150, 80, 130, 60), (10, 150, 200, 50)):
        for _ in range(bw // 8):
            sx, sy = x0 + int(rng.integers(0, bw - 8)), y0 + int(rng.integers(0, bh - 10))
            page[sy:sy + int(rng.integers(3, 10)), sx:sx + int(rng.integers(2, 7))] = int(rng.integers(0, 80))
        pred[y0:y0 + bh, x0:x0 + bw] = rng.uniform(0.6, 1.0)
        quads.append(TL.Quadrilateral(np.array([[x0, y0], [x0 + bw, y0], [x0 + bw, y0 + bh], [x0, y0 + bh]], np.float64)))
    pred = (np.clip(pred + rng.normal(0, 0.05, (H, W)), 0, 1) * 255).astype(np.uint8)
    ref = HG.refine_mask(page, pred, quads, None)
    assert ref.any()

    wins = [HG.enlarge_window(q.xyxy, W, H) for q in quads]
    crops = [(np.ascontiguousarray(page[b:d, a:c]), np.ascontiguousarray(pred[b:d, a:c])) for a, b, c, d in wins]
    hist = np.zeros((len(wins), 4, 256), np.int32)                       # phase A: what mit_ctd_refine_hist returns
    for i, (im, msk) in enumerate(crops):
        grey = HG._gray_bgr2gray(im)
        hist[i, 0] = np.bincount(grey[HG._erode(msk, HG._RECT3) > 127], minlength=256)
        for ch in range(3):
            hist[i, 1 + ch] = np.bincount(im[..., ch].reshape(-1), minlength=256)
    raw = HG.refine_candidates(hist)

    def cand_mask(im, kind, lo, hi):
        if kind == 1:
            g = HG._gray_bgr2gray(im)
            return np.where((g >= lo) & (g <= hi), 255, 0).astype(np.uint8)
        return np.where(im[..., kind - 2] > lo, 255, 0).astype(np.uint8)

    sums = np.zeros((len(wins), 6), np.uint64)                           # phase B: what mit_ctd_refine_scores returns
    for i, (im, msk) in enumerate(crops):
        for k, (kind, lo, hi, _) in enumerate(raw[i]):
            if kind:
                sums[i, k] = np.bitwise_xor(cand_mask(im, kind, lo, hi), msk).sum(dtype=np.uint64)
    ordered = HG.refine_merge_order(raw, sums, [m.size for _, m in crops])
    out = np.zeros_like(pred)                                            # phase C: merge_mask_list over the ordered candidates
    for (a, b, c, d), (im, msk), cands in zip(wins, crops, ordered):
        masks = []
        for slot, (kind, lo, hi, inv) in enumerate(cands):
            m = cand_mask(im, kind, lo, hi)
            masks.append([255 - m if inv else m, slot])
        out[b:d, a:c] |= HG._merge_mask_list(masks, msk, False)
    assert np.array_equal(out, ref)


def _rows(boxes, scores):
    return sorted((tuple(map(tuple, b.tolist())), round(float(s), 6)) for b, s in zip(boxes, scores) if s > 0)


def test_native_box_extraction_against_the_references_own_flow():
    """csrc/hostglue.hip (contours -> min-area rectangle -> score -> round offset -> rectangle -> scale) against the reference's OWN
    SegDetectorRepresenter Python of both detectors, executed by oracle/make_golden.py with stand-ins for cv2 / pyclipper / shapely
    (tests/golden/boxes.npz): same boxes (integer-exact), same scores, same skipped contours — and, row for row, the same order."""
    import os

    from manga_image_translator_amd import hostglue as HG

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "boxes.npz"))
    for i in range(3):
        pred, (dh, dw) = G[f"pred{i}"], G[f"dest{i}"]
        lm = np.stack([pred, pred])[None]
        b, s = HG.ctd_boxes(lm, int(dh), int(dw))
        assert b.shape == G[f"ctd_boxes{i}"].shape and np.array_equal(b, G[f"ctd_boxes{i}"]), i
        assert np.abs(s - G[f"ctd_scores{i}"]).max() < 1e-6
        assert any(sc == 0 for sc in s) or i == 2          # the scenes exercise the size filter
        for j, (tt, bt, ur) in enumerate(G["dbnet_params"]):
            b2, s2 = HG.dbnet_boxes(lm, int(dh), int(dw), float(tt), float(bt), float(ur))
            assert _rows(b2, s2) == _rows(G[f"dbnet_boxes{i}_{j}"], G[f"dbnet_scores{i}_{j}"]), (i, j)
