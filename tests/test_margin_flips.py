"""What "thresholded bitmap exact outside a 1e-4 margin of 0.3" means for the BOXES: the detector's float map is
reproduced to 1e-4, so a pixel whose probability lies within 1e-4 of the threshold may come out on either side.  Here every such pixel
is forced above and below the threshold and `SegDetectorRepresenter` (ctd_utils/utils/db_utils.py:127-216, native: mit_ctd_boxes) is
run on each variant: the consequence is stated as numbers, and bounded.

The maps are what a trained head emits for the synthetic bench pages (coupled.synthetic_head_outputs, softened over a few pixels like a
real sigmoid output); beside the pixels that fall inside the margin by themselves, 300 pixels ON THE BOUNDARY of the bitmap per page are
moved into the margin — the worst place for a flip."""
import numpy as np
from scipy import ndimage as ndi

from manga_image_translator_amd import coupled, hostglue, synth

H, W = 2048, 1456


def _boxes(p):
    lines = np.zeros((1, 2) + p.shape, np.float32)
    lines[0, 0] = p
    return hostglue.ctd_boxes(lines, H, W)


def _iou_axis_aligned(a, b):
    ax0, ay0, ax1, ay1 = a[:, 0].min(), a[:, 1].min(), a[:, 0].max(), a[:, 1].max()
    bx0, by0, bx1, by1 = b[:, 0].min(), b[:, 1].min(), b[:, 0].max(), b[:, 1].max()
    iw, ih = max(0, min(ax1, bx1) - max(ax0, bx0)), max(0, min(ay1, by1) - max(ay0, by0))
    return iw * ih / float((ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - iw * ih)


def test_flipping_every_in_margin_pixel_moves_boxes_by_a_bounded_amount():
    natural = moved_up = moved_down = total = 0
    worst_shift, worst_iou = 0, 1.0
    for g in range(4):
        page, quads, _ = synth.synth_page(g, H, W, n_boxes=32, disjoint=True)
        prob, _ = coupled.synthetic_head_outputs(page, quads, (H // 2, W // 2))
        p = ndi.gaussian_filter(prob, 1.5).astype(np.float32)
        bm = p > 0.3
        ring = (bm ^ ndi.binary_erosion(bm)) | (ndi.binary_dilation(bm) ^ bm)      # the bitmap's inner and outer boundary pixels
        ys, xs = np.nonzero(ring)
        pick = np.random.default_rng(g).choice(len(ys), size=300, replace=False)
        p[ys[pick], xs[pick]] = np.float32(0.3) + np.float32(3e-5) * np.where(bm[ys[pick], xs[pick]], 1, -1).astype(np.float32)
        margin = np.abs(p - np.float32(0.3)) < 1e-4
        natural += int(margin.sum()) - 300
        b0, s0 = _boxes(p)
        assert len(b0) == 32
        total += len(b0)
        for sign in (1, -1):
            q = p.copy()
            q[margin] = np.float32(0.3) + np.float32(sign * 2e-4)                   # every in-margin pixel on one side of the threshold
            b1, s1 = _boxes(q)
            assert len(b1) == len(b0), "a margin flip must not create, drop, merge or split a box on these pages"
            d = np.abs(b1.astype(np.int64) - b0.astype(np.int64)).max(axis=(1, 2))
            worst_shift = max(worst_shift, int(d.max()))
            worst_iou = min(worst_iou, min(_iou_axis_aligned(x, y) for x, y in zip(b0, b1)))
            if sign > 0:
                moved_up += int((d > 0).sum())
            else:
                moved_down += int((d > 0).sum())
            assert np.abs(s1 - s0).max() < 0.02                                     # box scores (mean probability inside the box)
    print(f"margin flips: {natural} px inside the margin by themselves + 1200 placed on box boundaries over 4 pages; {total} boxes; "
          f"all above -> {moved_up} boxes moved, all below -> {moved_down}; worst corner shift {worst_shift} page px, worst IoU {worst_iou:.3f}")
    # the stated consequence (measured: 126 of 128 boxes move when ~9 boundary pixels per box flip to foreground, none when they flip to
    # background; worst corner shift 10 px; worst IoU 0.76, on the thinnest boxes): the SET of boxes is stable; a box whose boundary pixels
    # flip keeps its corners within 12 px on a 2048 x 1456 page (one map pixel = 2 page pixels; the min-area rectangle of a side with one
    # extra pixel grows and tilts, unclip x 1.5 scales it) and >= 0.7 IoU with itself.  A map as steep as a trained head's has no pixel
    # inside the margin by itself (0 here); the random-init network of the bench has ~260 per page and finds no box either way.
    assert worst_shift <= 12 and worst_iou >= 0.7
    assert moved_up + moved_down > 0            # the test does exercise flips that matter
