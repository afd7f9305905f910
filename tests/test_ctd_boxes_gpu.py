"""SegDetectorRepresenter.boxes_from_bitmap on the GPU (csrc/ctd_boxes.hip; db_utils.py:127-216, dbnet_utils.py:97-144) against the host
routine (csrc/hostglue.hip, pinned to the reference's Python by tests/golden/boxes.npz) and against that fixture directly: same boxes
(integer-exact), same order, same skipped slots; scores equal to the float32 bit except where the order of the double summation moves
the 53rd bit across a float32 rounding boundary (bound 1 ulp, asserted)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(got, want, what):
    (gb, gs), (wb, ws) = got, want
    assert gb.shape == wb.shape and gs.shape == ws.shape, (what, gb.shape, wb.shape)
    assert np.array_equal(gb, wb), (what, np.argwhere((gb != wb).any(axis=(1, 2)))[:5].tolist())
    ulp = np.abs(gs.view(np.int32).astype(np.int64) - ws.view(np.int32).astype(np.int64))
    assert ulp.max(initial=0) <= 1, (what, float(np.abs(gs - ws).max()))
    return int((ulp != 0).sum())


def _scene(seed, H, W, n_lines, noise=0.0, holes=True):
    """Text-line-like rotated rectangles with a probability ramp, specks, blobs on the border, holes, optional salt noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    m = (rng.random((H, W)).astype(np.float32) * 0.2)
    for _ in range(n_lines):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        w, h, a = rng.uniform(10, 0.25 * W), rng.uniform(2, 24), rng.uniform(-0.5, 0.5) + (np.pi / 2 if rng.random() < 0.3 else 0)
        c, s = np.cos(a), np.sin(a)
        u, v = (xx - cx) * c + (yy - cy) * s, -(xx - cx) * s + (yy - cy) * c
        inside = (np.abs(u) <= w) & (np.abs(v) <= h)
        m[inside] = np.maximum(m[inside], rng.uniform(0.35, 0.98) - 0.1 * np.abs(v[inside]) / h)
        if holes and rng.random() < 0.4:
            hx, hy = rng.uniform(-w / 2, w / 2), rng.uniform(-h / 2, h / 2)
            m[(np.abs(u - hx) <= rng.uniform(1, 4)) & (np.abs(v - hy) <= rng.uniform(1, 3))] = 0.05
    if noise:
        salt = rng.random((H, W)) < noise
        m[salt] = rng.uniform(0.31, 0.99, int(salt.sum())).astype(np.float32)
    return m.astype(np.float32)


def test_golden_scenes_ctd_and_dbnet_parameters(cuda):
    """The fixture the host routine is pinned with (the reference's own SegDetectorRepresenter over stand-ins for cv2 / pyclipper)."""
    from manga_image_translator_amd import hostglue as HG

    G = np.load(os.path.join(GOLDEN, "boxes.npz"))
    for i in range(3):
        pred = G[f"pred{i}"]
        lm = torch.from_numpy(np.stack([pred, pred])[None]).cuda()              # [1,2,H,W]: channel 0 is a strided view
        dh, dw = (int(v) for v in G[f"dest{i}"])
        got = HG.ctd_boxes_gpu(lm, dh, dw)[0]
        assert np.array_equal(got[0], G[f"ctd_boxes{i}"]) and np.abs(got[1] - G[f"ctd_scores{i}"]).max() < 1e-6, i
        _same(got, HG.ctd_boxes(lm.cpu().numpy(), dh, dw), f"ctd golden {i}")
        for j, (tt, bt, ur) in enumerate(G["dbnet_params"]):
            got = HG.dbnet_boxes_gpu(lm, dh, dw, float(tt), float(bt), float(ur))[0]
            _same(got, HG.dbnet_boxes(lm.cpu().numpy(), dh, dw, float(tt), float(bt), float(ur)), f"dbnet golden {i} {j}")
            keep = got[1] > 0
            assert np.array_equal(got[0][keep], G[f"dbnet_boxes{i}_{j}"][G[f"dbnet_scores{i}_{j}"] > 0])


@pytest.mark.parametrize("H,W,n_lines,noise", [(1024, 728, 32, 0.0), (512, 384, 24, 0.0005), (200, 333, 10, 0.01), (64, 64, 3, 0.2)])
def test_batches_of_text_line_scenes_equal_the_host_routine(cuda, H, W, n_lines, noise):
    """The detector's map size (1024 x 728 for a 2048 x 1456 page) and smaller, denser ones: rotated lines, holes, specks, borders on
    the image frame, salt noise (hundreds of one-pixel borders; at 0.2 more borders than max_candidates)."""
    from manga_image_translator_amd import hostglue as HG

    B = 5
    maps = np.stack([_scene(100 + 7 * b + H, H, W, n_lines, noise) for b in range(B)])
    maps[1, :, :3] = 0.9                                   # a bar on the left frame: borders that start in column 0
    maps[2] = 0.0                                          # an empty page
    maps[3, H // 4:H // 2, W // 4:W // 2] = 0.95           # a big blob ...
    maps[3, H // 4 + 5:H // 2 - 5, W // 4 + 5:W // 2 - 5] = 0.1   # ... that is a ring (one large hole border)
    maps[3, H // 4 + 12:H // 4 + 20, W // 4 + 12:W // 4 + 40] = 0.8  # ... with an island inside the hole
    lines = torch.from_numpy(np.stack([maps, 1 - maps], axis=1)).cuda()
    got = HG.ctd_boxes_gpu(lines, 2 * H, 2 * W)
    flips = 0
    for b in range(B):
        want = HG.ctd_boxes(lines[b:b + 1].cpu().numpy(), 2 * H, 2 * W)
        flips += _same(got[b], want, f"page {b}")
    assert len(got[2][0]) == 0
    assert flips <= 1                                      # score bits that moved (see the module docstring): practically never
    got2 = HG.dbnet_boxes_gpu(lines, 2 * H, 2 * W, 0.3, 0.5, 1.5)
    for b in range(B):
        _same(got2[b], HG.dbnet_boxes(lines[b:b + 1].cpu().numpy(), 2 * H, 2 * W, 0.3, 0.5, 1.5), f"dbnet page {b}")


def test_a_border_longer_than_a_waves_lds_goes_to_the_host_routine(cuda):
    """A comb whose outer border has > 8192 points: the page is flagged and computed by mit_boxes_from_bitmap (same results by
    construction); the other page of the batch stays on the GPU."""
    from manga_image_translator_amd import hostglue as HG, lib as L

    H, W = 600, 700
    comb = np.zeros((H, W), np.float32)
    comb[10:20, 10:690] = 0.9
    for x in range(12, 688, 4):
        comb[20:580, x:x + 2] = 0.9                        # 169 teeth x 2 x 560 px of border
    maps = np.stack([comb, _scene(5, H, W, 12)])
    t = torch.from_numpy(maps).cuda()
    lib = L.load()
    ws = torch.empty(int(lib.mit_boxes_from_bitmap_dev_workspace_bytes(2, H, W, 1000)), dtype=torch.uint8, device="cuda")
    boxes = torch.empty(2, 1000, 4, 2, dtype=torch.int64, device="cuda")
    scores = torch.empty(2, 1000, device="cuda")
    meta = torch.zeros(2, 2, dtype=torch.int32, device="cuda")
    L.check(lib.mit_boxes_from_bitmap_dev(t.data_ptr(), 0, None, 0, 0.3, 2, H, W, W, H, 1000, 1.5, 2.0, 0.0, 0.0, 0, ws.data_ptr(), ws.numel(),
                                          boxes.data_ptr(), scores.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(), None), "dev")
    torch.cuda.synchronize()
    assert meta[1].tolist() == [1, 0] and meta[0, 0].item() == 1
    got = HG.boxes_from_bitmap_gpu(t, 0.3, W, H, unclip_ratio=1.5, min_sside=2.0)
    for b in range(2):
        _same(got[b], HG.boxes_from_bitmap(maps[b], 0.3, W, H, unclip_ratio=1.5, min_sside=2.0), f"page {b}")


def test_plugin_boxes_are_the_host_routines(cuda):
    """HipComicTextDetector._infer takes its boxes from the GPU chain: same text lines as with the host routine injected."""
    import asyncio

    from manga_image_translator_amd import pipeline, plugins as P, synth

    page = synth.synth_page(3, 512, 384, n_boxes=5, disjoint=True)[0]
    w = pipeline.synthetic_weights(dict_size=64)
    lines_seen = {}

    def host_boxes(lines_map, im_h, im_w):
        lines_seen["map"] = lines_map
        return P._native_ctd_boxes(lines_map, im_h, im_w)

    loop = asyncio.new_event_loop()
    outs = []
    for boxes_fn in (None, host_boxes):
        det = P.HipComicTextDetector(weights=w, boxes_from_maps=boxes_fn)
        loop.run_until_complete(det.load("cuda"))
        tls, mask, _ = loop.run_until_complete(det.infer(page, 1024, 0.5, 0.7, 2.3))
        outs.append(([np.asarray(t.pts).tolist() for t in tls], [float(t.prob) for t in tls], mask))
    assert "map" in lines_seen and outs[0][0] == outs[1][0] and np.array_equal(outs[0][2], outs[1][2])
    assert np.allclose(outs[0][1], outs[1][1], rtol=0, atol=1e-7)
