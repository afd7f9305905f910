"""``default`` detector network parity: HIP engine vs the CPU oracle restatement of TextDetection + DBHead.

Tolerance: 36 backbone convs + 30 decoder convs in fp32: maps after sigmoid at 2e-4 absolute (observed ~1e-5); the
thresholded bitmap ``db[:, 0] > text_threshold`` (dbnet_utils, text_threshold default 0.5) must agree outside a 2e-4 margin."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (2, 256, 512), (1, 1024, 1024)])  # 1024^2: BASELINE config 1's page before its 2x upscale
def test_dbnet_parity(cuda, B, H, W):
    from manga_image_translator_amd import dbnet, dbnet_schema, synth
    from oracle import dbnet as OD

    sd = synth.synth_state_dict(dbnet_schema.text_detection_schema(), gain=1.2)
    eng = dbnet.DbnetEngine(sd, device=cuda)
    pages = np.stack([synth.synth_page(30 + i, H, W, n_boxes=4)[0] for i in range(B)])
    db, mask = eng.forward(torch.from_numpy(pages).to(cuda))
    torch.cuda.synchronize()
    rdb, rmask = OD.det_batch_forward(sd, pages)
    assert tuple(db.shape) == rdb.shape and tuple(mask.shape) == (B, H // 2, W // 2)
    e1 = np.abs(db.cpu().numpy() - rdb).max()
    e2 = np.abs(mask.cpu().numpy() - rmask[:, 0]).max()
    assert e1 < 2e-4 and e2 < 2e-4, (e1, e2)
    near = np.abs(rdb[:, 0] - 0.5) < 2e-4
    assert np.array_equal((db[:, 0].cpu().numpy() > 0.5)[~near], (rdb[:, 0] > 0.5)[~near])
    assert rdb[:, 0].std() > 0.05 and rmask.std() > 0.05


def test_dbnet_rejects_bad_input(cuda):
    from manga_image_translator_amd import dbnet, dbnet_schema, synth

    eng = dbnet.DbnetEngine(synth.synth_state_dict(dbnet_schema.text_detection_schema(), gain=1.2), device=cuda)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 200, 256, 3, dtype=torch.uint8, device=cuda))


def test_default_detector_plugin(cuda):
    import asyncio

    from manga_image_translator_amd import dbnet_schema, plugins as P, synth
    from oracle import dbnet as OD

    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    sd = synth.synth_state_dict(dbnet_schema.text_detection_schema(), gain=1.2)
    page = synth.synth_page(8, 256, 200, n_boxes=3)[0]
    seen = {}

    def pre(image, detect_size):  # stands in for bilateralFilter + resize_aspect_ratio: pad 200 -> 256 columns
        canvas = np.zeros((256, 256, 3), np.uint8)
        canvas[:, :200] = image
        return canvas, 1.0, 56, 0

    def boxes(db, h, w, tt, bt, ur):
        seen["db"] = db
        return np.array([[[10, 10], [90, 10], [90, 40], [10, 40]]]), np.array([0.8])

    det = P.HipDefaultDetector(weights=sd, preprocess=pre, boxes_from_maps=boxes, resize2x=lambda m: np.repeat(np.repeat(m, 2, 0), 2, 1))
    run(det.load("cuda"))
    tls, raw_mask, extra = run(det.infer(page, 256, 0.5, 0.7, 2.3))
    assert extra is None and len(tls) == 1 and raw_mask.dtype == np.uint8 and raw_mask.shape == (256, 200)
    canvas, *_ = pre(page, 256)
    rdb, rmask = OD.det_batch_forward(sd, canvas[None])
    assert np.abs(seen["db"] - rdb).max() < 2e-4
    ref_mask = np.clip(np.repeat(np.repeat(rmask[0, 0], 2, 0), 2, 1)[:, :-56] * 255, 0, 255).astype(np.uint8)
    assert np.abs(raw_mask.astype(int) - ref_mask.astype(int)).max() <= 1
    run(det.unload())


@pytest.mark.gpu
def test_default_detector_native_preprocess(cuda):
    """HipDefaultDetector without an injected preprocess: cv2.bilateralFilter(image, 17, 80, 80) + resize_aspect_ratio
    (default.py:62, default_utils/imgproc.py:37-70) run on the device and equal the CPU restatement byte for byte
    (oracle bilateral filter, the oracle's INTER_LINEAR resize, zero canvas to a multiple of 256); the plugin then runs end to end
    with only native pieces (GPU preprocess + network, native host box extraction)."""
    import asyncio

    from manga_image_translator_amd import dbnet_schema, plugins as P, synth
    from oracle import ctd as OC, imgproc as OI

    page = synth.synth_page(9, 300, 210, n_boxes=3)[0]
    got, ratio, pad_w, pad_h = P.default_preprocess_gpu(torch.from_numpy(page).to(cuda), 512)
    assert ratio == 512 / 300 and (pad_w, pad_h) == (154, 0) and tuple(got.shape) == (1, 512, 512, 3)
    ref = np.zeros((512, 512, 3), np.uint8)
    ref[:, :358] = OC.resize_linear_u8(OI.bilateral_filter_u8(page, 17, 80.0, 80.0), (358, 512))
    assert np.array_equal(got[0].cpu().numpy(), ref)
    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    sd = synth.synth_state_dict(dbnet_schema.text_detection_schema(), gain=1.2)
    det = P.HipDefaultDetector(weights=sd)
    run(det.load("cuda"))
    tls, raw_mask, extra = run(det.infer(page, 512, 0.5, 0.7, 2.3))
    assert extra is None and raw_mask.dtype == np.uint8 and raw_mask.shape == (512, 358)
    run(det.unload())
