"""``default`` detector network parity: HIP engine vs the CPU oracle restatement of TextDetection + DBHead.

Tolerance: 36 backbone convs + 30 decoder convs in fp32: maps after sigmoid at 2e-4 absolute (observed ~1e-5); the
thresholded bitmap ``db[:, 0] > text_threshold`` (dbnet_utils, text_threshold default 0.5) must agree outside a 2e-4 margin."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (2, 256, 512)])
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
