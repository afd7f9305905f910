"""Tall / wide page tiling (rearrange.py) against the reference's own det_rearrange_forward, which produced
tests/golden/rearrange.npz with a deterministic stand-in for the detector network (oracle/make_golden.py:golden_rearrange):
same plan decisions, bit-identical stitched maps (float32 bytes, sha256)."""
import hashlib
import os

import numpy as np
import pytest

from manga_image_translator_amd import imgproc, rearrange as RA, synth
from oracle.make_golden import fake_detector

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rearrange.npz"))


@pytest.mark.parametrize("tag", ["tall", "wide", "shrink"])
def test_rearranged_maps_equal_the_reference(tag):
    H, W = (int(v) for v in G[f"shape_{tag}"])
    tgt = int(G[f"tgt_{tag}"])
    page = synth.synth_page(int(G[f"seed_{tag}"]), H, W, n_boxes=6)[0]
    calls = []

    def net(batch):
        calls.append(batch.shape)
        return fake_detector(batch)

    db, mask = RA.forward(page, net, tgt, resize=lambda a, ds: imgproc.resize_u8_host(a, ds))
    assert all(s[0] <= 4 and s[1:] == (tgt, tgt, 3) for s in calls)
    assert db.dtype == np.float32 and tuple(db.shape) == tuple(G[f"db_shape_{tag}"]) and tuple(mask.shape) == tuple(G[f"mask_shape_{tag}"])
    assert np.array_equal(db[..., ::7, ::5], G[f"db_sample_{tag}"]) and np.array_equal(mask[..., ::7, ::5], G[f"mask_sample_{tag}"])
    assert hashlib.sha256(np.ascontiguousarray(db).tobytes()).hexdigest() == str(G[f"db_sha_{tag}"])
    assert hashlib.sha256(np.ascontiguousarray(mask).tobytes()).hexdigest() == str(G[f"mask_sha_{tag}"])


def test_plan_decisions():
    H, W = (int(v) for v in G["shape_none"])
    assert RA.plan(H, W, int(G["tgt_none"])) is None and RA.forward(np.zeros((H, W, 3), np.uint8), None, int(G["tgt_none"])) == (None, None)
    assert RA.plan(2048, 1456, 1024) is None                      # the BASELINE page is processed whole
    assert RA.plan(3000, 1200, 1024) is None                      # long enough but not thin enough (aspect 2.5)
    assert RA.plan(2400, 300, 1024) is None                       # thin but 2400 / 1024 < 2.5
    pl = RA.plan(12000, 800, 1024)                                # a webtoon strip
    assert pl is not None and not pl.transpose and pl.pw_num == 2 and pl.patch == 1600 and pl.ph_num == 8 and pl.p_num == 4 and pl.pad_num == 0
    assert pl.ph_step == int((12000 - 1600) / 7) and pl.rel_steps[0] == 0.0 and (pl.ph_num - 1) * pl.ph_step + pl.patch <= 12000
    pw = RA.plan(700, 9000, 1024)
    assert pw is not None and pw.transpose and (pw.h, pw.w) == (9000, 700)
    sq, pad = RA.squares(np.zeros((12000, 800, 3), np.uint8), pl, 1024)   # unshrunk squares for a caller that shrinks on the GPU
    assert sq.shape == (4, 1600, 1600, 3) and pad == 0
