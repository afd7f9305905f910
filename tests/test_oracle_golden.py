"""The CPU oracle vs the committed golden vectors (tests/golden/*.npz, produced by oracle/make_golden.py from the
REFERENCE's own modules).  Runs anywhere (no /root/reference, no GPU).

Tolerances: both sides are fp32 ATen on a CPU, but thread count / ISA may differ between the machine that made the
fixture and this one, so floats are compared at 2e-5 absolute on O(1) activations (observed: bit-equal to 1e-6);
integer outputs (MPE maps, token ids, canonical corner order, rectified crops) must be identical.
"""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.mark.parametrize("name,mpe", [("lama_mpe.npz", True), ("lama_large.npz", False)])
def test_lama_oracle_matches_reference_fixture(name, mpe):
    from manga_image_translator_amd import lama_schema, synth
    from oracle import lama as OL

    g = _load(name)
    nb = int(g["n_blocks"])
    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(nb))
    mpe_sd = synth.synth_state_dict(lama_schema.lama_mpe_schema()) if mpe else None
    taps = {}
    OL.infer(sd, mpe_sd, g["page"], g["mask"], nb, taps)
    err = np.abs(taps["out_float"].numpy() - g["out_float"]).max()
    assert err < 2e-5, err
    if mpe:
        mk = (g["mask"].astype(np.float32) / 255.0 >= 0.5).astype(np.float32)
        rel, _, direct = OL.load_masked_position_encoding(mk)
        assert np.array_equal(rel, g["rel_pos"]) and np.array_equal(direct, g["direct"])
        assert rel.max() > 0 and direct.any()


def test_lama_oracle_resize_path_matches_the_reference_infer():
    """oracle.lama.infer with the resize legs (resize_keep_aspect, x8 INTER_LINEAR and back, composite with the original mask)
    against the output of the reference's own LamaMPEInpainter._infer run over the cv2 stand-in (bytes; at most a truncation
    flip of +-1 where the float sits on an integer)."""
    from manga_image_translator_amd import lama_schema, synth
    from oracle import lama as OL

    g = _load("lama_resize.npz")
    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(9))
    mpe_sd = synth.synth_state_dict(lama_schema.lama_mpe_schema())
    for tag in ("a", "b"):
        got = OL.infer(sd, mpe_sd, g[f"page_{tag}"], g[f"mask_{tag}"], 9, None, int(g[f"size_{tag}"]))
        d = np.abs(got.astype(np.int32) - g[f"out_{tag}"].astype(np.int32))
        assert got.shape == g[f"page_{tag}"].shape and d.max() <= 1 and (d != 0).mean() < 1e-3, (tag, d.max(), (d != 0).mean())
        changed = (got != g[f"page_{tag}"]).any(-1)
        assert changed.any() and not changed[g[f"mask_{tag}"] < 127].any()  # only pixels under the original mask may change


def test_ctd_oracle_matches_reference_fixture():
    from manga_image_translator_amd import ctd_schema as S, synth
    from oracle import ctd as OC

    g = _load("ctd.npz")
    gain = S.CTD_GAIN
    ysd = synth.synth_state_dict(S.yolo_schema(), gain=gain)
    ssd = synth.synth_state_dict(S.unet_head_schema(), gain=gain)
    dsd = synth.synth_state_dict(S.db_head_schema(), gain=gain)
    x, ratio, dw, dh = OC.preprocess_img(g["page"], input_size=(128, 128))
    assert (dw, dh) == (int(g["dw"]), int(g["dh"]))
    assert np.array_equal(x.numpy(), g["net_in"]), "letterbox / normalisation differs from the reference's preprocess"
    with torch.no_grad():
        mask, lines = OC.textdet_forward(ysd, ssd, dsd, x)
    assert np.abs(mask.numpy() - g["mask"]).max() < 2e-5
    assert np.abs(lines.numpy() - g["lines"]).max() < 2e-5
    assert 0.02 < g["mask"].mean() < 0.98 and g["lines"].std() > 0.01  # the fixture is not a saturated constant


def test_ocr_oracle_matches_reference_fixture():
    from manga_image_translator_amd import ocr_schema, synth
    from oracle import ocr48 as OO

    g = _load("ocr48.npz")
    D, T = int(g["dict_size"]), int(g["T"])
    sd = synth.synth_state_dict(ocr_schema.ocr48_schema(D))
    widths = g["widths"].tolist()
    img = ((torch.from_numpy(g["region"]).float() - 127.5) / 127.5).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        bb = OO.backbone(sd, img).squeeze(2).permute(0, 2, 1).numpy()
        res = OO.infer_beam_batch_tensor(sd, img, widths, max_seq_length=T)
    assert np.abs(bb - g["backbone"]).max() < 2e-5 * max(1.0, np.abs(g["backbone"]).max())
    for i, (idx, prob, fg, bg, fgi, bgi) in enumerate(res):
        n = int(g["length"][i])
        assert idx.tolist() == g["tokens"][i, :n].tolist()
        assert abs(prob - g["prob"][i]) < 1e-4 * g["prob"][i]
        col = torch.cat([fg, bg, fgi, bgi], dim=-1).numpy()
        assert np.abs(col - g["colors"][i, :n]).max() < 1e-4


def test_textline_oracle_matches_reference_fixture():
    from oracle import textline as OT

    g = _load("textline.npz")
    for k in range(len(g["quads"])):
        sp, vert = OT.sort_pnts(g["quads"][k])
        assert np.array_equal(sp, g["sorted_pts"][k])
        d = "v" if vert else "h"
        assert d == str(g["direction"][k])
        crop = OT.get_transformed_region(g["image"], sp, d, 48)
        w = int(g["crop_width"][k])
        assert crop.shape == (48, w, 3)
        assert np.array_equal(crop, g["crops"][k, :, :w])


def test_esrgan_oracle_matches_reference_fixture():
    from manga_image_translator_amd import esrgan_schema, synth
    from oracle import esrgan as OE

    g = _load("esrgan.npz")
    nb = int(g["nb"])
    sd = synth.synth_state_dict(esrgan_schema.rrdbnet_schema(nb))
    x = torch.from_numpy(g["page"][:, :, ::-1].copy()).float().div(255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        y = OE.rrdbnet_forward(sd, x, nb)
    assert np.abs(y.numpy() - g["out_float"]).max() < 2e-5
    out = OE.infer(sd, g["page"], nb)
    d = np.abs(out.astype(np.int32) - g["out_u8"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3  # identical up to a x255 truncation boundary
    assert 0.05 < (g["out_u8"] == 0).mean() < 0.95 and len(np.unique(g["out_u8"])) > 100  # not a saturated constant


def test_ocr_ctc_oracle_matches_reference_fixture():
    from manga_image_translator_amd import ocr_ctc_schema as S, synth
    from oracle import ocr_ctc as OC

    g = _load("ocr_ctc.npz")
    sd = synth.synth_state_dict(S.ocr_ctc_schema(int(g["dict_size"])), gain=S.CTC_GAIN)
    img = ((torch.from_numpy(g["region"]).float() - 127.5) / 127.5).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        logits, colors = OC.forward(sd, img)
    assert np.abs(logits.numpy() - g["logits"]).max() < 5e-5 * np.abs(g["logits"]).max()
    assert np.abs(colors.numpy() - g["colors"]).max() < 5e-5 * max(1.0, np.abs(g["colors"]).max())
    dec = OC.decode_ctc_top1(logits, colors)
    for i, line in enumerate(dec):
        n = int((g["ids"][i] >= 0).sum())
        assert [t[0] for t in line] == g["ids"][i, :n].tolist() and n >= 5
        assert np.allclose(np.array([t[1:] for t in line]), g["vals"][i, :n], atol=1e-4)
    # chunk geometry (model_48px_ctc.py:84)
    crops = [g["region"][i, :, :w] for i, w in enumerate(g["widths"])]
    idx, ws, t = next(OC.make_chunks(crops))
    assert t.shape[3] == max(ws) + 7 + 128 == g["region"].shape[2]


def test_dbnet_oracle_matches_reference_fixture():
    from manga_image_translator_amd import dbnet_schema, synth
    from oracle import dbnet as OD

    g = _load("dbnet.npz")
    sd = synth.synth_state_dict(dbnet_schema.text_detection_schema(), gain=1.2)
    db, mask = OD.det_batch_forward(sd, g["page"][None])
    assert np.abs(db[:, :, 96:160, 96:160] - g["db_f32_crop"]).max() < 2e-5
    assert np.abs(mask[:, :, 32:96, 32:96] - g["mask_f32_crop"]).max() < 2e-5
    assert np.abs(db - g["db"].astype(np.float32)).max() < 1e-3 and np.abs(mask - g["mask"].astype(np.float32)).max() < 1e-3  # fp16 maps
    assert g["db"].astype(np.float32).std() > 0.1 and g["mask"].astype(np.float32).std() > 0.1
