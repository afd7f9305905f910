"""The committed fixtures are reproducible from the reference tree (build container only; skipped on the GPU box).

Re-runs oracle/make_golden.py's generators — the reference's own modules imported from /root/reference — into a
temporary directory and compares with tests/golden/.  Together with test_oracle_golden.py this pins the oracle to the
reference code itself, not to a snapshot of unknown origin."""
import os

import numpy as np
import pytest

from oracle import ref_import as R

pytestmark = pytest.mark.skipif(not R.available(), reason="/root/reference not present")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("gen,files", [("golden_textline", ["textline.npz"]), ("golden_ocr", ["ocr48.npz"]),
                                       ("golden_ctd", ["ctd.npz"]), ("golden_lama", ["lama_mpe.npz", "lama_large.npz"]), ("golden_lama_resize", ["lama_resize.npz"]), ("golden_esrgan", ["esrgan.npz"]), ("golden_ocr_ctc", ["ocr_ctc.npz"]), ("golden_dbnet", ["dbnet.npz"]), ("golden_direction", ["direction.npz"]), ("golden_rearrange", ["rearrange.npz"]), ("golden_refine_mask", ["refine_mask.npz"]), ("golden_mask_refinement", ["mask_refinement.npz"]), ("golden_boxes", ["boxes.npz"]), ("golden_bubble", ["bubble.npz"])])
def test_fixture_regenerates(tmp_path, monkeypatch, gen, files):
    from oracle import make_golden as MG

    monkeypatch.setattr(MG, "GOLDEN", str(tmp_path))
    getattr(MG, gen)()
    for f in files:
        new, old = np.load(tmp_path / f), np.load(os.path.join(GOLDEN, f))
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            a, b = new[k], old[k]
            if a.dtype.kind == "f":
                tol = 2e-3 if a.dtype == np.float16 else 2e-6
                assert np.allclose(a.astype(np.float64), b.astype(np.float64), rtol=0, atol=tol * max(1.0, float(np.abs(b).max()))), (f, k)
            else:
                assert np.array_equal(a, b), (f, k)


def test_textline_merge_fixture_regenerates(tmp_path, monkeypatch):
    """The reference's test vectors + its own merge output, re-extracted from /root/reference, equal the committed JSON."""
    import json
    from oracle import make_golden as MG

    monkeypatch.setattr(MG, "GOLDEN", str(tmp_path))
    MG.golden_textline_merge()
    new = json.load(open(tmp_path / "textline_merge.json"))
    old = json.load(open(os.path.join(GOLDEN, "textline_merge.json")))
    assert new == old and len(new["cases"]) == 11 and all(c["ref_passes_own_test"] for c in new["cases"])


def test_share_stream_fixture_regenerates(tmp_path, monkeypatch):
    """The reference's own ``handle_buffer`` / ``extract_header`` (server/sent_data_internal.py), re-run over the committed stream scene,
    reproduce tests/golden/share_stream.json (which pins serve.parse_frames: tests/test_serve.py)."""
    import json
    from oracle import make_golden as MG

    monkeypatch.setattr(MG, "GOLDEN", str(tmp_path))
    MG.golden_share_stream()
    new, old = json.load(open(tmp_path / "share_stream.json")), json.load(open(os.path.join(GOLDEN, "share_stream.json")))
    assert new == old and len(new["calls"]) == 7 and new["header_of_first_frame"] == [1, 9]


def test_schemas_match_reference_modules():
    """Every synthetic state_dict has exactly the reference modules' parameter names and shapes."""
    from manga_image_translator_amd import ctd_schema, lama_schema, ocr_schema
    from oracle import make_golden as MG

    def check(schema, module_sd):
        mine = {n: tuple(s) for n, s, _ in schema}
        ref = {n: tuple(t.shape) for n, t in module_sd.items()}
        assert mine == ref, (sorted(set(mine) ^ set(ref))[:10])

    m, _, _ = MG.build_ref_lama(9, True)
    check(lama_schema.lama_generator_schema(9), m.generator.state_dict())
    check(lama_schema.lama_mpe_schema(), m.mpe.state_dict())
    m18, _, _ = MG.build_ref_lama(18, False)
    check(lama_schema.lama_generator_schema(18), m18.generator.state_dict())
    model, _ = MG.build_ref_ocr()
    check(ocr_schema.ocr48_schema(MG.OCR_DICT), model.state_dict())
    bm, yolo = R.ctd()
    check(ctd_schema.unet_head_schema(), bm.UnetHead(act="leaky").state_dict())
    check(ctd_schema.db_head_schema(), bm.DBHead(64, act="leaky").state_dict())
    check(ctd_schema.yolo_schema(), yolo.Model(ctd_schema.YOLOV5S_CFG).state_dict())
    from manga_image_translator_amd import esrgan_schema

    from manga_image_translator_amd import ocr_ctc_schema

    ctc, _ = MG.build_ref_ocr_ctc()
    check(ocr_ctc_schema.ocr_ctc_schema(MG.OCR_DICT), ctc.state_dict())
    from manga_image_translator_amd import dbnet_schema

    det, _ = MG.build_ref_dbnet()  # ResNet-34 part: the oracle's own restatement of torchvision's module (unpinned)
    check(dbnet_schema.text_detection_schema(), det.state_dict())
    net, _ = MG.build_ref_esrgan(3)
    check(esrgan_schema.rrdbnet_schema(3), net.state_dict())
