"""48px_ctc OCR stage parity: HIP engine vs the CPU oracle restatement of the reference model.

Tolerances: ~60 conv layers + 3 encoder layers in fp32 with the synthetic gain 1.5 (activations up to ~1e2): backbone and
logits at 3e-4 * max|ref|; the greedy CTC path (ids) must be identical wherever the top-2 log-prob gap exceeds the float
tolerance, log-probs 1e-3 absolute, colours 1e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ocr_ctc_parity(cuda):
    from manga_image_translator_amd import ocr_ctc, ocr_ctc_schema as S, synth
    from oracle import ocr_ctc as OC

    D = 97
    sd = synth.synth_state_dict(S.ocr_ctc_schema(D), gain=S.CTC_GAIN)
    eng = ocr_ctc.OcrCtcEngine(sd, D, device=cuda)
    rng = np.random.default_rng(4)
    crops = [rng.integers(0, 256, size=(48, w, 3), dtype=np.uint8) for w in (33, 61, 62, 140, 90)]
    for (indices, widths, region), (_, _, img) in zip(eng.make_chunks(crops, 3), OC.make_chunks(crops, 3)):
        taps = {}
        logits, colors = eng.forward(torch.from_numpy(region).to(cuda), taps=taps)
        got = eng.decode(logits, colors)
        torch.cuda.synchronize()
        with torch.no_grad():
            bb = OC.backbone(sd, img).squeeze(2).permute(0, 2, 1)
            rl, rc = OC.forward(sd, img)
        assert taps["backbone"].shape == bb.shape
        assert (taps["backbone"].cpu() - bb).abs().max().item() < 3e-4 * bb.abs().max().item()
        assert (logits.cpu() - rl).abs().max().item() < 3e-4 * rl.abs().max().item()
        assert (colors.cpu() - rc).abs().max().item() < 3e-4 * max(1.0, rc.abs().max().item())
        ref = OC.decode_ctc_top1(rl, rc)
        top2 = rl.log_softmax(2).topk(2, dim=2).values
        decisive = bool(((top2[..., 0] - top2[..., 1]) > 1e-3).all())
        for gl, rline in zip(got, ref):
            if decisive:
                assert [t[0] for t in gl] == [t[0] for t in rline]
                assert np.allclose([t[1] for t in gl], [t[1] for t in rline], atol=1e-3)
                assert np.allclose(np.array([t[2:] for t in gl]), np.array([t[2:] for t in rline]), atol=1e-3)
        assert sum(len(l) for l in ref) >= 10


def test_ocr_ctc_decode_collapse_rules(cuda):
    """Greedy CTC on crafted logits: repeats collapse, blanks (0) drop, a repeat after a blank is kept."""
    from manga_image_translator_amd import ocr_ctc, ocr_ctc_schema as S, synth
    from oracle import ocr_ctc as OC

    D = 12
    eng = ocr_ctc.OcrCtcEngine(synth.synth_state_dict(S.ocr_ctc_schema(D), gain=S.CTC_GAIN), D, device=cuda)
    path = [[0, 3, 3, 0, 3, 5, 5, 5, 0, 0, 7], [4, 4, 4, 4, 0, 4, 0, 0, 9, 9, 1]]
    logits = torch.full((2, 11, D), -2.0)
    for b, p in enumerate(path):
        for t, c in enumerate(p):
            logits[b, t, c] = 3.0 + 0.1 * t
    colors = torch.rand(2, 11, 6, generator=torch.Generator().manual_seed(0)) * 1.6 - 0.3
    got = eng.decode(logits.to(cuda), colors.to(cuda))
    ref = OC.decode_ctc_top1(logits, colors)
    assert [[t[0] for t in l] for l in got] == [[3, 3, 5, 7], [4, 4, 9, 1]] == [[t[0] for t in l] for l in ref]
    for gl, rl in zip(got, ref):
        assert np.allclose(np.array([t[1:] for t in gl]), np.array([t[1:] for t in rl]), atol=1e-5)


def test_ocr_ctc_plugin(cuda):
    import asyncio

    from manga_image_translator_amd import ocr_ctc_schema as S, plugins as P, synth, textline as TL

    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    D = 64
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x3041 + i) for i in range(D - 4)]
    ocr = P.HipModel48pxCTCOCR(weights=synth.synth_state_dict(S.ocr_ctc_schema(D), gain=S.CTC_GAIN), dictionary=dictionary)
    run(ocr.load("cuda"))
    page, quads, _ = synth.synth_page(3, 256, 320, n_boxes=5)
    lines = [TL.Quadrilateral(q) for q in quads]

    class Cfg:
        prob = 0.0
        ignore_bubble = 0

    out = run(ocr.infer(page, lines, Cfg()))
    assert 1 <= len(out) <= len(lines) and all(o in lines for o in out)
    assert all(isinstance(o.text, str) and len(o.text) >= 1 and 0.0 < o.prob <= 1.0 and 0 <= o.fg_r <= 255 for o in out)
    Cfg.prob = 1.1
    assert run(ocr.infer(page, lines, Cfg())) == []
    run(ocr.unload())
