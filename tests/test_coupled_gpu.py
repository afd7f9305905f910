"""The coupled page path (manga_image_translator_amd/coupled.py): detector -> boxes -> refine_mask -> OCR of the detected lines ->
text-line merge -> mask refinement -> inpainting with the refined mask, batched, against the same chain through the drop-in
plugins one page at a time (the order of manga_translator.py:432-622).  A trained detector head is stood in for by
coupled.synthetic_head_outputs (random-init weights fire on nothing); both sides get the same maps."""
import asyncio
import warnings

import numpy as np
import pytest
import torch



def test_injected_head_maps_give_back_the_generator_boxes():
    """CPU-side part (native host code only, no GPU; kept here with its user): the DB shrink map of the stand-in head, through the native
    SegDetectorRepresenter, returns one box per generator box with IoU > 0.75."""
    from manga_image_translator_amd import coupled, hostglue as HG, synth

    H, W = 2048, 1456
    page, quads, _ = synth.synth_page(1, H, W, n_boxes=32, disjoint=True)
    prob, mask = coupled.synthetic_head_outputs(page, quads, (1024, 728))
    lines = np.zeros((1, 2, 1024, 728), np.float32)
    lines[0, 0] = prob
    boxes, scores = HG.ctd_boxes(lines, H, W)
    boxes = boxes[scores > 0.6]
    assert len(boxes) == 32 and mask.any()
    for q in quads:
        a = (q[:, 0].min(), q[:, 1].min(), q[:, 0].max(), q[:, 1].max())
        best = 0.0
        for b in boxes:
            c = (b[:, 0].min(), b[:, 1].min(), b[:, 0].max(), b[:, 1].max())
            ix, iy = max(0, min(a[2], c[2]) - max(a[0], c[0])), max(0, min(a[3], c[3]) - max(a[1], c[1]))
            inter = ix * iy
            best = max(best, inter / ((a[2] - a[0]) * (a[3] - a[1]) + (c[2] - c[0]) * (c[3] - c[1]) - inter))
        assert best > 0.75, best


@pytest.mark.gpu
def test_coupled_batch_equals_the_plugin_chain_page_by_page(cuda):
    from manga_image_translator_amd import coupled, ctd as CTD, mask_refinement as MR, pipeline, plugins as P, synth, textline_merge as TM

    H, W, NB, T, D = 1024, 728, 8, 6, 211
    weights = pipeline.synthetic_weights(dict_size=D)
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(D - 4)]
    gen = [synth.synth_page(20 + i, H, W, n_boxes=NB, disjoint=True) for i in range(3)]
    pages = [g[0] for g in gen]
    nh, nw, dw, dh = CTD.CtdEngine.letterbox_geometry(H, W)
    heads = [coupled.synthetic_head_outputs(g[0], g[1], (CTD.INPUT_SIZE - dh, CTD.INPUT_SIZE - dw)) for g in gen]
    inj = {"prob": torch.from_numpy(np.stack([h[0] for h in heads])).to(cuda), "mask": torch.from_numpy(np.stack([h[1] for h in heads])).to(cuda)}
    pages_dev = torch.from_numpy(np.stack(pages)).to(cuda)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        eng = coupled.CoupledPageEngine(weights, dictionary, device=cuda, ctd_mb=2, lama_mb=2, host_workers=4)
        res = eng.run(pages_dev, max_seq_length=T, suppress_eos=True, prob_threshold=0.0, inject=inj)
        torch.cuda.synchronize()
        # the same pages as three pipeline slots of one page (stage threads working on different pages at once): identical results
        piped = eng.run(pages_dev, max_seq_length=T, suppress_eos=True, prob_threshold=0.0, inject=inj, group=1)
        torch.cuda.synchronize()
        eng.close()
        assert torch.equal(piped.mask, res.mask) and torch.equal(piped.inpainted, res.inpainted)
        assert [[(l.text, l.prob) for l in t] for t in piped.textlines] == [[(l.text, l.prob) for l in t] for t in res.textlines]
        assert [[r.text for r in rg] for rg in piped.regions] == [[r.text for r in rg] for rg in res.regions]
        assert [len(t) for t in res.textlines] == [NB] * 3          # every generator box detected and recognised
        assert all(len(r) >= 1 for r in res.regions) and res.mask.any()

        run = asyncio.new_event_loop().run_until_complete
        det = P.HipComicTextDetector(weights=weights)
        ocr = P.HipModel48pxOCR(weights=weights["ocr48"], dictionary=dictionary)
        inp = P.HipLamaMPEInpainter(weights=weights)
        for p in (det, ocr, inp):
            run(p.load("cuda"))
        plain = det.engine.forward
        cur = {"k": 0}

        def fwd(pages_u8, taps=None):
            m, lines, pad = plain(pages_u8, taps)
            k = cur["k"]
            lines[:, 0] = inj["prob"][k:k + 1]
            return inj["mask"][k:k + 1], lines, pad

        det.engine.forward = fwd

        class Cfg:
            prob = 0.0

        for k, page in enumerate(pages):
            cur["k"] = k
            tls, mask_raw, _ = run(det.infer(page, 1024, 0.5, 0.7, 2.3))
            lines = [l for l in run(ocr.infer(page, tls, Cfg(), False, 0, T, True)) if l.text.strip()]
            regions = TM.dispatch_sync(lines, W, H)
            mask = MR.dispatch_sync(regions, page, mask_raw, "fit_text", 20, 0, False, 3)
            out = run(inp.infer(page, mask, None, max(H, W)))
            # the batch engine's page k: same lines (geometry, text, colours) in the same order, same regions, same bytes
            got = res.textlines[k]
            assert len(got) == len(lines)
            for a, b in zip(got, lines):
                assert np.array_equal(np.asarray(a.pts), np.asarray(b.pts)) and a.text == b.text and a.prob == pytest.approx(b.prob, rel=1e-6)
                assert (a.fg_r, a.fg_g, a.fg_b, a.bg_r, a.bg_g, a.bg_b) == (b.fg_r, b.fg_g, b.fg_b, b.bg_r, b.bg_g, b.bg_b)
            assert [r.text for r in res.regions[k]] == [r.text for r in regions]
            assert np.array_equal(res.mask[k].cpu().numpy(), mask), int((res.mask[k].cpu().numpy() != mask).sum())
            assert np.array_equal(res.inpainted[k].cpu().numpy(), np.asarray(out).astype(np.uint8))
        for p in (det, ocr, inp):
            run(p.unload())
