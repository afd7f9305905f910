"""Text-line rectification (mit_ocr_warp_lines) and the three-stage page pipeline vs the CPU oracle.

Integer/byte work must be bit-exact: the rectified uint8 crops, the chunk packing, the detector's uint8 mask and
thresholded bitmap (away from the stated fp32 margin), OCR token ids.  Floating point: OCR probabilities 1e-3 relative,
LaMa uint8 output +-1 level only where the oracle's pre-truncation value is within 0.05 of an integer.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand_quads(rng, H, W, n):
    """Mix of axis-aligned, slightly rotated and perspective-skewed boxes, some touching the page border."""
    quads = []
    for k in range(n):
        vertical = k % 2 == 0
        bw, bh = (int(rng.integers(20, 40)), int(rng.integers(90, 200))) if vertical else (int(rng.integers(90, 200)), int(rng.integers(18, 40)))
        x0, y0 = int(rng.integers(0, W - bw)), int(rng.integers(0, H - bh))
        q = np.array([[x0, y0], [x0 + bw, y0], [x0 + bw, y0 + bh], [x0, y0 + bh]], dtype=np.int64)
        if k % 3 == 1:  # rotate by a few degrees about the centre
            a = np.deg2rad(rng.uniform(-8, 8))
            c = q.mean(0)
            R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            q = np.rint((q - c) @ R.T + c).astype(np.int64)
        elif k % 3 == 2:  # perspective-ish jitter
            q = q + rng.integers(-4, 5, size=(4, 2))
        if k == 0:
            q[:, 0] -= q[:, 0].min()  # flush with the left border
        quads.append(q)
    return quads


def test_warp_lines_bit_exact(cuda):
    from manga_image_translator_amd import lib as L, ocr48, textline as TL
    from oracle import textline as OT

    rng = np.random.default_rng(5)
    P, H, W = 2, 300, 420
    pages = rng.integers(0, 256, size=(P, H, W, 3), dtype=np.uint8)
    lib = L.load()
    pages_dev = torch.from_numpy(pages).to(cuda)
    for p in range(P):
        quads = [TL.Quadrilateral(q) for q in _rand_quads(rng, H, W, 9)]
        plans = [TL.warp_plan(q, q.direction, H, W) for q in quads]
        for idx, ws, wp in TL.chunk_plan([pl.width for pl in plans], max_chunk_size=4):
            recs = (L.MitWarpLine * len(idx))()
            for row, i in enumerate(idx):
                pl, r = plans[i], recs[row]
                r.minv[:] = pl.minv.reshape(-1).tolist()
                r.page, r.x1, r.y1, r.cw, r.ch, r.dw, r.dh, r.vertical, r.out_row = p, pl.x1, pl.y1, pl.cw, pl.ch, pl.dw, pl.dh, int(pl.vertical), row
            rec_dev = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(cuda)
            out = torch.full((len(idx), 48, wp, 3), 77, dtype=torch.uint8, device=cuda)
            L.check(lib.mit_ocr_warp_lines(pages_dev.data_ptr(), H, W, rec_dev.data_ptr(), len(idx), out.data_ptr(), 48, wp, None))
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            for row, i in enumerate(idx):
                q = quads[i]
                ref = OT.get_transformed_region(pages[p], q.pts, q.direction, 48)
                assert ref.shape == (48, ws[row], 3)
                assert np.array_equal(got[row, :, :ws[row]], ref), f"page {p} line {i}: crop differs"
                assert not got[row, :, ws[row]:].any(), "chunk padding must be zero"


def test_recognize_pages_matches_host_chunking(cuda):
    """recognize_pages (GPU warp + device chunk packing) == recognize() on oracle-made crops of the same lines."""
    from manga_image_translator_amd import ocr48, ocr_schema, synth, textline as TL
    from oracle import textline as OT

    D = 97
    eng = ocr48.Ocr48Engine(synth.synth_state_dict(ocr_schema.ocr48_schema(D)), D, device=cuda)
    rng = np.random.default_rng(11)
    H, W = 256, 384
    page = rng.integers(0, 256, size=(1, H, W, 3), dtype=np.uint8)
    quads = [TL.Quadrilateral(q) for q in _rand_quads(rng, H, W, 6)]
    got = eng.recognize_pages(torch.from_numpy(page).to(cuda), [quads], max_seq_length=6, suppress_eos=True)
    crops = [OT.get_transformed_region(page[0], q.pts, q.direction, 48) for q in quads]
    ref = eng.recognize(crops, max_seq_length=6, suppress_eos=True)
    torch.cuda.synchronize()
    assert [i for _, i in got["order"]] == ref["order"]
    assert torch.equal(got["tokens"].cpu(), ref["tokens"].cpu())
    assert torch.allclose(got["prob"].cpu(), ref["prob"].cpu(), rtol=1e-5, atol=0)


def test_page_pipeline_parity(cuda, gemm_mode):
    from manga_image_translator_amd import pipeline, synth
    from oracle import ctd as OC, lama as OL, ocr48 as OO, textline as OT

    D = 211
    weights = pipeline.synthetic_weights(dict_size=D)
    from manga_image_translator_amd import lib as L, ops

    eng = pipeline.PageEngine(weights, device=cuda, dict_size=D, ctd_mb=2, lama_mb=2, group=2)
    B, H, W, T = 3, 256, 192, 5
    pages, quads, masks = zip(*[synth.synth_page(i, H, W, n_boxes=4) for i in range(B)])
    qobjs = [pipeline.quads_from_array(q) for q in quads]
    # these pages are far below the launch size the split tiles normally take: lower the threshold so that in the split mode every
    # eligible layer runs on them, and check with the launch probe which tiles ran
    lib = L.load()
    with ops.gemm_mode(gemm_mode, min_tiles=1):
        L.check(lib.mit_prof_enable(1), "mit_prof_enable")
        res = eng.run(torch.from_numpy(np.stack(pages)).to(cuda), qobjs, torch.from_numpy(np.stack(masks)).to(cuda),
                      max_seq_length=T, suppress_eos=True)
        torch.cuda.synchronize()
        stats = (L.MitProfStat * 64)()
        ncfg = C.c_int(0)
        L.check(lib.mit_prof_read(stats, 64, C.byref(ncfg)), "mit_prof_read")
        L.check(lib.mit_prof_enable(0), "mit_prof_enable")
    split_launches = sum(stats[i].launches for i in range(ncfg.value) if lib.mit_conv_gemm_config_name(i).decode().startswith("split"))
    assert (split_launches > 100) if gemm_mode else (split_launches == 0), (gemm_mode, split_launches)
    toks, probs = res.ocr_tokens.cpu().numpy(), res.ocr_prob.cpu().numpy()
    row = 0
    for b in range(B):
        # detect: uint8 mask and thresholded bitmap
        rmask, rlines = OC.infer_maps(weights["ctd.yolo"], weights["ctd.seg"], weights["ctd.det"], pages[b])
        gm = res.det_mask[b].cpu().numpy()
        assert gm.shape == rmask.shape
        dm = gm.astype(np.int32) - rmask.astype(np.int32)
        assert np.abs(dm).max() <= 1 and (dm != 0).mean() < 1e-2  # truncation of v*255 at an integer boundary
        near = np.abs(rlines[0, 0] - 0.3) < 1e-4
        gs = res.det_shrink[b].cpu().numpy().astype(bool)
        assert np.array_equal(gs[~near], (rlines[0, 0] > 0.3)[~near]), "thresholded bitmap differs outside the 1e-4 margin"
        # ocr: same crops, same chunks -> same tokens
        crops = [OT.get_transformed_region(pages[b], q.pts, q.direction, 48) for q in qobjs[b]]
        for indices, widths, img in OO.make_chunks(crops):
            out = OO.infer_beam_batch_tensor(weights["ocr48"], img, widths, max_seq_length=T, suppress_eos=True)
            for j, i in enumerate(indices):
                assert res.ocr_order[row] == (b, i)
                ref_tok = out[j][0].numpy()
                n = int(res.ocr_length[row])
                assert np.array_equal(toks[row, 1:n], ref_tok), (b, i, toks[row, :n], ref_tok)
                assert abs(probs[row] - out[j][1]) <= 1e-3 * max(out[j][1], 1e-6) + 1e-7
                row += 1
        # inpaint
        otaps = {}
        ref = OL.infer(weights["lama.gen"], weights["lama.mpe"], pages[b], masks[b], 9, otaps)
        diff = res.inpainted[b].cpu().numpy().astype(np.int32) - ref.astype(np.int32)
        bad = np.argwhere(diff != 0)
        if len(bad):
            of = otaps["out_float"][0].permute(1, 2, 0).numpy() * 255.0
            frac = np.abs(of - np.round(of))
            assert np.abs(diff).max() <= 1 and all(frac[tuple(x)] < 0.05 for x in bad) and len(bad) < 1e-3 * diff.size
    assert row == len(res.ocr_order)
    # the packed gather payload covers every result tensor
    assert res.packed().numel() == (res.det_mask.numel() + res.det_shrink.numel() + res.inpainted.numel()
                                    + 4 * (res.ocr_tokens.numel() + res.ocr_length.numel() + res.ocr_prob.numel() + res.ocr_colors.numel()))


def test_plugins_end_to_end(cuda):
    """The three drop-in plugins (load -> infer -> unload) return what the stage engines compute, in the reference's types."""
    import asyncio

    from manga_image_translator_amd import pipeline, plugins as P, synth, textline as TL
    from oracle import lama as OL

    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    D = 64
    weights = pipeline.synthetic_weights(dict_size=D)
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(D - 4)]
    H, W = 256, 192
    page, quads, mask = synth.synth_page(2, H, W, n_boxes=4)

    det = P.HipComicTextDetector(weights=weights,
                                 boxes_from_maps=lambda lm, h, w: (np.array([[[1, 1], [40, 1], [40, 9], [1, 9]]]), np.array([0.9])),
                                 refine=lambda img, m, tl, h, w: m)
    run(det.load("cuda"))
    tls, raw_mask, extra = run(det.infer(page, 1024, 0.5, 0.7, 2.3))
    assert extra is None and len(tls) == 1 and tls[0].prob == pytest.approx(0.9)
    assert raw_mask.dtype == np.uint8 and raw_mask.shape == (1024, 768)  # un-padded letterbox area (the injected refine hook skips the resize back)
    run(det.unload())
    assert not det.is_loaded()

    ocr = P.HipModel48pxOCR(weights=weights["ocr48"], dictionary=dictionary)
    run(ocr.load("cuda"))
    lines = [TL.Quadrilateral(q) for q in quads]

    class Cfg:
        prob = 0.0

    out = run(ocr.infer(page, lines, Cfg(), max_seq_length=6, suppress_eos=True))
    assert len(out) == len(lines) and all(o in lines for o in out)  # same objects, mutated
    assert all(isinstance(o.text, str) and len(o.text) <= 6 and 0 <= o.fg_r <= 255 and 0 <= o.bg_b <= 255 for o in out)
    widths = [TL.warp_plan(q, q.direction, H, W).width for q in lines]
    assert [lines.index(o) for o in out] == sorted(range(len(lines)), key=lambda i: widths[i])  # chunk order (:79,176)
    Cfg.prob = 2.0
    assert run(ocr.infer(page, lines, Cfg(), max_seq_length=6, suppress_eos=True)) == []  # everything below the threshold
    run(ocr.unload())

    inp = P.HipLamaMPEInpainter(weights=weights)
    run(inp.load("cuda"))
    got = run(inp.infer(page, mask, None, 2048))
    ref = OL.infer(weights["lama.gen"], weights["lama.mpe"], page, mask, 9)
    assert got.shape == page.shape and got.dtype == np.uint8
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    assert np.array_equal(got[mask < 127], page[mask < 127])  # outside the mask the page is returned untouched
    # 250 x 187 is not a multiple of 8: the plugin resizes to 256 x 192 and back on the GPU (inpainting_lama_mpe.py:69-79,112-117)
    p2, m2 = np.ascontiguousarray(page[:250, :187]), np.ascontiguousarray(mask[:250, :187])
    got2 = run(inp.infer(p2, m2, None, 2048))
    ref2 = OL.infer(weights["lama.gen"], weights["lama.mpe"], p2, m2, 9, inpainting_size=2048)
    d2 = np.abs(got2.astype(np.int32) - ref2.astype(np.int32))
    assert got2.shape == p2.shape and d2.max() <= 1 and (d2 != 0).mean() < 1e-3
    assert np.array_equal(got2[m2 < 127], p2[m2 < 127])
    run(inp.unload())

