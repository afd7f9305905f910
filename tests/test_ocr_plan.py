"""Host planning of the OCR stage (no GPU): Ocr48Engine.plan_pages reproduces Model48pxOCR._infer's per-page chunking
(model_48px.py:79-86) for every page of a batch and the key-length rule of infer_beam_batch_tensor (:684-688)."""
import numpy as np

from manga_image_translator_amd import ocr48, synth, textline as TL


def test_plan_pages_matches_reference_batching():
    eng = ocr48.Ocr48Engine.__new__(ocr48.Ocr48Engine)  # planning needs no weights and no device
    H, W = 512, 384
    quads_per_page = [[TL.Quadrilateral(q) for q in synth.synth_page(i, H, W, n_boxes=n)[1]] for i, n in ((0, 5), (1, 0), (2, 20))]
    plan = eng.plan_pages(quads_per_page, H, W)
    assert len(plan["records"]) == 25 == len(plan["order"]) == len(plan["klens"])
    row = 0
    for p, quads in enumerate(quads_per_page):
        widths = [TL.warp_plan(q, q.direction, H, W).width for q in quads]
        for idx, ws, wp in TL.chunk_plan(widths):
            chunk = next(c for c in plan["chunks"] if c[0] == row)
            assert chunk[1:] == (len(idx), ws, wp, p)
            L = (wp // 2) // 2
            for j, i in enumerate(idx):
                r = plan["records"][row]
                assert plan["order"][row] == (p, i) and r["page"] == p and r["out_row"] == j
                assert (r["dh"] if r["vertical"] else r["dw"]) == ws[j]
                assert plan["klens"][row] == min((ws[j] + 3) // 4 + 2, L)
                row += 1
    assert row == 25
    assert plan["Lmax"] == max((c[3] // 2) // 2 for c in plan["chunks"])
    empty = eng.plan_pages([[], []], H, W)
    assert len(empty["records"]) == 0 and empty["Lmax"] == 0 and empty["chunks"] == []
