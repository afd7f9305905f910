"""Why do page records depend on the number of pages in the batch (test_dist_gpu)?  Same pages through PageEngine as one batch of 4 and
as two batches of 2; which fields of the OCR result differ, and whether one batch is reproducible run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import pipeline, synth
H, W, LINES, T, D = 256, 192, 3, 4, 96
dev = torch.device("cuda:0")
weights = pipeline.synthetic_weights(dict_size=D)
eng = pipeline.PageEngine(weights, device=dev, dict_size=D)
def run(lo, hi):
    pages, quads, masks = zip(*[synth.synth_page(i, H, W, n_boxes=LINES) for i in range(lo, hi)])
    res = eng.run(torch.from_numpy(np.stack(pages)).to(dev), [pipeline.quads_from_array(q) for q in quads],
                  torch.from_numpy(np.stack(masks)).to(dev), max_seq_length=T, suppress_eos=True)
    torch.cuda.synchronize()
    o = {"order": list(res.ocr_order), "tokens": res.ocr_tokens.clone(), "prob": res.ocr_prob.clone(), "colors": res.ocr_colors.clone(),
         "length": res.ocr_length.clone()}
    return o, res.packed_pages(LINES).cpu().numpy()
a, pa = run(0, 4)
a2, pa2 = run(0, 4)
print("keys", list(a.keys()))
print("same batch twice: records equal", np.array_equal(pa, pa2))
b, pb = run(0, 2)
c, pc = run(2, 4)
full = np.concatenate([pb, pc])
print("4 pages vs 2 + 2: records equal", np.array_equal(pa, full), "bytes differing", int((pa != full).sum()))
for name, o in (("batch of 4", a), ("pages 0-1", b), ("pages 2-3", c)):
    print(name, "order", o["order"])
# per-line comparison through the order lists
ia = {pi: r for r, pi in enumerate(a["order"])}
for o, off in ((b, 0), (c, 2)):
    for r, (p, i) in enumerate(o["order"]):
        ra = ia[(p + off, i)]
        dp = float((a["prob"][ra] - o["prob"][r]).abs())
        tok = bool((a["tokens"][ra] != o["tokens"][r]).any())
        dc = float((a["colors"][ra].float() - o["colors"][r].float()).abs().max())
        print(f"page {p + off} line {i}: prob {float(a['prob'][ra]):.7f} vs {float(o['prob'][r]):.7f} (diff {dp:.2e}) tokens differ {tok} max colour diff {dc:.2e}")
print("a vs a2: prob equal", torch.equal(a["prob"], a2["prob"]), "colours equal", torch.equal(a["colors"], a2["colors"]))
