"""Diagnose the two-stream mode: which result tensors differ from the one-stream run, and for which stage combinations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import pipeline, synth
cuda = torch.device("cuda:0")
D = 97
weights = pipeline.synthetic_weights(dict_size=D)
B, H, W, T = 4, 256, 320, 5
pages, quads, masks = zip(*[synth.synth_page(10 + i, H, W, n_boxes=5) for i in range(B)])
qobjs = [pipeline.quads_from_array(q) for q in quads]
pd, md = torch.from_numpy(np.stack(pages)).to(cuda), torch.from_numpy(np.stack(masks)).to(cuda)
names = ("det_mask", "det_shrink", "inpainted", "ocr_tokens", "ocr_length", "ocr_prob", "ocr_colors")
def run(overlap, stages, reps=2):
    eng = pipeline.PageEngine(weights, device=cuda, dict_size=D, ctd_mb=2, lama_mb=2, group=2, overlap=overlap)
    for _ in range(reps):
        r = eng.run(pd, qobjs, md, max_seq_length=T, suppress_eos=True, stages=stages)
    torch.cuda.synchronize()
    return r
def diff(a, b, tag):
    out = []
    for n in names:
        x, y = getattr(a, n), getattr(b, n)
        if x is None or y is None:
            continue
        if not torch.equal(x, y):
            d = (x.float() - y.float()).abs()
            out.append(f"{n}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3g}")
    print(tag, "->", out or "identical")
full = ("detect", "ocr", "inpaint")
a = run(False, full)
diff(a, run(False, full), "one-stream vs one-stream (second engine instance)")
for st in (full, ("detect", "inpaint"), ("ocr", "inpaint")):
    for rep in range(2):
        diff(run(False, st), run(True, st), f"one-stream vs two-stream {st} try {rep}")
b = run(True, full, reps=1)
diff(a, b, "two-stream, first call of a fresh engine")
