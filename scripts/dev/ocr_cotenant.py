"""Which part of the OCR encode path changes its output when kernels of another stream (LaMa forwards) share the GPU?

encode() of one 16-line chunk runs REPS times on a side stream while LaMa forwards run on the main stream; backbone output, encoder
memory and the per-layer cross-attention K / V are compared with the quiet run.   usage: python scripts/dev/ocr_cotenant.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from manga_image_translator_amd import lib as L, pipeline, lama, ocr48, ops, synth

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
REPS = int(os.environ.get("REPS", "30"))
w = pipeline.synthetic_weights()
leng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
oeng = ocr48.Ocr48Engine(w["ocr48"], pipeline.DICT_SIZE, device=dev)
page, quads, mask = synth.synth_page(3, bench.H, bench.W, n_boxes=bench.N_BOXES)
pages = torch.from_numpy(np.stack([page] * 4)).to(dev)
masks = torch.from_numpy(np.stack([mask] * 4)).to(dev)
rng = np.random.default_rng(0)
widths = sorted(int(x) for x in rng.integers(180, 600, size=16))
Wp = max(widths) + 7
region = torch.from_numpy(rng.integers(0, 256, size=(16, 48, Wp, 3), dtype=np.uint8)).to(dev)
side = torch.cuda.Stream()

def run():
    taps = {}
    mk, mv, kl, Lm = oeng.encode(region, widths, taps)
    return dict(backbone=taps["backbone"].clone(), memory=taps["memory"].clone(), mem_k=mk.clone(), mem_v=mv.clone())

with torch.cuda.stream(side):
    ref = run()
torch.cuda.synchronize()
for label, n_lama in (("idle", 0), ("beside LaMa", 4), ("beside LaMa", 4)):
    torch.cuda.synchronize()
    for _ in range(n_lama):
        leng.forward(pages, masks)
    outs = []
    with torch.cuda.stream(side):
        for _ in range(REPS):
            outs.append(run())
    torch.cuda.synchronize()
    bad = {k: sum(int(not torch.equal(o[k], ref[k])) for o in outs) for k in ref}
    first = {k: next((int((o[k] != ref[k]).sum()) for o in outs if not torch.equal(o[k], ref[k])), 0) for k in ref}
    print(f"{label:12s} calls of {REPS} that differ: {bad}   elements in the first differing call: {first}", flush=True)
