"""Does a stage write outside its own tensors?  Sentinel tensors (filled with a pattern) are allocated around the stage's run; after the
stage has run ALONE (one stream, synchronised) every sentinel must be unchanged.   usage: python scripts/dev/stray_write_check.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from manga_image_translator_amd import lib as L, pipeline, lama, ocr48, synth

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
w = pipeline.synthetic_weights()
leng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
page, quads, mask = synth.synth_page(3, bench.H, bench.W, n_boxes=bench.N_BOXES)
pages = torch.from_numpy(np.stack([page] * 4)).to(dev)
masks = torch.from_numpy(np.stack([mask] * 4)).to(dev)

def sentinels(tag):
    out = []
    for i, n in enumerate((1 << 10, 3 << 12, 1 << 16, 768000, 3 * 768000, 1 << 22, 5 << 22, 1 << 26)):
        t = torch.full((n,), float(1000 + i), device=dev)
        out.append((f"{tag}{i} ({n} floats)", t))
    return out

S = sentinels("before-first-forward ")
leng.forward(pages, masks)
torch.cuda.synchronize()
S += sentinels("after-first-forward ")
for it in range(3):
    out = leng.forward(pages, masks)
    torch.cuda.synchronize()
    for name, t in S:
        v = float(t[0])
        bad = int((t != v).sum())
        if bad:
            idx = torch.nonzero(t != v).flatten()
            print(f"forward {it}: sentinel {name}: {bad} elements changed, first at {int(idx[0])}, last at {int(idx[-1])}, e.g. {t[idx[:4]].tolist()}", flush=True)
            t.fill_(v)
print("done: sentinels checked after 3 forwards", flush=True)
