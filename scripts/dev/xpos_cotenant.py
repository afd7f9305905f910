"""xpos_rotate_kernel beside LaMa forwards of another stream: where and by how much do its outputs differ?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from manga_image_translator_amd import lib as L, pipeline, lama, ocr48, synth
from manga_image_translator_amd.ocr48 import EMBD

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
w = pipeline.synthetic_weights()
leng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
oeng = ocr48.Ocr48Engine(w["ocr48"], pipeline.DICT_SIZE, device=dev)
page, quads, mask = synth.synth_page(3, bench.H, bench.W, n_boxes=bench.N_BOXES)
pages = torch.from_numpy(np.stack([page] * 4)).to(dev)
masks = torch.from_numpy(np.stack([mask] * 4)).to(dev)
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(3)
N, Lm = 16, 152
M = N * Lm
x = torch.randn(M, EMBD, generator=g).to(dev)
LE = Lm * EMBD
minpos = -((Lm + 1) // 2)
def rot():
    out = torch.empty(M, EMBD, device=dev); oeng._rotate(x, out, N, Lm, 0, minpos, True, LE, EMBD, LE, EMBD); return out
torch.cuda.synchronize()
with torch.cuda.stream(side):
    ref = rot().clone()
torch.cuda.synchronize()
for _ in range(3):
    leng.forward(pages, masks)
outs = []
with torch.cuda.stream(side):
    for _ in range(20):
        outs.append(rot())
torch.cuda.synchronize()
refc = ref.cpu().numpy(); xc = x.cpu().numpy()
for k, o in enumerate(outs[:6]):
    oc = o.cpu().numpy()
    idx = np.argwhere(oc != refc)
    print(f"call {k}: {len(idx)} elements differ; rows {sorted(set(idx[:, 0].tolist()))[:12]}...")
    for (r, c) in idx[:8]:
        print(f"   row {r} (line {r // Lm}, t {r % Lm}) col {c}: ref {refc[r, c]!r} got {oc[r, c]!r}  in ({xc[r, c & ~1]!r}, {xc[r, c | 1]!r})")
    rows = idx[:, 0]
    if len(rows):
        print("   distinct rows:", len(set(rows.tolist())), " cols per row (first rows):", [int((rows == r).sum()) for r in sorted(set(rows.tolist()))[:10]])
