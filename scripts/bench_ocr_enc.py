"""OCR stage timing on the GPU box (not the contract bench): one page group through Ocr48Engine, ms per page."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import pipeline, synth
dev = torch.device("cuda:0")
P = int(os.environ.get("PAGES", "16"))
weights = pipeline.synthetic_weights(dict_size=6004)
eng = pipeline.PageEngine(weights, device=dev, dict_size=6004)
pages, quads, masks = zip(*[synth.synth_page(i, 2048, 1456, n_boxes=32) for i in range(P)])
pg = torch.from_numpy(np.stack(pages)).to(dev); mk = torch.from_numpy(np.stack(masks)).to(dev)
qs = [pipeline.quads_from_array(q) for q in quads]
for _ in range(2):
    eng.run(pg, qs, mk, max_seq_length=32, suppress_eos=True, stages=("ocr",))
torch.cuda.synchronize()
t = time.time(); n = 3
for _ in range(n):
    eng.run(pg, qs, mk, max_seq_length=32, suppress_eos=True, stages=("ocr",))
torch.cuda.synchronize()
print(f"ocr stage: {(time.time() - t) / n / P * 1e3:.3f} ms/page ({P} pages)")
