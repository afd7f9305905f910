"""Is the two-stream mode of PageEngine (LaMa on the caller's stream, detector + OCR beside it) bit-identical to the one-stream mode?

usage: [MIT_COTENANT_SAFE=0|1] python scripts/dev/overlap_check.py [pages] [repeats]
Prints, per repeat, how many result elements of each stage differ from the one-stream run of the same engine."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from manga_image_translator_amd import pipeline, lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
w = pipeline.synthetic_weights()
pages, quads, masks, _, _ = bench.make_inputs(n, min(n, 16), 0, dev)
kw = dict(max_seq_length=bench.DECODE_STEPS, suppress_eos=True, stages=tuple(os.environ.get("STAGES", "detect,ocr,inpaint").split(",")))
one = pipeline.PageEngine(w, device=dev, overlap=False)
two = pipeline.PageEngine(w, device=dev, overlap=True)
if os.environ.get("PER_CHUNK"):
    one.ocr.per_chunk_attention = two.ocr.per_chunk_attention = True
def grab(r):
    torch.cuda.synchronize()
    return dict(det_mask=r.det_mask.cpu(), det_shrink=r.det_shrink.cpu(), tokens=r.ocr_tokens.cpu(), prob=r.ocr_prob.cpu(), colors=r.ocr_colors.cpu(),
                inpainted=r.inpainted.cpu(), mem_k=r.keep[-2].cpu(), mem_v=r.keep[-1].cpu())
ref = grab(one.run(pages, quads, masks, **kw))
again = grab(one.run(pages, quads, masks, **kw))
print("one-stream twice:", {k: int((ref[k] != again[k]).sum()) for k in ref})
for i in range(reps):
    got = grab(two.run(pages, quads, masks, **kw))
    d = {k: int((ref[k] != got[k]).sum()) for k in ref}
    extra = ""
    if d["prob"]:
        extra = " max |dprob| %.3g" % float((ref["prob"] - got["prob"]).abs().max())
    print(f"two-stream run {i} (MIT_COTENANT_SAFE={os.environ.get('MIT_COTENANT_SAFE', '0')}):", d, extra)
