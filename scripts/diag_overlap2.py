"""Second diagnosis of the two-stream mismatch: (1) each stage alone on a NON-default stream (a kernel that ignored its stream
argument would lose its ordering), (2) each stage alone with an unrelated load running on another stream, (3) the same in GEMM mode 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import ops, pipeline, synth
cuda = torch.device("cuda:0")
D = 97
weights = pipeline.synthetic_weights(dict_size=D)
B, H, W, T = 4, 256, 320, 5
pages, quads, masks = zip(*[synth.synth_page(10 + i, H, W, n_boxes=5) for i in range(B)])
qobjs = [pipeline.quads_from_array(q) for q in quads]
pd, md = torch.from_numpy(np.stack(pages)).to(cuda), torch.from_numpy(np.stack(masks)).to(cuda)
names = ("det_mask", "det_shrink", "inpainted", "ocr_tokens", "ocr_length", "ocr_prob", "ocr_colors")
side, other = torch.cuda.Stream(), torch.cuda.Stream()
xa = torch.randn(4096, 4096, device=cuda)
def run(stages, stream=None, load=False, reps=3):
    eng = pipeline.PageEngine(weights, device=cuda, dict_size=D, ctd_mb=2, lama_mb=2, group=2, overlap=False)
    r = None
    for _ in range(reps):
        if load:
            with torch.cuda.stream(other):
                for _ in range(40):
                    torch.mm(xa, xa)
        if stream is None:
            r = eng.run(pd, qobjs, md, max_seq_length=T, suppress_eos=True, stages=stages)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                r = eng.run(pd, qobjs, md, max_seq_length=T, suppress_eos=True, stages=stages)
        torch.cuda.synchronize()
    return r
def diff(a, b, tag):
    out = []
    for n in names:
        x, y = getattr(a, n), getattr(b, n)
        if x is None or y is None:
            continue
        if x.shape != y.shape or not torch.equal(x, y):
            d = (x.float() - y.float()).abs()
            out.append(f"{n}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3g}")
    print(tag, "->", out or "identical", flush=True)
for mode in (6, 0):
    with ops.gemm_mode(mode):
        for st in (("inpaint",), ("ocr",), ("detect",)):
            base = run(st)
            diff(base, run(st, stream=side), f"mode {mode} {st}: default stream vs side stream, nothing else running")
            diff(base, run(st, load=True), f"mode {mode} {st}: default stream, unrelated GEMMs on another stream")
            diff(base, run(st, stream=side, load=True), f"mode {mode} {st}: side stream + unrelated load")
