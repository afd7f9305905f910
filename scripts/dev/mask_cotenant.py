"""Which kernel of the mask-refinement stage changes its output when kernels of another stream (a LaMa forward) share the GPU?

Each candidate runs REPS times on a side stream with fixed inputs while LaMa forwards run on the main stream; every output is compared
with the output of the same call on an idle GPU.   usage: python scripts/dev/mask_cotenant.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from manga_image_translator_amd import lib as L, pipeline, lama, imgproc, mask_refinement as MR, synth

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
REPS = int(os.environ.get("REPS", "40"))
w = pipeline.synthetic_weights()
eng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
page, quads, mask = synth.synth_page(3, bench.H, bench.W, n_boxes=bench.N_BOXES)
pages = torch.from_numpy(np.stack([page] * 4)).to(dev)
masks = torch.from_numpy(np.stack([mask] * 4)).to(dev)
side = torch.cuda.Stream(priority=-1)
be = MR.GpuMaskBackend(dev)
page_d, mask_d = pages[0], masks[0]
H, W = bench.H, bench.W
scale = max(min((H - H / 3) / H, 1), 0.5)
size = (int(W * scale), int(H * scale))

def disturb(n=3):
    for _ in range(n):
        eng.forward(pages, masks)

def check(name, fn, eq=torch.equal):
    torch.cuda.synchronize()
    ref = fn()
    torch.cuda.synchronize()
    ref = [r.clone() for r in (ref if isinstance(ref, (list, tuple)) else [ref])]
    for label, dist in (("idle", False), ("beside LaMa", True)):
        bad = 0
        torch.cuda.synchronize()
        if dist:
            disturb()
        outs = []
        with torch.cuda.stream(side):
            for _ in range(REPS):
                o = fn()
                outs.append([x.clone() for x in (o if isinstance(o, (list, tuple)) else [o])])
        torch.cuda.synchronize()
        for o in outs:
            bad += any(not torch.equal(a, b) for a, b in zip(o, ref))
        print(f"{name:38s} {label:12s} {bad:3d} of {REPS} calls differ", flush=True)

small = be.resize_image(page_d, size)
filt = be.filter_page(small)
check("resize_u8 (page -> 2/3)", lambda: be.resize_image(page_d, size))
check("resize + binarize (mask)", lambda: be.resize_binarize(mask_d, size))
check("bilateral 17 / 80 / 80", lambda: be.filter_page(small))
# CRF on fixed crops: the generator's boxes on the scaled page
rects, cms = [], []
msmall = be.resize_binarize(mask_d, size).cpu().numpy()
for q in quads[:24]:
    x0, y0 = (q.min(0) * scale).astype(int)
    x1, y1 = (q.max(0) * scale).astype(int) + 1
    x0, y0, x1, y1 = max(x0 - 4, 0), max(y0 - 4, 0), min(x1 + 4, size[0]), min(y1 + 4, size[1])
    rects.append((int(x0), int(y0), int(x1 - x0), int(y1 - y0)))
    cms.append(np.ascontiguousarray(msmall[y0:y1, x0:x1]))
check("densecrf_refine (24 crops)", lambda: be._crf.refine(filt, rects, cms, packed=True)[0])
jobs = [(r, (max(r[0] - 12, 0), max(r[1] - 12, 0), min(r[2] + 24, size[0] - max(r[0] - 12, 0)), min(r[3] + 24, size[1] - max(r[1] - 12, 0))), 17) for r in rects]
check("crf + dilate jobs + union + closing", lambda: be.refine_dilate_union(filt, jobs, cms, size[1], size[0], 3))
fin = be.refine_dilate_union(filt, jobs, cms, size[1], size[0], 3)
check("resize + binarize (final mask up)", lambda: be.resize_binarize(fin, (W, H)))
