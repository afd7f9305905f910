"""Timing of the GPU halves of the host stages on one realistic page (not the contract bench): the ctd detector's refine_mask and
the mask refinement between OCR and inpainting (bilateral filter + batched DenseCRF), each against its host counterpart.
Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image, ImageDraw
from manga_image_translator_amd import hostglue as HG, imgproc, mask_refinement as MR, synth, textline as TL

dev = torch.device("cuda:0")
H, W = 2048, 1456
page, quads, mask = synth.synth_page(0, H, W, n_boxes=32)
rng = np.random.default_rng(0)
# glyph-like strokes inside the text boxes and a soft network-like mask over them
im = Image.new("L", (W, H), 0)
d = ImageDraw.Draw(im)
for q in quads:
    d.polygon([tuple(p) for p in q], fill=230)
pred = np.asarray(im).copy()
for q in quads:
    x0, y0, x1, y1 = int(q[:, 0].min()), int(q[:, 1].min()), int(q[:, 0].max()), int(q[:, 1].max())
    for _ in range(max((x1 - x0) * (y1 - y0) // 600, 4)):
        sx, sy = rng.integers(x0, max(x1 - 8, x0 + 1)), rng.integers(y0, max(y1 - 12, y0 + 1))
        page[sy:sy + rng.integers(3, 12), sx:sx + rng.integers(2, 8)] = rng.integers(0, 70)
lines = [TL.Quadrilateral(q.astype(np.float64)) for q in quads]


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3, out


pd, md = torch.from_numpy(page).to(dev), torch.from_numpy(pred).to(dev)
t_gpu, got = timed(lambda: HG.refine_mask_gpu(pd, md, lines, None).cpu().numpy())
t0 = time.time(); ref = HG.refine_mask(page, pred, lines, None); t_host = (time.time() - t0) * 1e3
res = {"page": [H, W], "lines": len(lines),
       "ctd_refine_mask": {"gpu_ms": round(t_gpu, 2), "host_ms": round(t_host, 1), "identical": bool(np.array_equal(got, ref)),
                           "note": "page + mask resident on the device, result copied back (what the plugin does)"}}
region = type("Region", (), {"lines": np.stack([l.pts for l in lines]).astype(np.int64)})()
t_mr, out = timed(lambda: MR.dispatch_sync([region], page, ref.copy()), n=2)
be = MR.default_backend()
t_bil, _ = timed(lambda: be.filter_page(page).cpu())
res["mask_refinement_dispatch"] = {"gpu_backend_ms": round(t_mr, 1), "bilateral_filter_ms_incl_copies": round(t_bil, 2),
                                   "mask_pixels": int((out > 0).sum()),
                                   "note": "dispatch(): host labelling / assignment / dilation + bilateral filter and the batched DenseCRF on the device"}
print(json.dumps(res))
