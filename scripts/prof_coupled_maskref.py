"""Host profile of mask_refinement.dispatch on the coupled scenario (detector with the stand-in head -> refine_mask -> OCR stub ->
merge -> dispatch): where its time goes on a page whose raw mask comes out of the detector's own refine_mask."""
import asyncio, cProfile, io, json, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy import ndimage
from manga_image_translator_amd import coupled, ctd as CTD, mask_refinement as MR, pipeline, plugins as P, synth, textline_merge as TM
warnings.simplefilter("ignore", RuntimeWarning)
run = lambda c: asyncio.new_event_loop().run_until_complete(c)
H, W = 2048, 1456
w = pipeline.synthetic_weights(dict_size=6004)
page, quads, _ = synth.synth_page(0, H, W, n_boxes=32, disjoint=True)
nh, nw, dw, dh = CTD.CtdEngine.letterbox_geometry(H, W)
prob, m = coupled.synthetic_head_outputs(page, quads, (CTD.INPUT_SIZE - dh, CTD.INPUT_SIZE - dw))
dev = torch.device("cuda:0")
inj = {"prob": torch.from_numpy(prob[None]).to(dev), "mask": torch.from_numpy(m[None]).to(dev)}
det = P.HipComicTextDetector(weights=w)
run(det.load("cuda"))
plain = det.engine.forward
def fwd(pages_u8, taps=None):
    mm, lines, pad = plain(pages_u8, taps)
    lines[:, 0] = inj["prob"]
    return inj["mask"], lines, pad
det.engine.forward = fwd
tls, mask_raw, _ = run(det.infer(page, 1024, 0.5, 0.7, 2.3))
for t in tls:
    t.text, t.prob = "漢字", 0.9
regions = TM.dispatch_sync(tls, W, H)
small = MR.HG.resize_linear_u8(mask_raw, (int(W * 0.667), int(H * 0.667)))
lab, n = ndimage.label(small > 0, structure=np.ones((3, 3)))
print(json.dumps({"lines": len(tls), "regions": len(regions), "raw_mask_coverage": float((mask_raw > 0).mean()), "components_at_working_scale": int(n)}))
for _ in range(2):
    MR.dispatch_sync(regions, page, mask_raw.copy(), "fit_text", 20, 0, False, 3)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t = time.perf_counter(); out = MR.dispatch_sync(regions, page, mask_raw.copy(), "fit_text", 20, 0, False, 3); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("dispatch ms", [round(x, 1) for x in ts], "coverage", float((out > 0).mean()))
pr = cProfile.Profile(); pr.enable(); MR.dispatch_sync(regions, page, mask_raw.copy(), "fit_text", 20, 0, False, 3); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:3500])
