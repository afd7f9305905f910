"""Where the B = 1 plugin calls (detect / OCR / inpaint) and the mask refinement between them spend their time on the GPU box:
wall clock per call, a host cProfile of one call each, and (run the same command under ``rocprofv3 --kernel-trace --stats``)
the kernel side.  Not the contract bench; evidence for DESIGN §8."""
import asyncio, cProfile, io, json, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import hostglue as HG, mask_refinement as MR, pipeline, plugins as P, synth, textline as TL

run = lambda c: asyncio.new_event_loop().run_until_complete(c)
D = 6004
H, W = 2048, 1456
w = pipeline.synthetic_weights(dict_size=D)
dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(D - 4)]
pages = [synth.synth_page(i, H, W, n_boxes=32) for i in range(3)]
det = P.HipComicTextDetector(weights=w)
ocr = P.HipModel48pxOCR(weights=w["ocr48"], dictionary=dictionary)
inp = P.HipLamaMPEInpainter(weights=w)
for p in (det, ocr, inp):
    run(p.load("cuda"))
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["detect", "ocr", "inpaint", "maskref"]


def calls(i):
    page, quads, mask = pages[i % len(pages)]
    lines = [TL.Quadrilateral(q) for q in quads]
    region = type("Region", (), {"lines": np.stack([l.pts for l in lines]).astype(np.int64)})()
    return {
        "detect": lambda: run(det.infer(page, 1024, 0.5, 0.7, 2.3)),
        "ocr": lambda: run(ocr.infer(page, [TL.Quadrilateral(q) for q in quads], None, False, 0, 32, True)),
        "inpaint": lambda: run(inp.infer(page, mask, None, 2048)),
        "maskref": lambda: MR.dispatch_sync([region], page, mask.copy()),
    }


out = {}
for name in which:
    for i in range(2):
        calls(i)[name]()
    torch.cuda.synchronize()
    ts = []
    for i in range(6):
        f = calls(i)[name]
        torch.cuda.synchronize()
        t = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    out[name] = {"ms_mean": round(float(np.mean(ts)), 2), "ms_min": round(min(ts), 2), "ms_all": [round(t, 1) for t in ts]}
    if not os.environ.get("PROF_B1_NO_CPROFILE"):
        pr = cProfile.Profile()
        f = calls(0)[name]
        pr.enable(); f(); torch.cuda.synchronize(); pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
        print(f"==== cProfile of one {name} call ====", file=sys.stderr)
        print(s.getvalue()[:6000], file=sys.stderr)
print(json.dumps(out))
