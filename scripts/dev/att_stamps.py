"""Phase stamps of one workgroup of the decoder's cross-attention kernel (a MIT_CONV_EXPERIMENTS build: `MIT_CONV_EXPERIMENTS=1 python -c
"import __graft_entry__ as g; g.build()"`), read after a B = 1 OCR call: where the 16-22 us of a launch go."""
import asyncio, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from manga_image_translator_amd import lib as L, pipeline, plugins as P, synth, textline as TL

run = lambda c: asyncio.new_event_loop().run_until_complete(c)
D = 6004
w = pipeline.synthetic_weights(dict_size=D)
dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(D - 4)]
page, quads, mask = synth.synth_page(0, 2048, 1456, n_boxes=32)
ocr = P.HipModel48pxOCR(weights=w["ocr48"], dictionary=dictionary)
run(ocr.load("cuda"))
lib = L.load()
lib.mit_dev_att_stamps.restype = C.c_int
names = ["start", "q ready / loads issued", "keys in LDS", "scores done", "softmax done", "values in LDS", "weighted sum done", "planes stored"]
for q2 in ("0", "1"):
    os.environ["MIT_OCR_Q2_FUSED"] = q2
    for _ in range(3):
        run(ocr.infer(page, [TL.Quadrilateral(q) for q in quads], None, False, 0, 32, True))
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 32)()
    assert lib.mit_dev_att_stamps(st) == 0
    t = [int(x) for x in st[:8]]
    c = [int(x) for x in st[8:16]]
    print(f"MIT_OCR_Q2_FUSED={q2}: " + ", ".join(f"{n} +{(b - a) * 10} ns" for n, a, b in zip(names[1:], t, t[1:])) + f"; total {(t[7] - t[0]) * 10} ns; "
          f"[q phase: rows normalised +{(int(st[16]) - t[0]) * 10} ns, K loop +{(int(st[17]) - int(st[16])) * 10} ns] shader clock {(c[7] - c[0]) / max(1, (t[7] - t[0]) * 10) :.2f} GHz ({c[7] - c[0]} cycles)")

s2 = [int(x) for x in st[20:26]]
print("self-attention (row 0, last step): " + ", ".join(f"{n} +{(b - a) * 10} ns" for n, a, b in zip(
    ["q + key history staged", "scores", "softmax", "weighted sum", "stored"], s2, s2[1:])) + f"; total {(s2[5] - s2[0]) * 10} ns")
print("  inside 'softmax': compute +%d ns, barrier +%d ns, values parked +%d ns, barrier +%d ns" % tuple((int(st[16 + b]) - int(st[16 + a])) * 10 for a, b in ((6, 10), (10, 11), (11, 12), (12, 7))))

lib.mit_dev_rows_stamps.restype = C.c_int
rs = (C.c_ulonglong * 16)()
assert lib.mit_dev_rows_stamps(rs) == 0
r_ = [int(x) for x in rs[:5]]
print("few-row GEMM (last launch, workgroup 0): " + ", ".join(f"{n} +{(b - a) * 10} ns" for n, a, b in zip(
    ["ring requested", "first step done", "K loop done", "epilogue done"], r_, r_[1:])) + f"; total {(r_[4] - r_[0]) * 10} ns")
