"""Where the B = 1 OCR plugin call spends its time (host profile + wall clock), on the GPU box."""
import asyncio, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import pipeline, plugins as P, synth, textline as TL
run = lambda c: asyncio.new_event_loop().run_until_complete(c)
D = 6004
w = pipeline.synthetic_weights(dict_size=D)
dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(D - 4)]
page, quads, mask = synth.synth_page(0, 2048, 1456, n_boxes=32)
ocr = P.HipModel48pxOCR(weights=w["ocr48"], dictionary=dictionary)
run(ocr.load("cuda"))
class Cfg: prob = 0.0
def once():
    lines = [TL.Quadrilateral(q) for q in quads]
    return run(ocr.infer(page, lines, Cfg(), max_seq_length=32, suppress_eos=True))
for _ in range(2): once()
torch.cuda.synchronize()
t = time.time()
for _ in range(3): once()
torch.cuda.synchronize()
print(f"OCR plugin B=1: {(time.time() - t) / 3 * 1e3:.1f} ms per page (32 lines, 32 steps)")
pr = cProfile.Profile(); pr.enable(); once(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
