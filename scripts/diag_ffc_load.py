"""Which kernel of LaMa's first FFC layer gives load-dependent results?  Process A records the intermediate tensors of the first FFC
layer (LamaEngine._dbg) quiet, then again while process B loops the full PageEngine; prints the first intermediate that differs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.multiprocessing as mp
from diag_concurrent2 import inputs, other, D

def main_proc(go, stop, q):
    from manga_image_translator_amd import pipeline, lama
    dev = torch.device("cuda:0")
    w = pipeline.synthetic_weights(dict_size=D)
    eng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
    pg, qd, mk = inputs(dev)
    def run():
        eng._dbg = {}
        eng.forward(pg, mk)
        torch.cuda.synchronize()
        d, eng._dbg = eng._dbg, None
        return d
    run()
    ref, again = run(), run()
    order = list(ref.keys())
    q.put(("order", order))
    q.put(("quiet repeat identical", all(torch.equal(ref[k], again[k]) for k in order)))
    go.set(); time.sleep(25.0)
    for it in range(10):
        o = run()
        diff = [k for k in order if not torch.equal(ref[k], o[k])]
        if not diff:
            q.put((f"iter {it}", "identical")); continue
        k = diff[0]
        d = (ref[k] - o[k]).abs()
        q.put((f"iter {it}", f"first {k}: {int((d > 0).sum())} of {d.numel()} values, max abs {float(d.max()):.3e} (ref max {float(ref[k].abs().max()):.3e}); all differing: {diff}"))
    stop.set()
    q.put(("done", None))

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    go, stop, q = ctx.Event(), ctx.Event(), ctx.Queue()
    ps = [ctx.Process(target=other, args=(go, stop)), ctx.Process(target=main_proc, args=(go, stop, q))]
    for p in ps: p.start()
    while True:
        m = q.get(timeout=900)
        if m[0] == "done": break
        print(*m)
    for p in ps: p.join(timeout=60)
