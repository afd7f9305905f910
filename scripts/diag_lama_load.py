"""Which LaMa tap first differs when ANOTHER PROCESS loads the GPU?  Process A: LamaEngine.forward with taps, quiet reference then repeated
under load; process B: a loop of large matmuls.  Prints, per iteration, the first tap (in network order) that differs and how much."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.multiprocessing as mp

def loader(go, stop):
    dev = torch.device("cuda:0")
    a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
    go.wait()
    while not stop.is_set():
        for _ in range(10):
            c = a @ b
        torch.cuda.synchronize()

def main_proc(go, stop, q, H, W, B):
    from manga_image_translator_amd import pipeline, synth, lama
    dev = torch.device("cuda:0")
    w = pipeline.synthetic_weights(dict_size=96)
    eng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
    pages, quads, masks = zip(*[synth.synth_page(i, H, W, n_boxes=3) for i in range(B)])
    pg, mk = torch.from_numpy(np.stack(pages)).to(dev), torch.from_numpy(np.stack(masks)).to(dev)
    def run():
        t = {}
        out = eng.forward(pg, mk, taps=t)
        torch.cuda.synchronize()
        t["out"] = out.clone()
        return t
    run()
    ref = run(); ref2 = run()
    order = list(ref.keys())
    q.put(("quiet repeat identical", all(torch.equal(ref[k], ref2[k]) for k in order), order))
    go.set(); time.sleep(1.0)
    for it in range(8):
        o = run()
        first = next((k for k in order if not torch.equal(ref[k], o[k])), None)
        if first is None:
            q.put((f"iter {it}", "identical"))
        else:
            d = (ref[first].float() - o[first].float()).abs()
            q.put((f"iter {it}", f"first differing tap {first}: {int((d > 0).sum())} of {d.numel()} values, max abs {float(d.max()):.3e} (ref max {float(ref[first].float().abs().max()):.3e})"))
    stop.set()
    q.put(("done", None))

if __name__ == "__main__":
    H, W, B = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 192, 4)))
    ctx = mp.get_context("spawn")
    go, stop, q = ctx.Event(), ctx.Event(), ctx.Queue()
    ps = [ctx.Process(target=loader, args=(go, stop)), ctx.Process(target=main_proc, args=(go, stop, q, H, W, B))]
    for p in ps: p.start()
    while True:
        m = q.get(timeout=600)
        if m[0] == "done": break
        print(*m)
    for p in ps: p.join()
