"""Are PageEngine results sensitive to ANOTHER PROCESS using the same GPU at the same time?  Rank 0 computes a quiet reference, then both
processes loop over the same batch concurrently and compare every iteration, field by field, with the reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.multiprocessing as mp
H, W, LINES, T, D = 256, 192, 3, 4, 96

def fields(res):
    return {"det_mask": res.det_mask, "det_shrink": res.det_shrink, "inpainted": res.inpainted, "tokens": res.ocr_tokens, "prob": res.ocr_prob,
            "colors": res.ocr_colors}

def worker(rank, ev_ref, ev_go, q, iters):
    from manga_image_translator_amd import pipeline, synth
    dev = torch.device("cuda:0")
    eng = pipeline.PageEngine(pipeline.synthetic_weights(dict_size=D), device=dev, dict_size=D)
    pages, quads, masks = zip(*[synth.synth_page(i, H, W, n_boxes=LINES) for i in range(4)])
    pg, mk = torch.from_numpy(np.stack(pages)).to(dev), torch.from_numpy(np.stack(masks)).to(dev)
    qd = [pipeline.quads_from_array(x) for x in quads]
    def run():
        r = eng.run(pg, qd, mk, max_seq_length=T, suppress_eos=True)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in fields(r).items()}
    run()
    if rank == 0:
        ref = run(); ref2 = run()
        q.put(("quiet", {k: bool(torch.equal(ref[k], ref2[k])) for k in ref}))
        torch.save({k: v.cpu() for k, v in ref.items()}, "/tmp/diag_ref.pt")
        ev_ref.set()
    else:
        ev_ref.wait()
        ref = {k: v.to(dev) for k, v in torch.load("/tmp/diag_ref.pt").items()}
        q.put(("other process, quiet start", {k: bool(torch.equal(ref[k], v)) for k, v in run().items()}))
    ev_go[rank].set()
    for e in ev_go:
        e.wait()
    bad = {k: 0 for k in ref}
    for it in range(iters):
        out = run()
        for k in ref:
            if not torch.equal(ref[k], out[k]):
                bad[k] += 1
    q.put((f"rank {rank} concurrent: iterations (of {iters}) that differ from the quiet reference", bad))

if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    ctx = mp.get_context("spawn")
    ev_ref, ev_go, q = ctx.Event(), [ctx.Event(), ctx.Event()], ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, ev_ref, ev_go, q, iters)) for r in range(2)]
    for p in ps: p.start()
    for _ in range(4):
        print(*q.get(timeout=600))
    for p in ps: p.join()
