"""Which co-running kernel disturbs mit_rfft_rows?  Process B loops ONE kernel kind; process A repeats the same rfft_rows launch and counts
launches whose output leaves the quiet reference.  usage: diag_rfft_culprit.py <kind>   kind = att142 | att64 | gemm | rfft | ctd | ocr | lama"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.multiprocessing as mp

def other(go, stop, kind):
    go.wait()
    if kind == "gemmfp32":
        os.environ["MIT_GEMM_SPLIT"] = "0"   # the co-runner's GEMM on the fp32 MFMA tiles (conv_gemm_fast_kernel)
        kind = "gemm"
    from manga_image_translator_amd import lib as L, ops, lama, pipeline
    import diag_concurrent2 as D2
    lib = L.load()
    dev = torch.device("cuda:0")
    st = C.c_void_p(ops.current_stream())
    g = torch.Generator().manual_seed(2)
    if kind.startswith("att"):
        T, heads, hd, R = int(kind[3:]), 4, 80, 64
        E = heads * hd
        q, k, v = (torch.randn(R, T, E, generator=g).to(dev) for _ in range(3))
        out = torch.empty_like(q)
        klen = torch.full((R,), T, dtype=torch.int32, device=dev)
        f = lambda: L.check(lib.mit_attention_heads(q.data_ptr(), T * E, E, k.data_ptr(), T * E, E, v.data_ptr(), T * E, E, out.data_ptr(), T * E, E,
                                                    klen.data_ptr(), R, T, T, 1, heads, hd, st), "att")
    elif kind in ("gemm", "gemmn64"):   # gemmn64: N = 64 -> the 128x64 split tile (same bf16 MFMA loop, no scratch spills)
        layer = ops.Conv2d(torch.randn(64 if kind == "gemmn64" else 256, 256, 1, 1, generator=g) * 0.05, None, device=dev)
        x = torch.randn(8, 128, 128, 256, generator=g).to(dev)
        f = lambda: layer(x)
    elif kind == "rfft":
        B, h, w, Cc = 4, 64, 48, 192
        t1 = torch.randn(B, h, w, Cc, generator=g).to(dev)
        wk = w // 2 + 1; plane = h * wk * Cc
        Y = torch.empty(B, 2, h, wk, Cc, device=dev)
        tb = lama.rfft_row_tables(w).to(dev)
        f = lambda: L.check(lib.mit_rfft_rows(t1.data_ptr(), h * w * Cc, w * Cc, Cc, Y.data_ptr(), 2 * plane, plane, wk * Cc, Cc, tb.data_ptr(), B, h, w, Cc,
                                              C.c_float(0.1), st), "rfft")
    else:
        eng = pipeline.PageEngine(pipeline.synthetic_weights(dict_size=D2.D), device=dev, dict_size=D2.D)
        pg, qd, mk = D2.inputs(dev)
        stage = {"ctd": ("detect",), "ocr": ("ocr",), "lama": ("inpaint",)}[kind]
        f = lambda: eng.run(pg, qd, mk, max_seq_length=D2.T, suppress_eos=True, stages=stage)
    while not stop.is_set():
        for _ in range(20):
            f()
        torch.cuda.synchronize()

def main_proc(go, stop, q, iters):
    from manga_image_translator_amd import lib as L, ops, lama
    lib = L.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    B, h, w, Cc = 4, 32, 24, 192
    t1 = torch.randn(B, h, w, Cc, generator=g).to(dev)
    wk = w // 2 + 1; plane = h * wk * Cc
    tables = lama.rfft_row_tables(w).to(dev)
    st = C.c_void_p(ops.current_stream())
    def run():
        Y = torch.full((B, 2, h, wk, Cc), float("nan"), device=dev)
        L.check(lib.mit_rfft_rows(t1.data_ptr(), h * w * Cc, w * Cc, Cc, Y.data_ptr(), 2 * plane, plane, wk * Cc, Cc, tables.data_ptr(), B, h, w, Cc,
                                  C.c_float(1.0 / np.sqrt(w)), st), "mit_rfft_rows")
        torch.cuda.synchronize()
        return Y
    ref = run()
    go.set(); time.sleep(14.0)
    bad = 0
    for _ in range(iters):
        bad += not torch.equal(ref, run())
    q.put(f"{bad} of {iters} rfft_rows launches differ")
    stop.set()

if __name__ == "__main__":
    kind = sys.argv[1]
    ctx = mp.get_context("spawn")
    go, stop, q = ctx.Event(), ctx.Event(), ctx.Queue()
    ps = [ctx.Process(target=other, args=(go, stop, kind)), ctx.Process(target=main_proc, args=(go, stop, q, 300))]
    for p in ps: p.start()
    print(kind, q.get(timeout=600))
    for p in ps: p.join(timeout=60)
