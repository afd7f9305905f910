"""Do the GEMM tiles give the same bits when another stream's kernels share the CUs?

A victim GEMM (OCR-sized: a few thousand rows, K = 320) runs REPS times on one stream while a disturber loop (LaMa-sized GEMMs, or none)
runs on another; every victim output is compared with the output of the same launch on an idle GPU.
usage: python scripts/dev/gemm_cotenant.py   (modes and shapes are swept inside)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from manga_image_translator_amd import lib as L, ops
from manga_image_translator_amd.ocr48 import Linear

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
g = torch.Generator().manual_seed(1)
REPS = int(os.environ.get("REPS", "300"))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def make(M, K, N, mode):
    ops.set_split_mode(mode)
    lin = Linear(torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g), dev)
    x = torch.randn(M, K, generator=g).to(dev)
    return lin, x

def run(victim, disturber, vmode, dmode):
    (vl, vx), M, N = victim, victim[1].shape[0], victim[0].N
    ops.set_split_mode(vmode)
    ref = torch.empty(M, N, device=dev)
    vl(vx, ref)
    torch.cuda.synchronize()
    outs = torch.zeros(REPS, M, N, device=dev)
    torch.cuda.synchronize()
    if disturber is not None:
        dl, dx = disturber
        dout = torch.empty(dx.shape[0], dl.N, device=dev)
        ops.set_split_mode(dmode)
        with torch.cuda.stream(sa):
            for _ in range(REPS * 2):
                dl(dx, dout)
    ops.set_split_mode(vmode)
    with torch.cuda.stream(sb):
        for i in range(REPS):
            vl(vx, outs[i])
    torch.cuda.synchronize()
    bad = (outs != ref).reshape(REPS, -1).any(1)
    nbad = int(bad.sum())
    el = int((outs != ref).sum())
    return nbad, el

for vm in (6, 0):
    for (M, K, N) in ((800, 320, 960), (3200, 320, 960), (6400, 320, 320), (3200, 320, 1280), (3200, 1280, 320), (12800, 160, 320)):
        victim = make(M, K, N, 6)
        for dm, dshape in ((None, None), (6, (65536, 512, 512)), (0, (65536, 512, 512)), (6, (8192, 320, 960))):
            dist = None if dm is None else make(*dshape, 6)
            nbad, el = run(victim, dist, vm, dm if dm is not None else vm)
            print(f"victim mode {vm} [{M} x {K}] @ [{K} x {N}]   disturber {('mode %d %s' % (dm, dshape)) if dm is not None else 'none':34s}"
                  f" {nbad:4d} of {REPS} launches differ ({el} elements)", flush=True)
