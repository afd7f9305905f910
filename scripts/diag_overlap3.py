"""Third diagnosis: one conv layer looped on stream A, checked bit for bit against its result in isolation, while stream B loops
(a) another conv layer of ours, (b) a Winograd layer, (c) torch.mm, (d) nothing — in GEMM modes 6 and 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manga_image_translator_amd import ops
cuda = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
xm = torch.randn(4096, 4096, device=cuda)
for mode in (6, 0):
    with ops.gemm_mode(mode):
        A = ops.Conv2d(torch.randn(128, 128, 3, 3, generator=g) * 0.03, None, padding=1, pad_mode=ops.PAD_REFLECT, act=ops.ACT_RELU, device=cuda)
        Bc = ops.Conv2d(torch.randn(320, 1280, 1, 1, generator=g) * 0.03, None, act=ops.ACT_NONE, device=cuda)
        Wg = ops.WinogradConv3x3(torch.randn(128, 512, 3, 3, generator=g) * 0.02, None, pad_mode=ops.PAD_REFLECT, act=ops.ACT_RELU, device=cuda)
        xa = torch.randn(4, 256, 184, 128, generator=g).to(cuda)
        xb = torch.randn(1, 256, 256, 1280, generator=g).to(cuda)
        xw = torch.randn(2, 256, 184, 512, generator=g).to(cuda)
        ref = A(xa).clone()
        refw = Wg(xw).clone()
        torch.cuda.synchronize()
        outs = [torch.empty_like(ref) for _ in range(12)]
        def loadB(kind):
            if kind == "conv":
                ob = torch.empty(1, 256, 256, 320, device=cuda)
                for _ in range(30): Bc(xb, out=ob)
            elif kind == "wino":
                for _ in range(10): Wg(xw)
            elif kind == "mm":
                for _ in range(30): torch.mm(xm, xm)
        for kind in ("none", "conv", "wino", "mm"):
            bad = 0
            for rep in range(3):
                sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sB):
                    loadB(kind)
                with torch.cuda.stream(sA):
                    for o in outs: A(xa, out=o)
                torch.cuda.synchronize()
                bad += sum(int(not torch.equal(o, ref)) for o in outs)
            print(f"mode {mode}: conv A on stream A, '{kind}' on stream B -> {bad} of {3 * len(outs)} outputs differ from the isolated result", flush=True)
        # and the Winograd layer as the victim
        for kind in ("none", "conv"):
            bad = 0
            for rep in range(3):
                sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sB):
                    loadB(kind)
                with torch.cuda.stream(sA):
                    res = [Wg(xw).clone() for _ in range(4)]
                torch.cuda.synchronize()
                bad += sum(int(not torch.equal(o, refw)) for o in res)
            print(f"mode {mode}: Winograd layer on stream A, '{kind}' on stream B -> {bad} of 12 differ", flush=True)
