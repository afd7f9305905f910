"""Micro-benchmark of mit_conv_gemm on the LaMa/ctd conv shapes (GPU box only)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manga_image_translator_amd import ops, lib

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4"))
SHAPES = [
    # name, Cin, Cout, H, W, k, s, p, mode
    ("g2l 384->128 3x3", 384, 128, 256, 182, 3, 1, 1, ops.PAD_REFLECT),
    ("l2g 128->384 3x3", 128, 384, 256, 182, 3, 1, 1, ops.PAD_REFLECT),
    ("l2l 128->128 3x3", 128, 128, 256, 182, 3, 1, 1, ops.PAD_REFLECT),
    ("fused 512->128 3x3", 512, 128, 256, 182, 3, 1, 1, ops.PAD_REFLECT),
    ("st1 384->192 1x1", 384, 192, 256, 182, 1, 1, 0, ops.PAD_ZERO),
    ("fu 384->384 1x1 (256x92)", 384, 384, 256, 92, 1, 1, 0, ops.PAD_ZERO),
    ("st2 192->384 1x1", 192, 384, 256, 182, 1, 1, 0, ops.PAD_ZERO),
    ("down 64->128 3x3 s2", 64, 128, 2048, 1456, 3, 2, 1, ops.PAD_REFLECT),
    ("stem 4->64 7x7", 4, 64, 2048, 1456, 7, 1, 3, ops.PAD_REFLECT),
    ("out 64->3 7x7", 64, 3, 2048, 1456, 7, 1, 3, ops.PAD_REFLECT),
]
cfgs = [int(c) for c in os.environ.get("CFGS", "-1,0,3,4").split(",")]
res = []
for name, Cin, Cout, H, W, k, s, p, mode in SHAPES:
    b = B if H <= 512 else 1
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    layer = ops.Conv2d(w, None, stride=s, padding=p, pad_mode=mode, act=ops.ACT_RELU, device=dev)
    x = torch.randn(b, H, W, layer.Cin, device=dev)
    Ho, Wo = layer.out_hw(H, W)
    out = torch.empty(b, Ho, Wo, Cout, device=dev)
    flops = 2.0 * b * Ho * Wo * Cout * Cin * k * k
    for cfg in cfgs:
        if cfg == 4 and Cin % 32:
            continue
        try:
            for _ in range(2):
                layer(x, out=out, cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                layer(x, out=out, cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            tf = flops / ms / 1e9
            cname = "auto" if cfg < 0 else lib.load().mit_conv_gemm_config_name(cfg).decode()
            print(f"{name:28s} B={b} cfg={cname:12s} {ms:8.3f} ms  {tf:7.1f} TFLOP/s", flush=True)
            res.append(dict(name=name, B=b, cfg=cname, ms=ms, tflops=tf))
        except Exception as ex:
            print(name, cfg, "FAILED", ex)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_conv.json", "w"), indent=1)
