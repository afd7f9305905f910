"""Micro-benchmark of mit_conv_gemm tile configurations on the LaMa / OCR / ctd conv shapes (GPU box only).

Interleaved rounds inside one process (guide §5.4 rule 24); every configuration's output is checked bit-for-bit against
configuration 0 (same k-ordered accumulation)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manga_image_translator_amd import ops, lib

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4"))
def _tiles(names):
    """tile names -> indices (mit_conv_gemm_config_name); names a default build does not carry (MIT_CONV_EXPERIMENTS tiles) are skipped"""
    L_, by, i = lib.load(), {}, 0
    while L_.mit_conv_gemm_config_name(i) is not None:
        by[L_.mit_conv_gemm_config_name(i).decode()] = i
        i += 1
    return [by[n] for n in names.split(',') if n in by]


WIDE = _tiles(os.environ.get('WIDE', 'fast128x128x16w4b,fast128x128x16w4c,fast256x128x16w2c'))
NARROW = _tiles(os.environ.get('NARROW', 'fast128x64x16,fast128x64x16c,fast128x64x16w5c'))
SHAPES = [
    # name, Cin, Cout, H, W, k, s, p, mode, batch, cfgs
    ("lama fused 512->128 3x3", 512, 128, 256, 182, 3, 1, 1, ops.PAD_REFLECT, B, WIDE),
    ("lama l2g 128->384 3x3", 128, 384, 256, 182, 3, 1, 1, ops.PAD_REFLECT, B, WIDE),
    ("lama st1 384->192 1x1", 384, 192, 256, 182, 1, 1, 0, ops.PAD_ZERO, B, NARROW),
    ("lama st2 192->384 1x1", 192, 384, 256, 182, 1, 1, 0, ops.PAD_ZERO, B, WIDE),
    ("lama down2 128->256 3x3 s2", 128, 256, 1024, 728, 3, 2, 1, ops.PAD_REFLECT, B, WIDE),
    ("lama down1 64->128 3x3 s2", 64, 128, 2048, 1456, 3, 2, 1, ops.PAD_REFLECT, 1, WIDE),
    ("lama stem 4->64 7x7", 4, 64, 2048, 1456, 7, 1, 3, ops.PAD_REFLECT, 1, _tiles("128x64x16,256x64x16")),
    ("ocr grp pw1 80->320 (1.2M rows)", 80, 320, 1200, 1024, 1, 1, 0, ops.PAD_ZERO, 1, WIDE),
    ("ocr grp pw2 320->80 (1.2M rows)", 320, 80, 1200, 1024, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ocr grp pw2 1280->320 (150k rows)", 1280, 320, 150, 1024, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ocr grp pw2 640->160 (300k rows)", 640, 160, 300, 1024, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ocr grp pw1 160->640 (722k rows)", 160, 640, 705, 1024, 1, 1, 0, ops.PAD_ZERO, 1, WIDE),
    ("ocr grp pw2 640->160 (722k rows)", 640, 160, 705, 1024, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ocr grp pw1 320->1280 (360k rows)", 320, 1280, 352, 1024, 1, 1, 0, ops.PAD_ZERO, 1, WIDE),
    ("ocr grp pw2 1280->320 (360k rows)", 1280, 320, 352, 1024, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ocr pw1 320->1280 (16x6x128)", 320, 1280, 96, 128, 1, 1, 0, ops.PAD_ZERO, 1, WIDE),
    ("ocr pw2 1280->320 (16x6x128)", 1280, 320, 96, 128, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ocr pw1 160->640 (16x12x128)", 160, 640, 192, 128, 1, 1, 0, ops.PAD_ZERO, 1, WIDE),
    ("ocr pw2 640->160 (16x12x128)", 640, 160, 192, 128, 1, 1, 0, ops.PAD_ZERO, 1, NARROW),
    ("ctd 3x3 128->128 @128^2 x8", 128, 128, 128, 128, 3, 1, 1, ops.PAD_ZERO, 8, WIDE),
    ("ctd 1x1 512->256 @64^2 x8", 512, 256, 64, 64, 1, 1, 0, ops.PAD_ZERO, 8, WIDE),
]
only = os.environ.get("ONLY")
ROUNDS = int(os.environ.get("ROUNDS", "3"))
ACT = {"relu": ops.ACT_RELU, "gelu": ops.ACT_GELU, "none": ops.ACT_NONE, "silu": ops.ACT_SILU}[os.environ.get("ACT", "relu")]
res = []
L = lib.load()
for name, Cin, Cout, H, W, k, s, p, mode, b, cfgs in SHAPES:
    if only and only not in name:
        continue
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    layer = ops.Conv2d(w, None, stride=s, padding=p, pad_mode=mode, act=ACT, device=dev)
    x = torch.randn(b, H, W, layer.Cin, device=dev)
    if os.environ.get('ZERO'):
        x.zero_()
    if os.environ.get('RELU'):
        x.relu_()
    Ho, Wo = layer.out_hw(H, W)
    flops = 2.0 * b * Ho * Wo * Cout * Cin * k * k
    ref = None
    times = {c: [] for c in cfgs}
    ok = {}
    for c in cfgs:
        out = torch.zeros(b, Ho, Wo, Cout, device=dev)
        try:
            layer(x, out=out, cfg=c)
            torch.cuda.synchronize()
        except Exception as ex:
            ok[c] = f"FAILED {ex}"
            continue
        if ref is None:
            ref = out.clone()
            ok[c] = "ref"
        else:
            ok[c] = "bit-equal" if torch.equal(out, ref) else f"DIFF {(out - ref).abs().max().item():.3e}"
    out = torch.empty(b, Ho, Wo, Cout, device=dev)
    for r in range(ROUNDS):
        for c in cfgs:
            if ok[c].startswith("FAILED"):
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 3
            e0.record()
            for _ in range(n):
                layer(x, out=out, cfg=c)
            e1.record()
            torch.cuda.synchronize()
            times[c].append(e0.elapsed_time(e1) / n)
    for c in cfgs:
        if not times[c]:
            print(f"{name:32s} cfg={c} {ok[c]}")
            continue
        ms = sorted(times[c])[len(times[c]) // 2]
        cname = L.mit_conv_gemm_config_name(c).decode()
        print(f"{name:32s} B={b} {cname:16s} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF  ({ok[c]})", flush=True)
        res.append(dict(name=name, B=b, cfg=cname, ms=ms, tflops=flops / ms / 1e9, check=ok[c]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_conv.json", "w"), indent=1)
