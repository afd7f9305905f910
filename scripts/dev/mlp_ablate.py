"""Times mit_convnext_mlp (C = 80) alone at the bench's row count; with a MIT_CONV_EXPERIMENTS=1 build and MIT_MLP_ABLATE=<bits> in the
environment it times the ablated forms (wrong results): 1 no GELU, 2 no weight staging, 4 no second contraction, 8 no first."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manga_image_translator_amd import ocr48, ocr_schema, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
sd = synth.synth_state_dict(ocr_schema.ocr48_schema(64))
with ops.gemm_mode(6):
    blk = ocr48._Block(sd, "backbone.block1.0", 80, 7, dev)
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2898048
    t = torch.randn(rows, 80, device=dev)
    x = torch.randn(rows, 80, device=dev)
    h = torch.empty(rows, 320, device=dev)
    for fused in (True, False):
        ocr48.set_fused_mlp(fused)
        for _ in range(3):
            blk.mlp(t, h, x, rows)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            blk.mlp(t, h, x, rows)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print(f"ablate={os.environ.get('MIT_MLP_ABLATE', '0')} fused={fused} rows={rows}: {ms:.3f} ms per block pair "
              f"({2 * 2 * rows * 80 * 320 / ms / 1e9:.1f} TFLOP/s fp32-equivalent)", flush=True)
