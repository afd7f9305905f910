"""ESRGAN (RRDBNet 23 blocks, 4x) stage timing on the GPU box (BASELINE config 5's upscaler; not the contract bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import esrgan, esrgan_schema, synth
dev = torch.device("cuda:0")
H, W = int(os.environ.get("H", "1024")), int(os.environ.get("W", "720"))
eng = esrgan.EsrganEngine(synth.synth_state_dict(esrgan_schema.rrdbnet_schema(23)), nb=23, device=dev)
page = torch.from_numpy(synth.synth_page(0, H, W, n_boxes=8)[0][None]).to(dev)
eng.forward(page); torch.cuda.synchronize()
t = time.time(); n = 2
for _ in range(n):
    out = eng.forward(page)
torch.cuda.synchronize()
dt = (time.time() - t) / n
fl = eng.flops_per_input_pixel() * H * W
print(f"esrgan {H}x{W} -> {out.shape[1]}x{out.shape[2]}: {dt*1e3:.1f} ms, {fl/dt/1e12:.1f} TFLOP/s executed, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
