"""BASELINE config 5 on one GPU (evidence run, not the contract bench): a 2048x1440 page through ESRGAN 4x (-> 8192x5760,
the tensor a --upscale-ratio 2 run resizes to 4096x2880) and a 2048x1456 page through lama_large (18 FFC blocks).
Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import esrgan, esrgan_schema, lama, lama_schema, synth

dev = torch.device("cuda:0")
out = {}
H, W = 2048, 1440
eng = esrgan.EsrganEngine(synth.synth_state_dict(esrgan_schema.rrdbnet_schema(23)), nb=23, device=dev)
page = torch.from_numpy(synth.synth_page(0, H, W, n_boxes=16)[0][None]).to(dev)
eng.forward(page); torch.cuda.synchronize()
t = time.time(); up = eng.forward(page); torch.cuda.synchronize(); dt = time.time() - t
out["esrgan_4x"] = dict(input=[H, W], output=list(up.shape[1:3]), ms=round(dt * 1e3, 1),
                        exec_tflops=round(eng.flops_per_input_pixel() * H * W / dt / 1e12, 1),
                        peak_mem_gib=round(torch.cuda.max_memory_allocated() / 2**30, 1))
del eng, up; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
H, W = 2048, 1456
leng = lama.LamaEngine(synth.synth_state_dict(lama_schema.lama_generator_schema(18)), None, n_blocks=18, device=dev)
pages, masks = zip(*[(p, m) for p, _, m in (synth.synth_page(i, H, W) for i in range(4))])
img, msk = torch.from_numpy(np.stack(pages)).to(dev), torch.from_numpy(np.stack(masks)).to(dev)
leng.forward(img, msk); torch.cuda.synchronize()
t = time.time(); leng.forward(img, msk); torch.cuda.synchronize(); dt = (time.time() - t) / 4
out["lama_large"] = dict(page=[H, W], ms_per_page=round(dt * 1e3, 1), alg_tflops=round(leng.flops_per_page(H, W) / dt / 1e12, 1))
out["pages_per_s_1gpu"] = round(1.0 / (out["esrgan_4x"]["ms"] / 1e3 + dt), 3)
print(json.dumps(out))
