import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import ctd, ctd_schema as S, synth
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "16"))
g = S.CTD_GAIN
eng = ctd.CtdEngine(synth.synth_state_dict(S.yolo_schema(), gain=g), synth.synth_state_dict(S.unet_head_schema(), gain=g),
                    synth.synth_state_dict(S.db_head_schema(), gain=g), device=dev)
pages = torch.from_numpy(np.stack([synth.synth_page(i)[0] for i in range(B)])).to(dev)
for _ in range(2):
    eng.forward(pages)
torch.cuda.synchronize()
t = time.time(); n = 3
for _ in range(n):
    eng.forward(pages)
torch.cuda.synchronize()
dt = (time.time() - t) / n
print(f"ctd B={B}: {dt*1e3:.1f} ms/batch {dt/B*1e3:.2f} ms/page {eng.flops_per_page()*B/dt/1e12:.1f} TFLOP/s, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
