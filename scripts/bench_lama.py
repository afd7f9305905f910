"""Quick LaMa stage timing on the GPU box (not the contract bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import lama, lama_schema, synth
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4")); H, W = 2048, 1456
nb = int(os.environ.get("NB", "9"))
sd = synth.synth_state_dict(lama_schema.lama_generator_schema(nb))
mpe_sd = synth.synth_state_dict(lama_schema.lama_mpe_schema()) if nb == 9 else None
eng = lama.LamaEngine(sd, mpe_sd, n_blocks=nb, device=dev, row_packed_stem=not os.environ.get("PLAIN_STEM"))
pages, masks = zip(*[(p, m) for p, _, m in (synth.synth_page(i) for i in range(B))])
img = torch.from_numpy(np.stack(pages)).to(dev); msk = torch.from_numpy(np.stack(masks)).to(dev)
for _ in range(2):
    eng.forward(img, msk)
torch.cuda.synchronize()
t = time.time(); n = 3
for _ in range(n):
    eng.forward(img, msk)
torch.cuda.synchronize()
dt = (time.time() - t) / n
fl = eng.flops_per_page(H, W) * B
print(f"B={B} blocks={nb}: {dt*1e3:.1f} ms/batch, {dt/B*1e3:.2f} ms/page, {fl/dt/1e12:.1f} TFLOP/s algorithmic, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
