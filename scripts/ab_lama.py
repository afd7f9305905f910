"""A/B timing of LamaEngine switches on the GPU box (not the contract bench): FW=0/1 (W-axis FFT kernel), FH, WINO."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import lama, lama_schema, synth
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "16")); H, W = 2048, 1456
sd = synth.synth_state_dict(lama_schema.lama_generator_schema(9))
mpe_sd = synth.synth_state_dict(lama_schema.lama_mpe_schema())
pages, masks = zip(*[(p, m) for p, _, m in (synth.synth_page(i) for i in range(B))])
img = torch.from_numpy(np.stack(pages)).to(dev); msk = torch.from_numpy(np.stack(masks)).to(dev)
for fw in (1, 0, 1, 0):
    eng = lama.LamaEngine(sd, mpe_sd, n_blocks=9, device=dev, fft_w=bool(fw))
    for _ in range(2):
        eng.forward(img, msk)
    torch.cuda.synchronize()
    t = time.time(); n = 3
    for _ in range(n):
        eng.forward(img, msk)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(f"fft_w={fw} B={B}: {dt/B*1e3:.3f} ms/page", flush=True)
    eng.release_workspace(); del eng; torch.cuda.empty_cache()
