import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manga_image_translator_amd import ocr48, ocr_schema, synth
dev = torch.device("cuda:0")
D = 6004
P = int(os.environ.get("P", "8"))
eng = ocr48.Ocr48Engine(synth.synth_state_dict(ocr_schema.ocr48_schema(D)), D, device=dev)
rng = np.random.default_rng(0)
crops = [rng.integers(0, 256, size=(48, int(rng.integers(180, 600)), 3), dtype=np.uint8) for _ in range(32 * P)]
for _ in range(2):
    eng.recognize(crops, max_seq_length=32, suppress_eos=True)
torch.cuda.synchronize()
t = time.time(); n = 3
for _ in range(n):
    out = eng.recognize(crops, max_seq_length=32, suppress_eos=True)
torch.cuda.synchronize()
dt = (time.time() - t) / n
print(f"ocr48 {P} pages x 32 lines: {dt*1e3:.1f} ms, {dt/P*1e3:.2f} ms/page, steps {out['steps_run']}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
# split timing
torch.cuda.synchronize(); t = time.time()
enc = [eng.encode(torch.from_numpy(r).to(dev), w) for _, w, r in eng.make_chunks(crops)]
torch.cuda.synchronize(); print(f"  encode all chunks: {(time.time()-t)*1e3:.1f} ms")
