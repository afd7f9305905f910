"""The encoder layer's operators one by one beside LaMa forwards of another stream (see ocr_cotenant.py: the encoder memory differs,
the backbone does not).   usage: python scripts/dev/ocr_ops_cotenant.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from manga_image_translator_amd import lib as L, pipeline, lama, ocr48, ops, synth
from manga_image_translator_amd.ocr48 import EMBD, FF, ACT_RELU

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.load(build_if_missing=False)
REPS = int(os.environ.get("REPS", "30"))
w = pipeline.synthetic_weights()
leng = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe"), n_blocks=9, device=dev)
oeng = ocr48.Ocr48Engine(w["ocr48"], pipeline.DICT_SIZE, device=dev)
page, quads, mask = synth.synth_page(3, bench.H, bench.W, n_boxes=bench.N_BOXES)
pages = torch.from_numpy(np.stack([page] * 4)).to(dev)
masks = torch.from_numpy(np.stack([mask] * 4)).to(dev)
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(3)
N, Lm = 16, 152
M = N * Lm
ly = oeng.enc[0]
mem0 = torch.randn(M, EMBD, generator=g).to(dev)
klen = torch.tensor([min(40 + 7 * i, Lm) for i in range(N)], dtype=torch.int32).to(dev)
LE = Lm * EMBD
minpos = -((Lm + 1) // 2)

def op_ln():
    out = torch.empty(M, EMBD, device=dev); oeng._layernorm(mem0, *ly.ln[0], out); return out
nrm = op_ln()
def op_qkv():
    out = torch.empty(3, M, EMBD, device=dev); ly.qkv(nrm, out[0], nsplit=EMBD, nhi=M * EMBD); return out
qkv = op_qkv()
qkv_sum = float(qkv.double().sum())
def op_qkv_plain():   # the same Linear without the column split
    out = torch.empty(M, 3 * EMBD, device=dev); ly.qkv(nrm, out); return out
def op_rot(down):
    out = torch.empty(M, EMBD, device=dev); oeng._rotate(qkv[1 if down else 0], out, N, Lm, 0, minpos, down, LE, EMBD, LE, EMBD); return out
qr, kr = op_rot(False), op_rot(True)
def op_att():
    out = torch.empty(M, EMBD, device=dev); oeng._attention(qr, kr, qkv[2], out, klen, N, Lm, Lm, 1, ((LE, EMBD),) * 4); return out
att = op_att()
def op_out_inplace():
    m = mem0.clone(); ly.out(att, m, post=m); return m
def op_out():
    out = torch.empty(M, EMBD, device=dev); ly.out(att, out, post=mem0); return out
def op_ff1():
    out = torch.empty(M, FF, device=dev); ly.ff1(nrm, out, act=ACT_RELU); return out
ffh = op_ff1()
def op_ff2_inplace():
    m = mem0.clone(); ly.ff2(ffh, m, post=m); return m
def op_memkv():
    k = torch.empty(M, EMBD, device=dev); v = torch.empty(M, EMBD, device=dev)
    oeng.mem_kv[0](mem0, k, nsplit=EMBD, nhi=(v.data_ptr() - k.data_ptr()) // 4); return torch.stack([k, v])

OPS = [("layernorm", op_ln), ("qkv Linear, 3-way column split", op_qkv), ("qkv Linear, plain", op_qkv_plain), ("xpos rotate q", lambda: op_rot(False)),
       ("xpos rotate k", lambda: op_rot(True)), ("attention_rows", op_att), ("out Linear + residual, in place", op_out_inplace), ("out Linear + residual", op_out),
       ("ff1 Linear + relu", op_ff1), ("ff2 Linear + residual, in place", op_ff2_inplace), ("mem_kv Linear, 2-way split into two tensors", op_memkv)]
for name, fn in OPS:
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ref = fn().clone()
    for label, n_lama in (("idle", 0), ("beside LaMa", 3)):
        torch.cuda.synchronize()
        for _ in range(n_lama):
            leng.forward(pages, masks)
        outs = []
        with torch.cuda.stream(side):
            for _ in range(REPS):
                outs.append(fn())
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o, ref)) for o in outs)
        el = next((int((o != ref).sum()) for o in outs if not torch.equal(o, ref)), 0)
        print(f"{name:44s} {label:12s} {bad:3d} of {REPS} differ   (first: {el} elements)", flush=True)
        if bad and os.environ.get("DUMP"):
            o = next(o for o in outs if not torch.equal(o, ref))
            idx = torch.nonzero(o != ref)
            rows = sorted(set(idx[:, 0].tolist()))
            print("    rows:", rows[:20], "... count", len(rows), " qkv checksum now", float(qkv.double().sum()), "at setup", qkv_sum)
            for i in idx[:6].tolist():
                print("    at", i, "ref", float(ref[tuple(i)]), "got", float(o[tuple(i)]))
            if "rotate" in name and os.environ.get("DUMP_NPZ"):
                np.savez_compressed(os.environ["DUMP_NPZ"] + ("_q" if "rotate q" in name else "_k") + ".npz", x=qkv[0 if "rotate q" in name else 1].cpu().numpy(),
                                    ref=ref.cpu().numpy(), got=o.cpu().numpy(), all_got=torch.stack(outs[:8]).cpu().numpy())
            again = fn()
            torch.cuda.synchronize()
            print("    the same call once more, GPU idle now: equal to ref =", bool(torch.equal(again, ref)))
