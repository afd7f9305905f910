"""End-to-end A/B of the split-bf16 GEMM mode (MIT_GEMM_SPLIT) against the fp32 MFMA mode, in two processes (the mode is read once
per process):

    python scripts/ab_split.py dump /tmp/a.npz                       # fp32 MFMA everywhere
    MIT_GEMM_SPLIT=6 python scripts/ab_split.py dump /tmp/b.npz      # split tiles where eligible
    python scripts/ab_split.py cmp /tmp/a.npz /tmp/b.npz

dump: 4 distinct synthetic 2048x1456 pages through PageEngine.run (detector maps, OCR tokens / probabilities, inpainted bytes) and the
time of one more step.  cmp: the differences with the bars bench.py's parity leg applies against the CPU oracle (bitmap flips away from
the threshold, mask bytes, token ids, probabilities, inpainted bytes)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(path, n_pages=4):
    import torch

    import bench
    from manga_image_translator_amd import lib as L, ops, pipeline

    L.load(build_if_missing=False)
    dev = torch.device("cuda", 0)
    weights = pipeline.synthetic_weights()
    pages, quads, masks, _, _ = bench.make_inputs(n_pages, n_pages, 0, dev)
    eng = pipeline.PageEngine(weights, device=dev)
    res = eng.run(pages, quads, masks, max_seq_length=bench.DECODE_STEPS, suppress_eos=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = eng.run(pages, quads, masks, max_seq_length=bench.DECODE_STEPS, suppress_eos=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    order = np.asarray(res.ocr_order, np.int64)
    key = np.lexsort((order[:, 1], order[:, 0]))                 # rows by (page, line): the two modes may chunk identically, but be safe
    np.savez_compressed(path, det_shrink=res.det_shrink.cpu().numpy(), det_mask=res.det_mask.cpu().numpy(),
                        tokens=res.ocr_tokens.cpu().numpy()[key], length=res.ocr_length.cpu().numpy()[key],
                        prob=res.ocr_prob.cpu().numpy()[key], inpainted=res.inpainted.cpu().numpy())
    print(json.dumps({"mode": ops.split_mode(), "split_layers": len(ops._SPLITS), "pages": n_pages, "ms_per_page": round(dt / n_pages * 1e3, 2)}))


def cmp(a_path, b_path):
    a, b = np.load(a_path), np.load(b_path)
    out = {}
    out["det_bitmap_flips"] = int((a["det_shrink"] != b["det_shrink"]).sum())
    md = np.abs(a["det_mask"].astype(np.int32) - b["det_mask"].astype(np.int32))
    out["det_mask_u8_max_abs_diff"], out["det_mask_frac_different"] = int(md.max()), float(f"{(md != 0).mean():.3e}")
    bad = 0
    for ta, tb, la, lb in zip(a["tokens"], b["tokens"], a["length"], b["length"]):
        bad += int(la != lb or not np.array_equal(ta[:la], tb[:lb]))
    out["ocr_lines"], out["ocr_lines_with_different_tokens"] = int(len(a["tokens"])), bad
    out["ocr_max_abs_prob_diff"] = float(f"{np.abs(a['prob'] - b['prob']).max():.3e}")
    d = np.abs(a["inpainted"].astype(np.int32) - b["inpainted"].astype(np.int32))
    out["inpaint_max_abs_u8_diff"], out["inpaint_frac_bytes_different"] = int(d.max()), float(f"{(d != 0).mean():.3e}")
    out["ok"] = bool(out["det_mask_u8_max_abs_diff"] <= 1 and out["det_mask_frac_different"] < 1e-3 and bad == 0
                     and out["ocr_max_abs_prob_diff"] <= 5e-4 and out["inpaint_max_abs_u8_diff"] <= 1 and out["inpaint_frac_bytes_different"] < 1e-3)
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    else:
        sys.exit(cmp(sys.argv[2], sys.argv[3]))
