"""Timing of SegDetectorRepresenter.boxes_from_bitmap: the GPU chain (csrc/ctd_boxes.hip) against the host routine (csrc/hostglue.hip) on
text-line-like maps of the detector's size.  usage: python scripts/bench_boxes.py [B ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from manga_image_translator_amd import hostglue as HG, lib as L  # noqa: E402
from tests.test_ctd_boxes_gpu import _scene  # noqa: E402

H, W = 1024, 728
STAMPS = "--stamps" in sys.argv
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 16, 64]:
    maps = np.stack([_scene(b, H, W, 32) for b in range(min(B, 8))])
    maps = np.concatenate([maps] * ((B + len(maps) - 1) // len(maps)))[:B]
    lines = torch.from_numpy(np.stack([maps, 1 - maps], axis=1)).cuda()
    for _ in range(2):
        out = HG.ctd_boxes_gpu(lines, 2 * H, 2 * W)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        out = HG.ctd_boxes_gpu(lines, 2 * H, 2 * W)
    wall = (time.perf_counter() - t) / 5 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        h = HG.boxes_from_bitmap_gpu_launch(lines[:, 0], 0.3, 2 * W, 2 * H, unclip_ratio=1.5, min_sside=2.0)
    e1.record()
    torch.cuda.synchronize()
    gpu = e0.elapsed_time(e1) / 5
    host_maps = lines.cpu().numpy()
    t = time.perf_counter()
    for b in range(min(B, 4)):
        HG.ctd_boxes(host_maps[b:b + 1], 2 * H, 2 * W)
    host = (time.perf_counter() - t) / min(B, 4) * 1e3
    t = time.perf_counter()
    lines.cpu()
    d2h = (time.perf_counter() - t) * 1e3 / B
    if STAMPS:
        lib = L.load()
        st = torch.zeros(B, 1000, 8, dtype=torch.int64, device="cuda")
        L.check(lib.mit_boxes_debug_stamps(st.data_ptr()), "stamps")
        HG.ctd_boxes_gpu(lines, 2 * H, 2 * W)
        L.check(lib.mit_boxes_debug_stamps(None), "stamps")
        s_ = st.cpu().numpy().reshape(-1, 8)
        s_ = s_[s_[:, 0] > 0]
        t0 = s_[:, 0].min()
        tot = (s_[:, 1:5].max(axis=1, initial=0) - s_[:, 0])
        k = int(np.argmax(tot))
        ph = lambda r: [round((max(r[i + 1], r[i]) - r[i]) / 100.0, 1) for i in range(4)]   # us per phase (100 MHz)
        print(f"  {len(s_)} borders; kernel span {(s_[:, 1:5].max() - t0) / 100:.0f} us; slowest border: n = {s_[k, 7]}, phases walk / rect / score / offset us = {ph(s_[k])}")
        big = s_[np.argsort(-s_[:, 7])[:3]]
        for r in big:
            print(f"    n = {r[7]}: phases {ph(r)}, start +{(r[0] - t0) / 100:.0f} us")
    print(f"B={B}: GPU chain {gpu:.3f} ms per batch ({gpu / B:.3f} per page), wall incl. collect {wall:.3f} ms; host routine {host:.3f} ms per page "
          f"+ {d2h:.3f} ms per page of D2H; boxes per page {np.mean([len(o[0]) for o in out]):.0f}")
