#!/usr/bin/env python3
"""Headline benchmark: pages/sec end-to-end (detect + OCR + inpaint) on 2048x1456 pages (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One *step* = one pass of the dense hot path over one batch of ``--pages`` synthetic pages per GPU (BASELINE config 3:
64 pages, full detect -> OCR -> inpaint).  Pages, text-line quads and inpainting masks are resident in HBM before the
timed region; weak scaling (every rank owns ``--pages`` pages); rank 0 broadcasts the weights over RCCL at load and
gathers the per-page results of every step (inside the timed region).  Prints ONE JSON line on rank 0.

Extra legs (outside the timed region):
  * roofline  — one instrumented pass with HIP events around every conv_gemm launch (the C-ABI's mit_prof_* probe);
                the dominant kernel (the conv_gemm tile configuration with the most GPU time) is MFMA-bound and priced
                against the fp32 matrix peak.
  * cpu_baseline — the oracle (CPU restatement of the reference modules, same ATen ops, fp32) timed on the host cores
                on a bounded sample (one page through all three stages), rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

H, W = 2048, 1456
N_BOXES = 32
DECODE_STEPS = 32          # fixed decode length with EOS suppressed (random weights never emit EOS)
FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, no xf32 on gfx950


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pages", type=int, default=64, help="pages per GPU per step (BASELINE config 3: 64)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pages generated per rank (cycled)")
    ap.add_argument("--stages", default="detect,ocr,inpaint")
    ap.add_argument("--lama-mb", type=int, default=16)
    ap.add_argument("--ctd-mb", type=int, default=16)
    ap.add_argument("--group", type=int, default=16)
    ap.add_argument("--overlap", action="store_true", help="two streams: detector + OCR beside LaMa (+8 %% pages/s; per-kernel roofline numbers then include the stretch of concurrent kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prof-dump", default="", help="write one CSV line per conv_gemm launch of the instrumented pass to this path")
    ap.add_argument("--probe-pages", type=int, default=0, help="pages in the instrumented pass (0 = a whole step)")
    return ap.parse_args()


def make_inputs(n_pages, distinct, rank, device):
    from manga_image_translator_amd import pipeline, synth

    pages, quads, masks = [], [], []
    for i in range(min(distinct, n_pages)):
        p, q, m = synth.synth_page(rank * 100003 + i, H, W, n_boxes=N_BOXES)
        pages.append(p)
        quads.append(q)
        masks.append(m)
    idx = [i % len(pages) for i in range(n_pages)]
    pages_t = torch.from_numpy(np.stack([pages[i] for i in idx])).to(device)
    masks_t = torch.from_numpy(np.stack([masks[i] for i in idx])).to(device)
    quad_objs = [pipeline.quads_from_array(quads[i]) for i in idx]
    return pages_t, quad_objs, masks_t, (pages, quads, masks)


def roofline_leg(engine, pages, quads, masks, stages, n_probe, dump=""):
    from manga_image_translator_amd import lib as L

    lib = L.load()
    n = pages.shape[0] if n_probe <= 0 else min(n_probe, pages.shape[0])  # a whole step by default: same launch mix as the timed steps
    torch.cuda.synchronize()
    L.check(lib.mit_prof_enable(1), "mit_prof_enable")
    engine.run(pages[:n], quads[:n], masks[:n], max_seq_length=DECODE_STEPS, suppress_eos=True, stages=stages)
    torch.cuda.synchronize()
    stats = (L.MitProfStat * 32)()
    ncfg = C.c_int(0)
    L.check(lib.mit_prof_read(stats, 32, C.byref(ncfg)), "mit_prof_read")
    if dump:
        L.check(lib.mit_prof_dump(dump.encode()), "mit_prof_dump")
    L.check(lib.mit_prof_enable(0), "mit_prof_enable")
    per_cfg = {}
    for i in range(ncfg.value):
        s = stats[i]
        if s.launches:
            per_cfg[lib.mit_conv_gemm_config_name(i).decode()] = dict(
                launches=int(s.launches), ms=round(s.ms, 3), alg_tflops=round(s.alg_flops / (s.ms * 1e-3) / 1e12, 2),
                exec_tflops=round(s.exec_flops / (s.ms * 1e-3) / 1e12, 2))
    dom = max(range(ncfg.value), key=lambda i: stats[i].ms)  # the tile configuration with the most GPU time
    d = stats[dom]
    if not d.launches:
        return None, per_cfg
    cname = lib.mit_conv_gemm_config_name(dom).decode()
    tile = cname.replace("fast", "").split("w")[0].replace("x", ",")
    kernel = ("conv_gemm_fast_kernel<%s,...>" if cname.startswith("fast") else "conv_gemm_kernel<%s,...>") % tile
    achieved = d.alg_flops / (d.ms * 1e-3) / 1e12
    roof = dict(bound="mfma", kernel=kernel, tile_config=cname, achieved=round(achieved, 2),
                peak=FP32_MATRIX_PEAK_TFLOPS, unit="TFLOP/s", frac=round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), traffic=None,
                launches=int(d.launches), avg_launch_us=round(d.ms * 1e3 / d.launches, 2),
                alg_gflop_per_launch=round(d.alg_flops / d.launches / 1e9, 3),
                exec_tflops=round(d.exec_flops / (d.ms * 1e-3) / 1e12, 2), pages_probed=n)
    return roof, per_cfg


def cpu_baseline_leg(weights, host_inputs, stages):
    """One page through the CPU oracle of each stage (fp32, all host threads), the reference's one-page-at-a-time order."""
    from oracle import ctd as OC, lama as OL, ocr48 as OO, textline as OT

    pages, quads, masks = host_inputs
    page, q, mask = pages[0], quads[0], masks[0]
    cores = torch.get_num_threads()
    per = {}
    t_all = 0.0
    with torch.no_grad():
        if "detect" in stages:
            t = time.time()
            OC.infer_maps(weights["ctd.yolo"], weights["ctd.seg"], weights["ctd.det"], page)
            per["detect"] = time.time() - t
        if "ocr" in stages:
            t = time.time()
            crops = []
            for pts in q:
                sp, vert = OT.sort_pnts(pts)
                crops.append(OT.get_transformed_region(page, sp, "v" if vert else "h", 48))
            for _, widths, img in OO.make_chunks(crops):
                OO.infer_beam_batch_tensor(weights["ocr48"], img, widths, max_seq_length=DECODE_STEPS, suppress_eos=True)
            per["ocr"] = time.time() - t
        if "inpaint" in stages:
            t = time.time()
            OL.infer(weights["lama.gen"], weights.get("lama.mpe"), page, mask, 9)
            per["inpaint"] = time.time() - t
    t_all = sum(per.values())
    return dict(value=round(1.0 / t_all, 5), unit="pages/s", cores=cores, kind="port",
                sample=f"1 page {H}x{W} ({N_BOXES} lines, {DECODE_STEPS} decode steps) through the CPU oracle of each stage",
                seconds_per_stage={k: round(v, 2) for k, v in per.items()})


def main():
    args = parse()
    from manga_image_translator_amd import dist as D

    rank, world, local = D.init()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("MIT_DIST_BACKEND") == "gloo":  # rehearsal mode: ranks may share a GPU
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    stages = tuple(s for s in args.stages.split(",") if s)

    from manga_image_translator_amd import lib as L, pipeline

    L.load(build_if_missing=False)
    weights = pipeline.synthetic_weights() if rank == 0 else None
    weights = D.broadcast_weights(weights)          # RCCL broadcast of one flat arena at load
    engine = pipeline.PageEngine(weights, device=device, ctd_mb=args.ctd_mb, lama_mb=args.lama_mb, group=args.group,
                                 overlap=args.overlap)
    pages, quads, masks, host_inputs = make_inputs(args.pages, args.distinct, rank, device)

    gather_state = {"on": world > 1, "note": None}

    def step():
        res = engine.run(pages, quads, masks, max_seq_length=DECODE_STEPS, suppress_eos=True, stages=stages)
        if gather_state["on"]:
            try:
                D.gather_pages(res.packed())        # per-page results to rank 0 (point-to-point over xGMI)
            except Exception as ex:                 # keep the data path measurable if this RCCL build rejects gather
                gather_state["on"], gather_state["note"] = False, f"result gather disabled: {type(ex).__name__}: {ex}"
        return res

    for _ in range(args.warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0)

    roof = per_cfg = cpu = None
    if not args.no_roofline and rank == 0:
        roof, per_cfg = roofline_leg(engine, pages, quads, masks, stages, args.probe_pages, args.prof_dump)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline_leg(weights, host_inputs, stages)
    D.barrier()

    if rank == 0:
        total_pages = args.pages * world * args.steps
        value = total_pages / dt
        out = {
            "metric": "pages/sec end-to-end (detect+OCR+inpaint), 2048x1456", "value": round(value, 3), "unit": "pages/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 3: {args.pages} synthetic {H}x{W} pages per GPU, detector=ctd + ocr=48px "
                                   f"({N_BOXES} lines/page, {DECODE_STEPS} decode steps, EOS suppressed) + inpainter=lama_mpe; "
                                   "random-init weights of the reference architectures",
                       "pages_per_gpu": args.pages, "distinct_pages": min(args.distinct, args.pages), "stages": list(stages),
                       "microbatch": {"ctd": args.ctd_mb, "lama": args.lama_mb, "ocr_group": args.group},
                       "streams": 2 if args.overlap else 1,
                       "parallelism": f"pages sharded one block per GPU x{world}; RCCL weight broadcast + result gather"},
            "roofline": roof, "cpu_baseline": cpu, "conv_gemm_by_tile": per_cfg,
        }
        if gather_state["note"]:
            out["config"]["gather"] = gather_state["note"]
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
