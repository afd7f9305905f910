#!/usr/bin/env python3
"""Headline benchmark: pages/sec end-to-end (detect + OCR + inpaint) on 2048x1456 pages (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU; started outside torch.distributed.run it launches
                                                        its own N ranks through it on 127.0.0.1 — the driver's command shape works as is)
  python bench.py --gpus 2 --launch-rehearsal          the launch / rendezvous / weight broadcast / verified gather / max-over-ranks
                                                        plumbing alone on gloo, no GPU work (a rehearsal line, not a measurement)
  python bench.py --config4 ...                        BASELINE config 4 preset: 128 pages per GPU
  python bench.py --mode dropin                        the drop-in path instead: B = 1, page at a time through the plugins

One *step* = one pass of the dense hot path over one batch of ``--pages`` synthetic pages per GPU (BASELINE config 3:
64 pages, full detect -> OCR -> inpaint).  Pages, text-line quads and inpainting masks are resident in HBM before the
timed region; weak scaling (every rank owns ``--pages`` pages; global page g has the same content at any world size);
rank 0 broadcasts the weights over RCCL at load and gathers the per-page result records of every step (inside the timed
region; a failing gather is a hard error).  Prints ONE JSON line on rank 0.

Extra legs (outside the timed region, rank 0):
  * roofline     — one instrumented pass per stage: HIP events (the C-ABI's mit_prof_* probe, on the launch stream) around
                   every conv_gemm launch and every transform / FFT / element-wise kernel.  ``roofline`` prices the
                   dominant kernel (the conv_gemm tile configuration with the most GPU time, MFMA-bound) against the fp32
                   matrix peak; ``roofline.stages`` gives every stage's executed conv TFLOP/s over its whole wall time;
                   ``roofline.hbm_kernels`` the HBM-bound kernels' algorithmic GB/s against 8 TB/s.  ``traffic`` is the
                   per-launch HBM traffic from the newest committed PMC pass (profiles/r*_pmc_traffic.json,
                   scripts/pmc_stage.sh: FETCH_SIZE / WRITE_SIZE in separate counters-only runs, gfx950 corrections).
  * cpu_baseline — the oracle (CPU restatement of the reference modules, same ATen ops, fp32) on the host cores, N = 1 only:
                   a thread-count sweep per stage on a reduced sample, then 1 warm-up page + 3 timed pages at the best
                   thread count, the reference's one-page-at-a-time order.
  * parity_checked — the GPU results of those same pages against the oracle outputs the CPU leg just produced.
  * dropin       — the three plugins called page by page (B = 1) with the native host glue included: what a user of the
                   reference's plugin API gets.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
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
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16, dense (the split-bf16 tiles' pipe)


def split_pairs(tile_name: str) -> int:
    """Plane products a split-bf16 tile executes per algorithmic multiply (0 for the fp32 tiles)."""
    if not tile_name.startswith("split"):
        return 0
    return 9 if "p9" in tile_name else 6 if "p6" in tile_name else 3


def split_tile_roofline(tile_name: str, alg_tflops: float):
    """Roofline view of a split-bf16 tile (``split…p6…`` / ``…p9…``): every algorithmic FLOP is executed as 6 (or 9) bf16 MFMA products,
    so the pipe it runs on sees ``pairs x`` the algorithmic rate against the bf16 dense peak.  None for the fp32 tiles."""
    if not tile_name.startswith("split"):
        return None
    pairs = 9 if "p9" in tile_name else 6 if "p6" in tile_name else 3
    executed = alg_tflops * pairs
    return dict(plane_pairs=pairs, executed_bf16_tflops=round(executed, 1), bf16_mfma_peak=BF16_MATRIX_PEAK_TFLOPS,
                frac_of_bf16_mfma_peak=round(executed / BF16_MATRIX_PEAK_TFLOPS, 4), fp32_equivalent_tflops=round(alg_tflops, 2))
FP32_VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: vector fp32 (the VALU output convolution is priced against it)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (about 6300 achievable)
STAGE_NAMES = {"detect": "ctd", "ocr": "ocr48", "inpaint": "lama_mpe"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pages", type=int, default=None, help="pages per GPU per step (default 64 = BASELINE config 3; 128 = config 4 when --gpus 8)")
    ap.add_argument("--details", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_details.json"),
                    help="the full record of the run (every sub-measurement) goes here and to stderr; stdout carries the <= 4 KB headline")
    ap.add_argument("--config4", action="store_true", help="BASELINE config 4 preset: 128 pages per GPU (1024 pages over 8 GPUs)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 on one GPU: ESRGAN 4x + lama_large (its own line)")
    ap.add_argument("--config1", action="store_true", help="BASELINE config 1 on one GPU: default detector + 48px + lama_mpe at 1024^2, B = 1 (its own line)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic pages (global page g shows page g %% distinct); 0 = every global page is its own page "
                         "(synth_page(g), SURVEY.md §8d), the default")
    ap.add_argument("--launch-rehearsal", action="store_true",
                    help="run only the N-rank plumbing (self-launch, rendezvous, weight-arena broadcast, checksummed result gather, max-over-ranks "
                         "timing) on the gloo backend with CPU tensors and print a line marked rehearsal; needs no GPU")
    ap.add_argument("--stages", default="detect,ocr,inpaint")
    ap.add_argument("--lama-mb", type=int, default=16)
    ap.add_argument("--ctd-mb", type=int, default=16)
    ap.add_argument("--group", type=int, default=16)
    ap.add_argument("--overlap", action="store_true",
                    help="two HIP streams for the HEADLINE: LaMa on the caller's stream, detector + OCR on a side stream that joins at the end of the step "
                         "(+3-4 %% pages/s, same results).  Not the default: concurrent kernels stretch each other, so the roofline leg's per-kernel "
                         "durations would no longer be those of the timed steps; the default run reports it as the two_streams sub-measurement instead")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help=argparse.SUPPRESS)
    ap.add_argument("--mode", choices=["batch", "dropin"], default="batch", help="dropin: time the plugin path (B = 1) and print its line instead of the headline")
    ap.add_argument("--dropin-pages", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--no-coupled", action="store_true", help="skip the coupled (glue-inclusive) secondary measurement")
    ap.add_argument("--coupled-group", type=int, default=16, help="pages per pipeline slot of the coupled batch path (0: one unpipelined pass)")
    ap.add_argument("--coupled-mask-workers", type=int, default=4, help="host threads of the coupled path's per-page mask stages")
    ap.add_argument("--coupled-only", action="store_true", help="only the coupled batch measurement (A/B runs of its knobs)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short sub-measurements of BASELINE configs 1, 2 and 5 (other_configs) the default one-GPU run appends")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the two-stream sub-measurement (two_streams)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-MFMA sub-measurement (fp32_mfma) taken beside a split-mode headline")
    ap.add_argument("--fp32-steps", type=int, default=3, help="timed steps of the fp32_mfma sub-measurement (after 1 warm-up step)")
    ap.add_argument("--cpu-pages", type=int, default=3, help="timed pages of the CPU baseline (after 1 warm-up page)")
    ap.add_argument("--cpu-threads", default="8,16,32,64,128", help="thread counts swept per stage")
    ap.add_argument("--prof-dump", default="", help="write one CSV line per conv_gemm launch of the instrumented passes to this path (suffix .<stage>)")
    args = ap.parse_args(argv)
    if args.config4 or (args.pages is None and args.gpus == 8):
        args.pages = 128   # BASELINE config 4: 1024 pages over 8 GPUs; --gpus 8 without --pages IS that configuration
    if args.pages is None:
        args.pages = 64
    return args


def make_inputs(n_pages, distinct, rank, device):
    """Rank r owns the contiguous global pages [r * n_pages, (r + 1) * n_pages); global page g shows synthetic page g (``distinct`` = 0,
    the default: every page of the job is its own page) or page g % distinct, so a page's content — and therefore its result record —
    does not depend on the world size.  Returns the device batch, and the host copies of the rank's distinct pages with the batch's
    index into them (the CPU baseline / parity legs read those)."""
    from manga_image_translator_amd import pipeline, synth

    if distinct <= 0:
        ids = [rank * n_pages + i for i in range(n_pages)]
    else:
        ids = [(rank * n_pages + i) % distinct for i in range(n_pages)]
    uniq = sorted(set(ids))
    slot = {g: k for k, g in enumerate(uniq)}
    pages, quads, masks = [], [], []
    for g in uniq:
        p, q, m = synth.synth_page(g, H, W, n_boxes=N_BOXES)
        pages.append(p)
        quads.append(q)
        masks.append(m)
    idx = [slot[g] for g in ids]
    pages_t = torch.from_numpy(np.stack([pages[i] for i in idx])).to(device)
    masks_t = torch.from_numpy(np.stack([masks[i] for i in idx])).to(device)
    quad_objs = [pipeline.quads_from_array(quads[i]) for i in idx]
    return pages_t, quad_objs, masks_t, (pages, quads, masks), idx


# ---------------------------------------------------------------------------------------------------------------------
# roofline legs
# ---------------------------------------------------------------------------------------------------------------------

def _pmc_traffic():
    """Newest committed PMC summary (scripts/pmc_traffic.py output): kernel name -> per-launch HBM read / write MB."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, {}
    try:
        return os.path.relpath(files[-1], ROOT), json.load(open(files[-1]))
    except (OSError, ValueError):
        return None, {}


def _mfma_busy(fp32=False):
    """Newest committed MFMA-busy summary (scripts/pmc_mfma.sh) OF THE GEMM MODE THE LEG RUNS IN (``…_mfma_busy.json`` = the shipped split
    mode, ``…_mfma_busy_fp32.json`` = mode 0): stage -> fraction of matrix-pipe cycles used (PMC)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_busy_fp32.json" if fp32 else "r*_mfma_busy.json")))
    if not files:
        return None, {}
    try:
        return os.path.relpath(files[-1], ROOT), json.load(open(files[-1])).get("stages", {})
    except (OSError, ValueError):
        return None, {}


def _traffic_entry(pmc, kernel_name):
    for k, e in pmc.items():
        if k.startswith(kernel_name) and "hbm_read_MB_per_launch" in e and "hbm_write_MB_per_launch" in e:
            return dict(read_GB=round(e["hbm_read_MB_per_launch"] / 1e3, 4), write_GB=round(e["hbm_write_MB_per_launch"] / 1e3, 4),
                        pages=e.get("pages"))
    return None


def stage_legs(engine, pages, quads, masks, stages, dump=""):
    """One instrumented pass per stage over the whole batch (same launch mix as the timed steps)."""
    from manga_image_translator_amd import lib as L

    lib = L.load()
    ncfg_max = 64
    per_stage, conv_tot, kern_tot, alg_bytes = {}, {}, {}, {}
    n = pages.shape[0]
    for s in stages:
        torch.cuda.synchronize()
        L.check(lib.mit_prof_enable(1), "mit_prof_enable")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.run(pages, quads, masks, max_seq_length=DECODE_STEPS, suppress_eos=True, stages=(s,))
        e1.record()
        torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1)
        stats = (L.MitProfStat * ncfg_max)()
        ncfg = C.c_int(0)
        L.check(lib.mit_prof_read(stats, ncfg_max, C.byref(ncfg)), "mit_prof_read")
        kst = (L.MitProfKernelStat * 64)()
        nk = C.c_int(0)
        L.check(lib.mit_prof_kernels_read(kst, 64, C.byref(nk)), "mit_prof_kernels_read")
        import tempfile

        with tempfile.NamedTemporaryFile(suffix=".csv") as tf:   # per-launch records (tile, M, N, K, taps, Z): the algorithmic bytes per tile
            L.check(lib.mit_prof_dump(tf.name.encode()), "mit_prof_dump")
            for ln in open(tf.name).read().splitlines()[1:]:
                f = ln.split(",")
                m_, n_, k_, taps_, z_ = (float(v) for v in f[1:6])
                ab = alg_bytes.setdefault(f[0], [0, 0.0, 0.0, 0.0])
                ab[0] += 1
                ab[1] += 4.0 * z_ * m_ * k_ / max(taps_, 1.0)   # A: every input element of a stride-1 layer once (M x Cin per slice)
                ab[2] += 4.0 * z_ * k_ * n_                     # W as fp32
                ab[3] += 4.0 * z_ * m_ * n_                     # C written once
        if dump:
            L.check(lib.mit_prof_dump(f"{dump}.{s}".encode()), "mit_prof_dump")
        L.check(lib.mit_prof_enable(0), "mit_prof_enable")
        conv_ms = conv_exec = peak_ms = bf16_exec = 0.0
        for i in range(ncfg.value):
            st = stats[i]
            if not st.launches:
                continue
            conv_ms += st.ms
            conv_exec += st.exec_flops
            pairs = split_pairs(lib.mit_conv_gemm_config_name(i).decode())
            # time this configuration's launches would take at the peak of the pipe they run on
            peak_ms += (st.exec_flops * pairs / (BF16_MATRIX_PEAK_TFLOPS * 1e9)) if pairs else (st.exec_flops / (FP32_MATRIX_PEAK_TFLOPS * 1e9))
            bf16_exec += st.exec_flops * pairs
            a = conv_tot.setdefault(i, [0, 0.0, 0.0, 0.0])
            a[0] += st.launches
            a[1] += st.ms
            a[2] += st.exec_flops
            a[3] += st.alg_flops
        other_ms = 0.0
        for i in range(nk.value):
            k = kst[i]
            if k.name.decode().startswith("convnext_mlp"):   # the fused pointwise pair: split-bf16 p6 contractions outside mit_conv_gemm —
                conv_ms += k.ms                               # counted with the conv work of the stage (6 plane products per algorithmic FLOP)
                conv_exec += k.alg_flops
                peak_ms += k.alg_flops * 6 / (BF16_MATRIX_PEAK_TFLOPS * 1e9)
                bf16_exec += k.alg_flops * 6
            else:
                other_ms += k.ms
            a = kern_tot.setdefault(k.name.decode(), [0, 0.0, 0.0, 0.0])
            a[0] += k.launches
            a[1] += k.ms
            a[2] += k.alg_bytes
            a[3] += k.alg_flops
        exec_tflops = conv_exec / (wall_ms * 1e-3) / 1e12
        per_stage[STAGE_NAMES.get(s, s)] = dict(
            ms_per_page=round(wall_ms / n, 3), conv_exec_tflops=round(exec_tflops, 2),
            frac_of_fp32_mfma_peak=round(exec_tflops / FP32_MATRIX_PEAK_TFLOPS, 4),
            conv_kernel_ms_per_page=round(conv_ms / n, 3), other_probed_kernel_ms_per_page=round(other_ms / n, 3),
            conv_kernels_alone_tflops=round(conv_exec / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
            # the stage against the MFMA roofline of the pipes it actually uses: (time its contractions need at peak — split launches:
            # pairs x FLOPs at the bf16 dense peak, fp32 launches: FLOPs at the fp32 matrix peak) / the stage's wall time
            mfma_ms_at_peak_per_page=round(peak_ms / n, 3), frac_of_mfma_roofline=round(peak_ms / wall_ms, 4),
            executed_bf16_tflops=round(bf16_exec / (wall_ms * 1e-3) / 1e12, 1) if bf16_exec else 0.0)
        per_stage[STAGE_NAMES.get(s, s)]["_peak_ms"] = peak_ms
    stage_legs.alg_bytes = alg_bytes
    return per_stage, conv_tot, kern_tot, n


def roofline_leg(engine, pages, quads, masks, stages, dump=""):
    from manga_image_translator_amd import lib as L

    lib = L.load()
    per_stage, conv_tot, kern_tot, n = stage_legs(engine, pages, quads, masks, stages, dump)
    src, pmc = _pmc_traffic()
    per_cfg = {}
    for i, (launches, ms, ex, alg) in conv_tot.items():
        per_cfg[lib.mit_conv_gemm_config_name(i).decode()] = dict(
            launches=int(launches), ms=round(ms, 3), alg_tflops=round(alg / (ms * 1e-3) / 1e12, 2),
            exec_tflops=round(ex / (ms * 1e-3) / 1e12, 2), kernel=lib.mit_conv_gemm_config_kernel(i).decode())
    for name, v in per_cfg.items():
        sp = split_tile_roofline(name, v["alg_tflops"])
        if sp is not None:
            v["bf16_pipe"] = sp
    if not conv_tot:
        return None, per_cfg
    dom = max(conv_tot, key=lambda i: conv_tot[i][1])  # the tile configuration with the most GPU time
    launches, ms, ex, alg = conv_tot[dom]
    cname, kname = lib.mit_conv_gemm_config_name(dom).decode(), lib.mit_conv_gemm_config_kernel(dom).decode()
    achieved = alg / (ms * 1e-3) / 1e12
    tr = _traffic_entry(pmc, kname)
    ab = getattr(stage_legs, "alg_bytes", {}).get(cname)
    if tr is not None:
        tr["source"] = src
        tr["joined_from_committed_pmc_pass"] = True    # counters need their own profiler passes: not re-measured by this run
        if ab and ab[0]:
            algb = dict(A_GB=round(ab[1] / ab[0] / 1e9, 4), W_GB=round(ab[2] / ab[0] / 1e9, 4), C_GB=round(ab[3] / ab[0] / 1e9, 4))
            algb["total_GB"] = round(sum(algb.values()), 4)
            tr["algorithmic_per_launch"] = algb     # measured in THIS run's launch mix (every input element once, W once, C once)
            tr["ratio_to_algorithmic"] = round((tr["read_GB"] + tr["write_GB"]) / max(algb["total_GB"], 1e-9), 3)
    hbm = {}
    for name, (kl, kms, kb, kf) in sorted(kern_tot.items(), key=lambda kv: -kv[1][1]):
        e = dict(launches=int(kl), avg_us=round(kms * 1e3 / kl, 2), ms_per_page=round(kms / n, 4))
        if kb > 0:
            gbs = kb / (kms * 1e-3) / 1e9
            e.update(alg_GB_per_launch=round(kb / kl / 1e9, 4), alg_GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
        if kf > 0:
            e["alg_tflops"] = round(kf / (kms * 1e-3) / 1e12, 2)
        t = _traffic_entry(pmc, name)
        if t is not None:
            e["pmc_traffic"] = t
        hbm[name] = e
    from manga_image_translator_amd import ops as _ops_mode

    busy_src, busy = _mfma_busy(fp32=not _ops_mode.split_mode())
    for sname, key in (("ctd", "detect"), ("ocr48", "ocr"), ("lama_mpe", "inpaint")):
        if sname in per_stage and key in busy:   # PMC view of the same stage (counters-only pass of scripts/pmc_mfma.sh, committed under profiles/);
            # (the summaries' derived effective_clock_GHz is not carried over: for stages made of short kernels it came out above the chip clock)
            per_stage[sname]["mfma_busy"] = dict(frac=busy[key].get("mfma_busy"), source=busy_src, joined_from_committed_pmc_pass=True)
    total_exec = sum(v[2] for v in conv_tot.values())
    total_wall = sum(s["ms_per_page"] for s in per_stage.values()) * n
    total_peak_ms = sum(s.pop("_peak_ms") for s in per_stage.values())
    pairs = split_pairs(cname)
    if pairs:
        # The dominant kernel is a split-bf16 tile: it is priced on the pipe it runs on.  Every algorithmic multiply-add is executed
        # as `pairs` bf16 MFMA products, so achieved = pairs x algorithmic FLOPs / kernel time against the dense bf16 MFMA peak.
        executed = achieved * pairs
        head = dict(achieved=round(executed, 1), peak=BF16_MATRIX_PEAK_TFLOPS, frac=round(executed / BF16_MATRIX_PEAK_TFLOPS, 4),
                    pipe="bf16 MFMA (v_mfma_f32_32x32x16_bf16), dense", plane_pairs=pairs, fp32_equivalent_tflops=round(achieved, 2),
                    fp32_equivalent_frac_of_fp32_mfma_peak=round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4))
    else:
        head = dict(achieved=round(achieved, 2), peak=FP32_MATRIX_PEAK_TFLOPS, frac=round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4),
                    pipe="fp32 MFMA (v_mfma_f32_32x32x2_f32)")
    ov = None
    if getattr(engine, "overlap", False) and set(stages) == {"detect", "ocr", "inpaint"}:
        # The timed steps ran the stages on two streams: a kernel's duration there includes what its co-tenants cost it.  One more
        # instrumented pass, a whole step exactly as timed (events on each launch's own stream), prices the dominant tile in THAT regime;
        # the per-stage passes above (one stage, one stream) stay as the kernel's own numbers under ``solo``.
        torch.cuda.synchronize()
        L.check(lib.mit_prof_enable(1), "mit_prof_enable")
        engine.run(pages, quads, masks, max_seq_length=DECODE_STEPS, suppress_eos=True, stages=stages)
        torch.cuda.synchronize()
        stats = (L.MitProfStat * 64)()
        ncfg = C.c_int(0)
        L.check(lib.mit_prof_read(stats, 64, C.byref(ncfg)), "mit_prof_read")
        L.check(lib.mit_prof_enable(0), "mit_prof_enable")
        if dom < ncfg.value and stats[dom].launches:
            st = stats[dom]
            ov = dict(launches=int(st.launches), ms=float(st.ms), alg=float(st.alg_flops), ex=float(st.exec_flops))
    if ov is not None:
        solo = dict(achieved=head["achieved"], frac=head["frac"], avg_launch_us=round(ms * 1e3 / launches, 2))
        ach2 = ov["alg"] / (ov["ms"] * 1e-3) / 1e12
        if pairs:
            head.update(achieved=round(ach2 * pairs, 1), frac=round(ach2 * pairs / BF16_MATRIX_PEAK_TFLOPS, 4), fp32_equivalent_tflops=round(ach2, 2),
                        fp32_equivalent_frac_of_fp32_mfma_peak=round(ach2 / FP32_MATRIX_PEAK_TFLOPS, 4))
        else:
            head.update(achieved=round(ach2, 2), frac=round(ach2 / FP32_MATRIX_PEAK_TFLOPS, 4))
        head["solo"] = solo
        head["note"] = "two streams: achieved / frac / avg_launch_us are those of a whole overlapped step (as timed); solo = the kernel with its stage alone on one stream"
        launches, ms, ex = ov["launches"], ov["ms"], ov["ex"]
    roof = dict(bound="mfma", kernel=kname, tile_config=cname, unit="TFLOP/s", **head, traffic=tr, launches=int(launches),
                avg_launch_us=round(ms * 1e3 / launches, 2), alg_gflop_per_launch=round(alg / launches / 1e9, 3),
                exec_tflops=round(ex / (ms * 1e-3) / 1e12, 2), pages_probed=n,
                whole_step=dict(conv_exec_tflops=round(total_exec / (total_wall * 1e-3) / 1e12, 2),
                                frac_of_fp32_mfma_peak=round(total_exec / (total_wall * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                                frac_of_mfma_roofline=round(total_peak_ms / total_wall, 4)),
                stages=per_stage, hbm_kernels=hbm, hbm_peak_GBps=HBM_PEAK_GBS, pmc_source=src)
    return roof, per_cfg


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle on the host cores) and the parity check against it
# ---------------------------------------------------------------------------------------------------------------------

def _oracle_stage_fns(weights):
    from oracle import ctd as OC, lama as OL, ocr48 as OO, textline as OT

    def detect(page, q, mask):
        return OC.infer_maps(weights["ctd.yolo"], weights["ctd.seg"], weights["ctd.det"], page)

    def ocr(page, q, mask, max_chunks=None):
        crops = []
        for pts in q:
            sp, vert = OT.sort_pnts(pts)
            crops.append(OT.get_transformed_region(page, sp, "v" if vert else "h", 48))
        out = []
        for ci, (indices, widths, img) in enumerate(OO.make_chunks(crops)):
            if max_chunks is not None and ci >= max_chunks:
                break
            r = OO.infer_beam_batch_tensor(weights["ocr48"], img, widths, max_seq_length=DECODE_STEPS, suppress_eos=True)
            out.append((indices, r))
        return out

    def inpaint(page, q, mask):
        return OL.infer(weights["lama.gen"], weights.get("lama.mpe"), page, mask, 9)

    return dict(detect=detect, ocr=ocr, inpaint=inpaint)


def cpu_baseline_leg(weights, host_inputs, stages, n_timed, thread_counts):
    """Thread sweep per stage on a reduced sample, then 1 warm-up page + ``n_timed`` pages through the CPU oracle of each stage
    at the stage's best thread count (the reference processes one page at a time: manga_translator.py:1491-1519).
    Returns (cpu_baseline dict, oracle outputs of the processed pages for the parity leg)."""
    pages, quads, masks = host_inputs
    fns = _oracle_stage_fns(weights)
    ncpu = os.cpu_count() or 1
    counts = sorted({min(t, ncpu) for t in thread_counts if t > 0}) or [ncpu]
    default_threads = torch.get_num_threads()
    sweep, best = {}, {}
    t_leg = time.time()
    with torch.no_grad():
        for s in stages:
            sweep[s] = {}
            for nt in counts:
                torch.set_num_threads(nt)
                t = time.time()
                if s == "inpaint":   # a quarter page (the network is fully convolutional: same ops, 1/4 the pixels)
                    fns[s](np.ascontiguousarray(pages[0][:H // 2, :W // 2]), None, np.ascontiguousarray(masks[0][:H // 2, :W // 2]))
                elif s == "ocr":     # one chunk of 16 lines, all decode steps
                    fns[s](pages[0], quads[0], None, max_chunks=1)
                else:
                    fns[s](pages[0], quads[0], masks[0])
                sweep[s][nt] = round(time.time() - t, 3)
            best[s] = min(sweep[s], key=sweep[s].get)
        n_pages = min(len(pages), 1 + n_timed)
        per_page = {s: [] for s in stages}
        outputs = []
        for i in range(n_pages):
            o = {}
            for s in stages:
                torch.set_num_threads(best[s])
                t = time.time()
                o[s] = fns[s](pages[i], quads[i], masks[i])
                per_page[s].append(time.time() - t)
            outputs.append(o)
    torch.set_num_threads(default_threads)
    timed = {s: v[1:] if len(v) > 1 else v for s, v in per_page.items()}  # page 0 is the warm-up
    sec = {s: float(np.mean(v)) for s, v in timed.items()}
    total = sum(sec.values())
    n_t = len(next(iter(timed.values()))) if timed else 0
    cpu = dict(value=round(1.0 / total, 5), unit="pages/s", cores=int(max(best.values())), kind="port",
               sample=f"1 warm-up + {n_t} timed pages {H}x{W} ({N_BOXES} lines, {DECODE_STEPS} decode steps) through the CPU oracle of "
                      f"each stage, one page at a time, each stage at its best thread count of a sweep over {counts}",
               threads_per_stage={s: int(best[s]) for s in stages}, host_cpus=ncpu,
               seconds_per_stage={s: round(v, 3) for s, v in sec.items()},
               seconds_per_stage_min_max={s: [round(min(v), 3), round(max(v), 3)] for s, v in timed.items()},
               warmup_seconds_per_stage={s: round(per_page[s][0], 3) for s in stages},
               thread_sweep_seconds={s: {str(k): v for k, v in sweep[s].items()} for s in stages},
               sweep_samples={"detect": "1 page", "ocr": "1 chunk of 16 lines, 32 steps", "inpaint": "1 quarter page 1024x728"},
               leg_seconds=round(time.time() - t_leg, 1))
    return cpu, outputs


def parity_leg(res, idx, oracle_outputs, stages):
    """GPU results of the last timed step against the oracle outputs of the CPU leg (same pages, same weights).
    Bars are those of tests/: thresholded bitmap exact outside a 1e-4 margin of 0.3, mask bytes equal away from a truncation
    boundary, token ids identical with probabilities within 5e-4, inpainted bytes within 1 level."""
    out = {"pages": len(oracle_outputs)}
    ok = True
    loc = {g: i for i, g in reversed(list(enumerate(idx)))}  # first batch slot showing distinct page g
    if "detect" in stages:
        flips = near_n = mask_bad = mask_max = mask_tot = 0
        for g, o in enumerate(oracle_outputs):
            rmask, rlines = o["detect"]
            b = loc[g]
            near = np.abs(rlines[0, 0] - 0.3) < 1e-4
            got = res.det_shrink[b].cpu().numpy().astype(bool)
            flips += int((got != (rlines[0, 0] > 0.3))[~near].sum())
            near_n += int(near.sum())
            md = np.abs(res.det_mask[b].cpu().numpy().astype(np.int32) - rmask.astype(np.int32))  # rmask: postprocess_mask bytes
            mask_max, mask_bad, mask_tot = max(mask_max, int(md.max())), mask_bad + int((md != 0).sum()), mask_tot + md.size
        # what the margin means for the boxes: every in-margin pixel of the oracle map forced above / below the threshold, box extraction
        # (mit_ctd_boxes = SegDetectorRepresenter, db_utils.py:127-216) on each variant (tests/test_margin_flips.py does the same on
        # trained-head-like maps, where boxes exist)
        from manga_image_translator_amd import hostglue as _hg

        n_base = n_up = n_dn = changed = 0
        for o in oracle_outputs:
            rl = np.ascontiguousarray(o["detect"][1], dtype=np.float32)
            near = np.abs(rl[0, 0] - 0.3) < 1e-4
            b0, _ = _hg.ctd_boxes(rl, H, W)
            var = []
            for sign in (1.0, -1.0):
                q = rl.copy()
                q[0, 0][near] = np.float32(0.3 + sign * 2e-4)
                var.append(_hg.ctd_boxes(q, H, W)[0])
            n_base, n_up, n_dn = n_base + len(b0), n_up + len(var[0]), n_dn + len(var[1])
            changed += sum(int(len(v) != len(b0) or (len(v) and not np.array_equal(v, b0))) for v in var)
        out["detect"] = dict(bitmap_flips_outside_margin=flips, px_inside_margin=near_n, mask_u8_max_abs_diff=mask_max,
                             mask_u8_frac_different=float(f"{mask_bad / max(mask_tot, 1):.3e}"),
                             boxes_with_margin_pixels_forced=dict(as_is=n_base, all_above=n_up, all_below=n_dn, variants_with_any_box_changed=changed,
                                                                  note="random-init weights: the map is noise around the threshold and the representer's 1000-candidate cap is hit either way (none "
                                                                       "passes the 0.6 score filter); trained-head-like maps: tests/test_margin_flips.py"))
        ok &= flips == 0 and mask_max <= 1 and mask_bad / max(mask_tot, 1) < 1e-3
    if "ocr" in stages and res.ocr_tokens is not None:
        toks, lens, probs = res.ocr_tokens.cpu().numpy(), res.ocr_length.cpu().numpy(), res.ocr_prob.cpu().numpy()
        row_of = {pl: r for r, pl in enumerate(res.ocr_order)}
        bad = lines = 0
        dprob = 0.0
        for g, o in enumerate(oracle_outputs):
            b = loc[g]
            for indices, r in o["ocr"]:
                for j, i in enumerate(indices):
                    row = row_of[(b, i)]
                    n = int(lens[row])
                    lines += 1
                    bad += int(not np.array_equal(toks[row, 1:n], r[j][0].numpy()))
                    dprob = max(dprob, abs(float(probs[row]) - float(r[j][1])))
        out["ocr"] = dict(lines=lines, lines_with_different_tokens=bad, max_abs_prob_diff=float(f"{dprob:.3e}"))
        ok &= bad == 0 and dprob <= 5e-4
    if "inpaint" in stages:
        mx, nz, tot = 0, 0, 0
        for g, o in enumerate(oracle_outputs):
            d = np.abs(res.inpainted[loc[g]].cpu().numpy().astype(np.int32) - o["inpaint"].astype(np.int32))
            mx, nz, tot = max(mx, int(d.max())), nz + int((d != 0).sum()), tot + d.size
        out["inpaint"] = dict(max_abs_u8_diff=mx, frac_bytes_different=float(f"{nz / tot:.3e}"))
        ok &= mx <= 1 and nz / tot < 1e-3
    out["ok"] = bool(ok)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the drop-in path: page at a time through the plugins (manga_translator.py:1491-1519 calls them exactly like this)
# ---------------------------------------------------------------------------------------------------------------------

def dropin_leg(weights, host_inputs, n_pages, device_str="cuda"):
    import asyncio

    from manga_image_translator_amd import plugins as P

    pages, quads, masks = host_inputs
    run = asyncio.new_event_loop().run_until_complete
    dict_size = weights["ocr48"]["embd.weight"].shape[0]
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(dict_size - 4)]
    det = P.HipComicTextDetector(weights=weights)
    ocr = P.HipModel48pxOCR(weights=weights["ocr48"], dictionary=dictionary)
    inp = P.HipLamaMPEInpainter(weights=weights)
    for p in (det, ocr, inp):
        run(p.load(device_str))
    from manga_image_translator_amd.textline import Quadrilateral

    per = {"detect": [], "ocr": [], "inpaint": []}
    n_found = []
    n = min(n_pages, len(pages))
    for i in range(n + 1):  # page 0 twice: the first call is the warm-up (workspace allocation)
        j = max(i - 1, 0)
        page, q, mask = pages[j], quads[j], masks[j]
        torch.cuda.synchronize()
        t = time.perf_counter()
        tls, _, _ = run(det.infer(page, 1024, 0.5, 0.7, 2.3))   # network + native boxes + mask resize + refine_mask (GPU)
        t1 = time.perf_counter()
        lines = [Quadrilateral(np.asarray(pts)) for pts in q]     # the OCR stage is fed the generator's text lines
        run(ocr.infer(page, lines, None, False, 0, DECODE_STEPS, True))  # direction vote + warps + recognition + decode
        t2 = time.perf_counter()
        run(inp.infer(page, mask, None, max(H, W)))
        t3 = time.perf_counter()
        if i:
            per["detect"].append(t1 - t)
            per["ocr"].append(t2 - t1)
            per["inpaint"].append(t3 - t2)
            n_found.append(len(tls))
    for p in (det, ocr, inp):
        run(p.unload())
    ms = {k: round(1e3 * float(np.mean(v)), 2) for k, v in per.items()}
    total = sum(ms.values())
    return dict(value=round(1e3 / total, 3), unit="pages/s", pages=n, batch=1, ms_per_page=round(total, 2), ms_per_stage=ms,
                includes="host<->device copies, ctd box extraction (GPU, csrc/ctd_boxes.hip), mask resize + refine_mask (GPU), OCR direction vote "
                         "and per-line planning, plugin result decoding",
                detector_boxes_found_per_page=n_found,
                note="synthetic weights: the detector's boxes are whatever the random network fires on, so its host-glue time is not "
                     "representative of real pages; OCR gets the generator's 32 lines, the inpainter the generator's mask")


# ---------------------------------------------------------------------------------------------------------------------
# the coupled path: every stage consumes what the previous one produced, host glue included (manga_translator.py:432-622)
# ---------------------------------------------------------------------------------------------------------------------

def _coupled_inputs(n_pages, distinct, device):
    """Pages for the coupled path: the generator's text boxes kept apart from each other (synth_page(disjoint=True)) — a detector cannot
    separate the overlapping boxes of the headline pages, which are fed to the stages as ground-truth quads instead."""
    from manga_image_translator_amd import synth

    distinct = n_pages if distinct <= 0 else max(1, min(distinct, n_pages))
    gen = [synth.synth_page(i, H, W, n_boxes=N_BOXES, disjoint=True) for i in range(distinct)]
    idx = [i % distinct for i in range(n_pages)]
    pages_dev = torch.from_numpy(np.stack([gen[i][0] for i in idx])).to(device)
    return pages_dev, ([g[0] for g in gen], [g[1] for g in gen], [g[2] for g in gen]), idx


def _injection(host_inputs, idx, device, map_hw):
    """Per-page maps a trained ctd head would emit for the synthetic pages (coupled.synthetic_head_outputs): random-init weights fire
    on nothing, so the coupled path would otherwise time empty glue.  The network still runs in full; the maps replace its (random) output."""
    from manga_image_translator_amd import coupled

    pages, quads, _ = host_inputs
    per = [coupled.synthetic_head_outputs(pages[g], quads[g], map_hw) for g in range(len(pages))]
    prob = torch.from_numpy(np.stack([per[g][0] for g in idx])).to(device)
    mask = torch.from_numpy(np.stack([per[g][1] for g in idx])).to(device)
    return {"prob": prob, "mask": mask}


def _probe_kernels(fn):
    """Run ``fn`` once under the C-ABI kernel probe: (wall ms, {kernel: launches, ms, algorithmic GB/s and fraction of the HBM peak})."""
    from manga_image_translator_amd import lib as L

    lib = L.load()
    torch.cuda.synchronize()
    L.check(lib.mit_prof_enable(1), "mit_prof_enable")
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    kst = (L.MitProfKernelStat * 64)()
    nk = C.c_int(0)
    L.check(lib.mit_prof_kernels_read(kst, 64, C.byref(nk)), "mit_prof_kernels_read")
    stats = (L.MitProfStat * 64)()
    ncfg = C.c_int(0)
    L.check(lib.mit_prof_read(stats, 64, C.byref(ncfg)), "mit_prof_read")
    L.check(lib.mit_prof_enable(0), "mit_prof_enable")
    out = {}
    for i in range(nk.value):
        k = kst[i]
        if not k.launches:
            continue
        e = dict(launches=int(k.launches), ms=round(k.ms, 3))
        if k.alg_bytes > 0 and k.ms > 0:
            gbs = k.alg_bytes / (k.ms * 1e-3) / 1e9
            e.update(alg_GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
        out[k.name.decode()] = e
    conv_ms = sum(stats[i].ms for i in range(ncfg.value) if stats[i].launches)
    if conv_ms:
        out["conv_gemm (all tiles)"] = dict(launches=int(sum(stats[i].launches for i in range(ncfg.value))), ms=round(conv_ms, 3))
    return round(wall, 2), dict(sorted(out.items(), key=lambda kv: -kv[1]["ms"]))


def coupled_leg(weights, n_pages, distinct, device, steps=2, b1_pages=4, group=16, mask_workers=4, batch_only=False):
    """(1) batch: coupled.CoupledPageEngine over the same resident pages as the headline — detector -> native box extraction on a host
    thread pool -> GPU refine_mask -> OCR of the DETECTED lines -> text-line merge -> mask refinement (bilateral + DenseCRF + dilations
    on the GPU) -> LaMa with the REFINED mask; (2) B = 1: the same chain through the plugins, page by page, host copies included."""
    import asyncio
    import warnings

    from manga_image_translator_amd import coupled, ctd as CTD, mask_refinement as MR, plugins as P, textline_merge as TM

    pages_dev, host_inputs, idx = _coupled_inputs(n_pages, distinct, device)
    dict_size = weights["ocr48"]["embd.weight"].shape[0]
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(dict_size - 4)]
    nh, nw, dw, dh = CTD.CtdEngine.letterbox_geometry(H, W)
    map_hw = (CTD.INPUT_SIZE - dh, CTD.INPUT_SIZE - dw)
    inj = _injection(host_inputs, idx, device, map_hw)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)   # log(prob) of random-weight recognitions underflows in the region statistics
        eng = coupled.CoupledPageEngine(weights, dictionary, device=device, mask_workers=mask_workers)
        kw = dict(max_seq_length=DECODE_STEPS, suppress_eos=True, prob_threshold=0.0, inject=inj, group=group or None)
        res = eng.run(pages_dev, **kw)   # warm-up
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.process_time()
        phases = {}
        for _ in range(steps):
            res = eng.run(pages_dev, **kw)
            for k, v in res.seconds.items():
                phases[k] = phases.get(k, 0.0) + v
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        cpu = (time.process_time() - c0) / steps   # CPU seconds of ALL host threads of the process (interpreter + native pools)
        n = pages_dev.shape[0]
        found = [len(t) for t in res.textlines]
        out["batch"] = dict(value=round(n / dt, 3), unit="pages/s", pages=n, steps=steps, ms_per_page=round(dt / n * 1e3, 2),
                            host_ms_per_page_by_phase={k: round(v / steps / n * 1e3, 2) for k, v in phases.items()},
                            host_cpu_ms_per_page=round(cpu / n * 1e3, 2),
                            lines_per_page_after_ocr={"min": min(found), "mean": round(float(np.mean(found)), 1), "max": max(found)},
                            text_regions_per_page=round(float(np.mean([len(r) for r in res.regions])), 1),
                            refined_mask_coverage=round(float((res.mask > 0).float().mean()), 4),
                            pipeline=dict(pages_per_slot=group or n, stage_threads=3 if group and n > group else 1, mask_workers=mask_workers),
                            note="host wall time per stage thread; the stages work on different page groups at the same time")
        # the same batch with the mask stage's side stream off (everything in the caller's stream): must give the same bytes and texts
        if eng.side_stream or eng.stage_streams:
            keep = (eng.side_stream, eng.stage_streams)
            eng.side_stream = eng.stage_streams = False
            ref = eng.run(pages_dev, **kw)
            torch.cuda.synchronize()
            if os.environ.get("MIT_BENCH_COUPLED_TWICE"):   # diagnostics: is the one-stream run itself reproducible?
                ref2 = eng.run(pages_dev, **kw)
                torch.cuda.synchronize()
                out["batch"]["pipeline"]["one_stream_twice_mask_bytes"] = int((ref2.mask != ref.mask).sum())
                res2 = None
                eng.side_stream, eng.stage_streams = keep
                res2 = eng.run(pages_dev, **kw)
                torch.cuda.synchronize()
                out["batch"]["pipeline"]["side_stream_twice_mask_bytes"] = int((res2.mask != res.mask).sum())
                del ref2, res2
            eng.side_stream, eng.stage_streams = keep
            dm = (ref.mask != res.mask).reshape(n, -1).sum(1)
            di = (ref.inpainted != res.inpainted).reshape(n, -1).sum(1)
            same_text = [[(l.text, l.prob) for l in t] for t in ref.textlines] == [[(l.text, l.prob) for l in t] for t in res.textlines]
            same = bool(dm.sum() == 0 and di.sum() == 0 and same_text)
            out["batch"]["pipeline"]["streams"] = ("one per stage thread" if keep[1] else "caller's") + (" + a high-priority side stream for the mask-refinement stage" if keep[0] else "")
            out["batch"]["pipeline"]["side_stream_results_equal_one_stream"] = same
            if not same:
                out["batch"]["pipeline"]["side_stream_diff"] = dict(pages_with_mask_diff=int((dm > 0).sum()), mask_bytes=int(dm.sum()),
                                                                    pages_with_inpainted_diff=int((di > 0).sum()), inpainted_bytes=int(di.sum()), texts_equal=same_text)
            del ref
        eng.close()
        del eng, res
        torch.cuda.empty_cache()
        if batch_only:
            return out

        # ---- B = 1 through the plugins ----
        pages, quads, _ = host_inputs
        run = asyncio.new_event_loop().run_until_complete
        det = P.HipComicTextDetector(weights=weights)
        ocr = P.HipModel48pxOCR(weights=weights["ocr48"], dictionary=dictionary)
        inp = P.HipLamaMPEInpainter(weights=weights)
        for p in (det, ocr, inp):
            run(p.load("cuda"))
        cur = {"g": 0}
        plain_forward = det.engine.forward

        def forward_with_trained_head(pages_u8, taps=None):   # the network runs in full; a trained head's maps replace its random output
            m, lines, pad = plain_forward(pages_u8, taps)
            k = idx.index(cur["g"])
            lines[:, 0] = inj["prob"][k:k + 1]
            return inj["mask"][k:k + 1], lines, pad

        det.engine.forward = forward_with_trained_head

        class Cfg:
            prob = 0.0

        per = {k: [] for k in ("detect", "ocr", "textline_merge", "mask_refinement", "inpaint")}
        n_found = []
        distinct = sorted(set(idx))[:b1_pages]
        for it, g in enumerate([distinct[0]] + distinct):   # the first page twice: the first call is the warm-up
            cur["g"] = g
            page = pages[g]
            torch.cuda.synchronize()
            t = [time.perf_counter()]
            tls, mask_raw, _ = run(det.infer(page, 1024, 0.5, 0.7, 2.3))
            t.append(time.perf_counter())
            lines = run(ocr.infer(page, tls, Cfg(), False, 0, DECODE_STEPS, True))
            lines = [l for l in lines if l.text.strip()]
            t.append(time.perf_counter())
            regions = TM.dispatch_sync(lines, W, H)
            t.append(time.perf_counter())
            mask = MR.dispatch_sync(regions, page, mask_raw, "fit_text", 20, 0, False, 3)
            t.append(time.perf_counter())
            run(inp.infer(page, mask, None, max(H, W)))
            torch.cuda.synchronize()
            t.append(time.perf_counter())
            if it:
                for k, (a, b) in zip(per, zip(t, t[1:])):
                    per[k].append(b - a)
                n_found.append(len(tls))
        # the glue stages under the kernel probe (one more call each on the last page; probe events serialise the launches, so the wall
        # clock of these passes is not the number above): where f1 (mask refinement) and a4 / a5 (boxes, refine_mask) spend GPU time
        glue = {}
        wall, kern = _probe_kernels(lambda: MR.dispatch_sync(regions, page, mask_raw, "fit_text", 20, 0, False, 3))
        glue["mask_refinement"] = dict(probed_wall_ms=wall, gpu_kernel_ms=round(sum(v["ms"] for v in kern.values()), 3), kernels=kern)
        wall, kern = _probe_kernels(lambda: run(det.infer(page, 1024, 0.5, 0.7, 2.3)))
        glue["detect (network + boxes + refine_mask)"] = dict(probed_wall_ms=wall, gpu_kernel_ms=round(sum(v["ms"] for v in kern.values()), 3), kernels=kern)
        for p in (det, ocr, inp):
            run(p.unload())
        ms = {k: round(1e3 * float(np.mean(v)), 2) for k, v in per.items()}
        tot = sum(ms.values())
        out["b1_plugins"] = dict(value=round(1e3 / tot, 3), unit="pages/s", pages=len(distinct), ms_per_page=round(tot, 2), ms_per_stage=ms,
                                 detector_boxes_found_per_page=n_found, glue_kernels=glue)
    out["order"] = "detect -> boxes -> refine_mask -> OCR (detected lines) -> textline merge -> mask refinement (dilation offset 20, kernel 3) -> inpaint (manga_translator.py:432-622)"
    out["detector_head"] = ("random-init weights fire on nothing: the maps a trained head would emit for the synthetic page (DB shrink map 0.9 inside the "
                            "shrunk text boxes, glyph mask) are put in place of the network's output after it has run in full (coupled.synthetic_head_outputs)")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# presets for the other BASELINE configurations (one GPU; secondary lines, not the headline)
# ---------------------------------------------------------------------------------------------------------------------

def _probe_pass(fn):
    """Run ``fn`` once under the C-ABI launch probe: (wall ms, executed conv FLOPs, ms at the MFMA peak of the pipes used)."""
    from manga_image_translator_amd import lib as L

    lib = L.load()
    torch.cuda.synchronize()
    L.check(lib.mit_prof_enable(1), "mit_prof_enable")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    stats = (L.MitProfStat * 64)()
    ncfg = C.c_int(0)
    L.check(lib.mit_prof_read(stats, 64, C.byref(ncfg)), "mit_prof_read")
    L.check(lib.mit_prof_enable(0), "mit_prof_enable")
    ex = peak_ms = 0.0
    tiles = {}
    for i in range(ncfg.value):
        st = stats[i]
        if not st.launches:
            continue
        name = lib.mit_conv_gemm_config_name(i).decode()
        pairs = split_pairs(name)
        ex += st.exec_flops
        peak_ms += (st.exec_flops * pairs / (BF16_MATRIX_PEAK_TFLOPS * 1e9)) if pairs else (st.exec_flops / (FP32_MATRIX_PEAK_TFLOPS * 1e9))
        tiles[name] = dict(launches=int(st.launches), ms=round(st.ms, 2), exec_tflops=round(st.exec_flops / (st.ms * 1e-3) / 1e12, 1))
    return e0.elapsed_time(e1), ex, peak_ms, tiles


def _stage_summary(ms, ex, peak_ms, tiles, per):
    return dict(ms=round(ms / per, 2), conv_exec_tflops=round(ex / (ms * 1e-3) / 1e12, 1),
                frac_of_fp32_mfma_peak=round(ex / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4), frac_of_mfma_roofline=round(peak_ms / ms, 4),
                conv_gemm_by_tile=tiles)


def config5_line(args, device):
    """BASELINE config 5 on one GPU: a 2048 x 1440 page through ESRGAN 4x (-> 8192 x 5760, the tensor an --upscale-ratio 2 run resizes
    to 4096 x 2880; upscaling/esrgan_pytorch.py:537-549) and a 2048 x 1456 page through lama_large (18 FFC blocks, no MPE;
    inpainting_lama_mpe.py:121-136).  One step = ``--pages`` pages through both; prints its own line."""
    from manga_image_translator_amd import esrgan, esrgan_schema, lama, lama_schema, ops as _ops, synth

    n = max(1, min(args.pages, 4))
    He, We = 2048, 1440
    eng = esrgan.EsrganEngine(synth.synth_state_dict(esrgan_schema.rrdbnet_schema(23)), nb=23, device=device)
    pages_e = torch.from_numpy(np.stack([synth.synth_page(i, He, We, n_boxes=16)[0] for i in range(n)])).to(device)
    leng = lama.LamaEngine(synth.synth_state_dict(lama_schema.lama_generator_schema(18)), None, n_blocks=18, device=device)
    gen = [synth.synth_page(i, H, W) for i in range(n)]
    img = torch.from_numpy(np.stack([g[0] for g in gen])).to(device)
    msk = torch.from_numpy(np.stack([g[2] for g in gen])).to(device)

    def up():
        for i in range(n):   # one page per call: the 4x output of a single page is 8192 x 5760 x 64 floats in its widest layer
            eng.forward(pages_e[i:i + 1])

    def inpaint():
        leng.forward(img, msk)

    def step():
        up()
        inpaint()

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    stages = {"esrgan_4x": _stage_summary(*_probe_pass(up), per=n), "lama_large": _stage_summary(*_probe_pass(inpaint), per=n)}
    stages["esrgan_4x"]["alg_tflops"] = round(eng.flops_per_input_pixel() * He * We / (stages["esrgan_4x"]["ms"] * 1e-3) / 1e12, 1)
    stages["lama_large"]["alg_tflops"] = round(leng.flops_per_page(H, W) / (stages["lama_large"]["ms"] * 1e-3) / 1e12, 1)
    return {"metric": "pages/sec, ESRGAN 4x upscale of a 2048x1440 page + lama_large inpainting at 2048x1456 (BASELINE config 5, one GPU)",
            "value": round(n / dt, 4), "unit": "pages/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 5: {n} pages per step, RRDBNet (nb = 23) 4x on 2048x1440 + lama_large (18 FFC blocks) on 2048x1456; "
                                   "random-init weights of the reference architectures", "pages_per_gpu": n},
            "gemm_mode": {"mode": _ops.split_mode()}, "stages_ms_per_page": stages,
            "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def config1_line(args, device):
    """BASELINE config 1 on one GPU: one 1024 x 1024 page at a time, --detector default (DBNet-R34 at detect_size 2048: the page is
    upsampled 2x, detection/default.py:56-103) + --ocr 48px + --inpainter lama_mpe."""
    from manga_image_translator_amd import dbnet, dbnet_schema, lama, lama_schema, ocr48, ocr_schema, ops as _ops, pipeline, synth

    Hc = Wc = 1024
    n = max(1, min(args.pages, 8))
    det = dbnet.DbnetEngine(synth.synth_state_dict(dbnet_schema.text_detection_schema()), device=device)
    D = pipeline.DICT_SIZE
    ocr = ocr48.Ocr48Engine(synth.synth_state_dict(ocr_schema.ocr48_schema(D)), D, device=device)
    leng = lama.LamaEngine(synth.synth_state_dict(lama_schema.lama_generator_schema(9)), synth.synth_state_dict(lama_schema.lama_mpe_schema()),
                           n_blocks=9, device=device)
    gen = [synth.synth_page(i, Hc, Wc, n_boxes=16) for i in range(n)]
    pages = [torch.from_numpy(g[0][None]).to(device) for g in gen]
    masks = [torch.from_numpy(g[2][None]).to(device) for g in gen]
    quads = [pipeline.quads_from_array(g[1]) for g in gen]
    from manga_image_translator_amd import imgproc

    def detect():
        for p in pages:   # resize_aspect_ratio to detect_size 2048 (default.py:62, imgproc.py:37-70): a 1024^2 page is upsampled 2x
            det.forward(imgproc.resize_u8(p, (2048, 2048)))

    def recognise():
        for p, q in zip(pages, quads):
            ocr.recognize_pages(p, [q], max_seq_length=DECODE_STEPS, suppress_eos=True)

    def inpaint():
        for p, m in zip(pages, masks):
            leng.forward(p, m)

    def step():
        detect()
        recognise()
        inpaint()

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    stages = {"default_dbnet": _stage_summary(*_probe_pass(detect), per=n), "ocr48": _stage_summary(*_probe_pass(recognise), per=n),
              "lama_mpe": _stage_summary(*_probe_pass(inpaint), per=n)}
    return {"metric": "pages/sec, one 1024x1024 page at a time: detector=default + ocr=48px + inpainter=lama_mpe (BASELINE config 1, one GPU)",
            "value": round(n / dt, 3), "unit": "pages/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 1: {n} synthetic 1024x1024 pages per step, one page per call (B = 1), DBNet-R34 at detect_size 2048 + "
                                   f"48px OCR (16 lines/page, {DECODE_STEPS} decode steps) + lama_mpe; random-init weights", "pages_per_gpu": n},
            "gemm_mode": {"mode": _ops.split_mode()}, "stages_ms_per_page": stages}


HEADLINE_MAX_BYTES = 4096   # the driver keeps an 8 KB tail of stdout: the line it parses must fit with room to spare


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(out: dict, details_path: str = "") -> dict:
    """The ONE stdout line: the contract keys, ``roofline`` and ``cpu_baseline`` reduced to what the contract names, and one number per
    sub-measurement.  Everything else the run measured stays in the full record (``bench_details.json`` + stderr)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data"))
    cfg = dict(out.get("config") or {})
    cfg.pop("microbatch", None)
    line["config"] = cfg
    roof = out.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "kernel", "unit", "achieved", "peak", "frac", "plane_pairs", "avg_launch_us", "launches",
                         "alg_gflop_per_launch", "fp32_equivalent_tflops"))
        t = roof.get("traffic")
        r["traffic"] = None if not t else {**_pick(t, ("read_GB", "write_GB", "ratio_to_algorithmic", "source")), "per": "launch"}
        r["stages"] = {k: _pick(v, ("ms_per_page", "frac_of_mfma_roofline")) | ({"mfma_busy": v["mfma_busy"].get("frac")} if v.get("mfma_busy") else {})
                       for k, v in (roof.get("stages") or {}).items()}
        if roof.get("whole_step"):
            r["whole_step_frac_of_mfma_roofline"] = roof["whole_step"].get("frac_of_mfma_roofline")
        line["roofline"] = r
    else:
        line["roofline"] = None
    cpu = out.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "seconds_per_stage")) if cpu else None
    if "speedup_vs_cpu_baseline" in out:
        line["speedup_vs_cpu_baseline"] = out["speedup_vs_cpu_baseline"]
    par = out.get("parity_checked")
    line["parity_checked"] = _pick(par, ("ok", "pages")) if par else None
    if out.get("fp32_mfma"):
        f = out["fp32_mfma"]
        line["fp32_mfma"] = {**_pick(f, ("value", "unit", "ms_per_step")),
                             **({"roofline_frac": f["roofline"].get("frac")} if f.get("roofline") else {})}
    line["gemm_mode"] = _pick(out.get("gemm_mode") or {}, ("mode",))
    if out.get("two_streams"):
        line["two_streams"] = _pick(out["two_streams"], ("value", "results_equal_one_stream"))
    if out.get("dropin"):
        line["dropin"] = _pick(out["dropin"], ("value", "unit", "ms_per_page", "ms_per_stage"))
    if out.get("coupled"):
        line["coupled"] = {k: _pick(v, ("value", "unit", "ms_per_page")) for k, v in out["coupled"].items()
                           if isinstance(v, dict) and "value" in v}
    if out.get("other_configs"):
        line["other_configs"] = {k: _pick(v, ("value", "unit", "pages_per_step")) for k, v in out["other_configs"].items() if v}
    if out.get("gather"):
        line["gather"] = _pick(out["gather"], ("verified_blocks", "overlap_with_compute", "bytes_per_step", "wait_ms_rank0_per_step"))
    if out.get("leg_errors"):
        line["leg_errors"] = {k: str(v)[:120] for k, v in out["leg_errors"].items()}
    if details_path:
        line["details"] = details_path
    if len(json.dumps(line)) > HEADLINE_MAX_BYTES:   # never lose the contract keys to a long tail: drop the optional ones, last first
        for k in ("other_configs", "coupled", "dropin", "two_streams", "fp32_mfma", "gather", "leg_errors"):
            line.pop(k, None)
            if len(json.dumps(line)) <= HEADLINE_MAX_BYTES:
                break
    return line


def emit(out: dict, details_path: str) -> None:
    """Full record -> ``details_path`` and stderr; compact headline -> the LAST line of stdout."""
    full = json.dumps(out)
    wrote = ""
    if details_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(details_path)), exist_ok=True)
            with open(details_path, "w") as f:
                f.write(full + "\n")
            wrote = details_path
        except OSError as ex:
            print(f"bench.py: could not write {details_path}: {ex}", file=sys.stderr)
    print("bench.py full record: " + full, file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    print(json.dumps(compact_line(out, wrote)), flush=True)


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """``python bench.py --gpus N`` started OUTSIDE torch.distributed.run (no WORLD_SIZE in the environment): start the N ranks ourselves,
    exactly the way the contract's launcher does — ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>`` — and pass their exit code on.  Rank 0 of that job prints the one JSON line on the stdout
    we share with it."""
    import subprocess

    if not args.launch_rehearsal and os.environ.get("MIT_DIST_BACKEND") != "gloo":
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:   # one rank per GPU over RCCL: refuse early, with the reason, instead of a rendezvous that hangs
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node shows {n} "
                  "(MIT_DIST_BACKEND=gloo rehearses the N-rank path on fewer GPUs)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def launch_rehearsal(args) -> None:
    """The N-rank plumbing of the headline run and nothing else, on gloo with CPU tensors: rendezvous, the weight arena broadcast from
    rank 0, ``--steps`` verified gathers of a synthetic per-rank result block, barrier + max-over-ranks timing, one line on rank 0.
    No kernel runs and the line says so (``rehearsal: true``, ``value`` null): it exists so that the launch path the driver's 8-GPU
    command takes can be exercised where there is no GPU (tests/test_bench_launch.py)."""
    from manga_image_translator_amd import dist as D

    os.environ["MIT_DIST_BACKEND"] = "gloo"
    rank, world, _ = D.init(backend="gloo")
    g = torch.Generator().manual_seed(7)
    weights = {"demo": {"w": torch.randn(257, 33, generator=g), "b": torch.arange(5, dtype=torch.int64)}} if rank == 0 else None
    weights = D.broadcast_weights(weights)
    wsum = float(weights["demo"]["w"].double().sum()) + float(weights["demo"]["b"].sum())
    gather = D.PageGather()
    D.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        block = ((torch.arange(args.pages * 1001, dtype=torch.int64) * (rank + 3) + k) % 251).to(torch.uint8)
        gather.submit(block)
    got = gather.wait()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    sums = [None] * world
    torch.distributed.all_gather_object(sums, wsum) if world > 1 else sums.__setitem__(0, wsum)
    if rank == 0:
        k = args.steps - 1
        blocks_ok = all(torch.equal(got[r], ((torch.arange(args.pages * 1001, dtype=torch.int64) * (r + 3) + k) % 251).to(torch.uint8))
                        for r in range(world))
        print(json.dumps({"metric": "pages/sec end-to-end (detect+OCR+inpaint), 2048x1456", "rehearsal": True, "value": None, "unit": "pages/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3),
                          "backend": "gloo", "weights_equal_on_all_ranks": len(set(sums)) == 1, "last_blocks_equal_what_ranks_sent": bool(blocks_ok),
                          "gather": {"verified_blocks": gather.check(), "bytes_per_step": gather.last_bytes, "wait_ms_rank0": round(gather.wait_ms, 3)},
                          "note": "launch / rendezvous / broadcast / gather plumbing only; no kernel ran, nothing here is a measurement"}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.launch_rehearsal:
        launch_rehearsal(args)
        return
    from manga_image_translator_amd import dist as D

    rank, world, local = D.init()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
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
    if args.config5 or args.config1:   # one-GPU presets of the other BASELINE configurations: their own line
        if rank == 0:
            print(json.dumps((config5_line if args.config5 else config1_line)(args, device)))
        D.barrier()
        return
    weights = pipeline.synthetic_weights() if rank == 0 else None
    weights = D.broadcast_weights(weights)          # RCCL broadcast of one flat arena at load
    pages, quads, masks, host_inputs, idx = make_inputs(args.pages, args.distinct, rank, device)

    if args.coupled_only:   # A/B runs of the coupled batch path's knobs: its own line
        if rank == 0:
            c = coupled_leg(weights, args.pages, args.distinct, device, steps=max(args.steps, 1), group=args.coupled_group,
                            mask_workers=args.coupled_mask_workers, batch_only=True)
            print(json.dumps({"metric": "pages/sec, coupled batch path (detect -> boxes -> OCR -> merge -> mask refinement -> inpaint), 2048x1456",
                              **c["batch"], "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic"}))
        return
    if args.mode == "dropin":
        if rank == 0:
            d = dropin_leg(weights, host_inputs, args.dropin_pages)
            print(json.dumps({"metric": "pages/sec through the drop-in plugins, one page at a time (B=1), 2048x1456", **d,
                              "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"ctd_hip + 48px_hip ({N_BOXES} lines/page, {DECODE_STEPS} decode steps) + lama_mpe_hip, "
                                                     f"{H}x{W} synthetic pages, one page per call"}}))
        return

    engine = pipeline.PageEngine(weights, device=device, ctd_mb=args.ctd_mb, lama_mb=args.lama_mb, group=args.group,
                                 overlap=args.overlap)
    gathered = {"bytes": 0}
    gather = D.PageGather()

    def step():
        res = engine.run(pages, quads, masks, max_seq_length=DECODE_STEPS, suppress_eos=True, stages=stages)
        if world > 1:  # per-page result records to rank 0 (point-to-point over xGMI), asynchronously: the next step's kernels do not
            gather.submit(res.packed_pages(N_BOXES) if "ocr" in stages else res.packed())  # wait for peer traffic; any failure ends the run
            gathered["bytes"] = max(gathered["bytes"], gather.last_bytes)
        return res

    def drain():
        if world > 1:
            gather.wait()  # the last step's gather belongs to the timed region

    res = None
    for _ in range(max(args.warmup, 1) if world > 1 else args.warmup):  # N > 1: at least one untimed step proves the gather works
        res = step()
    drain()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    drain()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0)

    roof = per_cfg = cpu = parity = dropin = fp32 = None
    leg_errors = {}
    gather_info = None
    if world > 1:   # every block rank 0 received in the timed region against the checksum its source rank computed before sending it
        try:
            gather_info = {"verified_blocks": gather.check(), "overlap_with_compute": bool(gather.async_op),
                           "bytes_per_step": gathered["bytes"], "wait_ms_rank0_per_step": round(gather.wait_ms / max(gather.submits, 1), 3),
                           "wait_note": "rank 0's time inside the gather (receive of world - 1 blocks + their checksums) per submit over the whole "
                                        "run, from events on the compute stream (nccl) or the host clock (gloo): what max-over-ranks timing "
                                        "charges to the step because of the gather",
                           "note": "each rank sends page_checksum(block) beside its block; rank 0 recomputes it on what arrived"}
        except Exception as ex:  # a corrupted gather must not lose the line, and must not go unnoticed
            leg_errors["gather_checksum"] = f"{type(ex).__name__}: {ex}"
    from manga_image_translator_amd import ops as _ops

    shipped_mode = _ops.split_mode()
    if shipped_mode and not args.no_fp32_leg:
        # the same engine (same packed weights, same batch) with the GEMM mode switched to the fp32 MFMA: timed in the same process,
        # same barrier discipline, max over ranks (VERDICT r02 #1b)
        _ops.set_split_mode(0)
        for _ in range(1):
            step()
        drain()
        D.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.fp32_steps):
            step()
        drain()
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        dt32 = D.max_over_ranks(time.perf_counter() - t1)
        fp32 = dict(value=round(args.pages * world * args.fp32_steps / dt32, 3), unit="pages/s", steps=args.fp32_steps, warmup=1,
                    ms_per_step=round(dt32 / args.fp32_steps * 1e3, 2), gemm_mode=0,
                    note="the same engine and batch with mit_gemm_mode_set(0): every contraction on v_mfma_f32_32x32x2_f32")
        _ops.set_split_mode(shipped_mode)

    two = None
    if not args.overlap and not args.no_two_streams and set(stages) == {"detect", "ocr", "inpaint"}:
        # the same engine with LaMa on the caller's stream and detector + OCR on a second one (PageEngine(overlap=True)): timed the same
        # way, and its results compared byte for byte with the one-stream step's.  A sub-measurement: the headline and its roofline
        # stay on one stream, where a kernel's duration is its own.
        engine.overlap = True
        r2 = step()
        drain()
        D.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.fp32_steps):
            r2 = step()
        drain()
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        dt2 = D.max_over_ranks(time.perf_counter() - t2)
        engine.overlap = False
        same = bool(torch.equal(r2.inpainted, res.inpainted) and torch.equal(r2.det_mask, res.det_mask) and torch.equal(r2.det_shrink, res.det_shrink)
                    and torch.equal(r2.ocr_tokens, res.ocr_tokens) and torch.equal(r2.ocr_prob, res.ocr_prob) and torch.equal(r2.ocr_colors, res.ocr_colors))
        two = dict(value=round(args.pages * world * args.fp32_steps / dt2, 3), unit="pages/s", steps=args.fp32_steps, warmup=1,
                   ms_per_step=round(dt2 / args.fp32_steps * 1e3, 2), results_equal_one_stream=same,
                   note="PageEngine(overlap=True): LaMa on the caller's stream, detector + OCR on a second stream that joins at the end of the step")
        del r2

    cfg2 = None
    if world == 1 and not args.no_other_configs and "detect" in stages:
        # BASELINE config 2 (the same 64-page batch, detector = ctd only) on the headline's engine: 1 warm-up + 2 timed steps
        def run_ctd():
            return engine.run(pages, quads, masks, max_seq_length=DECODE_STEPS, suppress_eos=True, stages=("detect",))
        run_ctd()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(2):
            run_ctd()
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - tc) / 2
        cfg2 = dict(metric="pages/sec, detector=ctd only (BASELINE config 2), 2048x1456", value=round(args.pages / dtc, 2), unit="pages/s",
                    pages=args.pages, steps=2, warmup=1, ms_per_step=round(dtc * 1e3, 2))

    def leg(name, fn):
        """The legs run after the timed region; a failing leg is reported in the line (``leg_errors``), it does not lose the headline."""
        try:
            return fn()
        except Exception as ex:  # noqa: BLE001 - reported, not swallowed
            import traceback

            leg_errors[name] = f"{type(ex).__name__}: {ex}"
            traceback.print_exc(file=sys.stderr)
            return None

    if not args.no_roofline and rank == 0:
        r = leg("roofline", lambda: roofline_leg(engine, pages, quads, masks, stages, args.prof_dump))
        roof, per_cfg = r if r is not None else (None, None)
        if fp32 is not None:
            def fp32_roof():
                with _ops.gemm_mode(0):
                    return roofline_leg(engine, pages, quads, masks, stages, args.prof_dump + ".fp32" if args.prof_dump else "")
            r = leg("fp32_mfma.roofline", fp32_roof)
            if r is not None:
                fr, fcfg = r
                for k in ("hbm_kernels", "hbm_peak_GBps", "pmc_source"):
                    fr.pop(k, None)
                fp32["roofline"], fp32["conv_gemm_by_tile"] = fr, fcfg
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        r = leg("cpu_baseline", lambda: cpu_baseline_leg(weights, host_inputs, stages, args.cpu_pages,
                                                         [int(t) for t in args.cpu_threads.split(",") if t]))
        if r is not None:
            cpu, oracle_out = r
            parity = leg("parity_checked", lambda: parity_leg(res, idx, oracle_out, stages))
    coupled = None
    if not args.no_dropin and rank == 0 and world == 1 and set(stages) == {"detect", "ocr", "inpaint"}:
        del engine
        torch.cuda.empty_cache()
        dropin = leg("dropin", lambda: dropin_leg(weights, host_inputs, args.dropin_pages))
        if not args.no_coupled:
            del pages, masks
            torch.cuda.empty_cache()
            coupled = leg("coupled", lambda: coupled_leg(weights, args.pages, args.distinct, device, group=args.coupled_group,
                                                         mask_workers=args.coupled_mask_workers))
    other = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        # The other one-GPU BASELINE configurations, each a SHORT run of its own preset (python bench.py --config1 / --config5 print the
        # full lines): so that the driver's default command leaves a number for every configuration that fits one GPU.
        import copy

        other = {"config2_ctd_only": cfg2}
        for key, fn, npg in (("config1_default_48px_lama_1024", config1_line, 4), ("config5_esrgan4x_lama_large", config5_line, 1)):
            a2 = copy.copy(args)
            a2.pages, a2.steps, a2.warmup = npg, 1, 1
            engine = pages = masks = None   # noqa: F841 - the headline's buffers are not needed any more
            torch.cuda.empty_cache()
            r = leg(key, lambda fn=fn, a2=a2: fn(a2, device))
            if r is not None:
                other[key] = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step") if k in r}
                other[key]["stage_ms_per_page"] = {k: v.get("ms") for k, v in r.get("stages_ms_per_page", {}).items()}
                other[key]["pages_per_step"] = npg
    D.barrier()

    if rank == 0:
        total_pages = args.pages * world * args.steps
        value = total_pages / dt
        cfg_name = "BASELINE config 4" if args.pages == 128 else "BASELINE config 3"
        out = {
            "metric": "pages/sec end-to-end (detect+OCR+inpaint), 2048x1456", "value": round(value, 3), "unit": "pages/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {args.pages} synthetic {H}x{W} pages per GPU, detector=ctd + ocr=48px "
                                   f"({N_BOXES} lines/page, {DECODE_STEPS} decode steps, EOS suppressed) + inpainter=lama_mpe; "
                                   "random-init weights of the reference architectures",
                       "pages_per_gpu": args.pages, "distinct_pages": len(host_inputs[0]), "stages": list(stages),
                       "microbatch": {"ctd": args.ctd_mb, "lama": args.lama_mb, "ocr_group": args.group},
                       "streams": 2 if args.overlap else 1,
                       "parallelism": f"pages sharded one contiguous block per GPU x{world}; RCCL weight broadcast"
                                      + (f" + per-step gather of {gathered['bytes']} result bytes to rank 0" if world > 1 else "")},
            "roofline": roof, "fp32_mfma": fp32, "two_streams": two, "cpu_baseline": cpu, "parity_checked": parity, "dropin": dropin, "coupled": coupled,
            "other_configs": other, "conv_gemm_by_tile": per_cfg,
        }
        if shipped_mode:  # say so wherever the number travels
            n = shipped_mode
            dropped = ("no plane product dropped: only the fp32 accumulation order differs from the fp32 MFMA" if n == 9 else
                       "the three products with p + q >= 3 dropped, each <= 2^-25 |a b|: below one fp32 multiply rounding")
            out["dtype"] = "f32"
            out["gemm_mode"] = {"mode": n, "arithmetic": f"fp32 operands as three exact bf16 planes (x = hi + mid + lo), {n} of the 9 plane products on "
                                f"v_mfma_f32_32x32x16_bf16, every product exact in fp32, fp32 accumulation; {dropped}",
                                "applies_to": "contractions with constant weights that fill the chip (conv_gemm_split_kernel); all other launches and "
                                              "every non-GEMM kernel compute in fp32 as in mode 0",
                                "headline_and_parity_checked_in_this_mode": True, "fp32_mfma_beside_it": "fp32_mfma",
                                "roofline_priced_on": "the bf16 MFMA pipe (2500 TFLOP/s dense) for the split tiles: executed = pairs x algorithmic FLOPs",
                                "switch": "MIT_GEMM_SPLIT=0|6|9 or mit_gemm_mode_set()"}
        else:
            out["gemm_mode"] = {"mode": 0, "arithmetic": "fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere"}
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        if gather_info:
            out["gather"] = gather_info
        if leg_errors:
            out["leg_errors"] = leg_errors
        emit(out, args.details)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
