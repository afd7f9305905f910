"""Multi-GPU path on one GPU box: what can be checked without a second GPU.

* two ranks (gloo rendezvous, both on cuda:0, running CONCURRENTLY) run the real PageEngine on their contiguous page blocks and rank 0 gathers the
  per-page records: byte-identical to one process running all pages — a page's result does not depend on the world size, the
  shard it landed in, or its neighbours in a batch (the reference processes pages independently, manga_translator.py:1491-1519);
* a world-size-1 RCCL group (backend "nccl") runs the collectives dist.py uses — uint8 arena broadcast, gather to rank 0,
  MAX all-reduce, barrier — on device tensors (inside rank 0's process of the same spawn: a fresh interpreter costs minutes of
  imports on a cold box)."""
import contextlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

H, W, LINES, T, D = 256, 192, 3, 4, 96
N_PAGES = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _records(weights, lo, hi):
    from manga_image_translator_amd import pipeline, synth

    dev = torch.device("cuda:0")
    eng = pipeline.PageEngine(weights, device=dev, dict_size=D)
    pages, quads, masks = zip(*[synth.synth_page(i, H, W, n_boxes=LINES) for i in range(lo, hi)])
    res = eng.run(torch.from_numpy(np.stack(pages)).to(dev), [pipeline.quads_from_array(q) for q in quads],
                  torch.from_numpy(np.stack(masks)).to(dev), max_seq_length=T, suppress_eos=True)
    torch.cuda.synchronize()
    return res.packed_pages(LINES)


def _shard_worker(rank, world, port, q, gpu_turn):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MIT_DIST_BACKEND="gloo")
    try:
        from manga_image_translator_amd import dist as Dm, pipeline

        Dm.init()
        weights = Dm.broadcast_weights(pipeline.synthetic_weights(dict_size=D) if rank == 0 else None)
        lo, hi = Dm.shard_range(N_PAGES, rank, world)
        # Both ranks of this rehearsal sit on ONE GPU (a real job has one GPU per rank) and run their engines AT THE SAME TIME: kernels of
        # the two processes share CUs.  Round 3 had to serialise them (the 128 x 128 split GEMM tile of one process disturbed the FFT rows
        # kernels of the other: packed-fp32 instructions of the SLP vectoriser); built without them the concurrent run — default
        # launches, no safe mode — is byte-identical to the single process.  MIT_TEST_SERIALISE_RANKS=1 restores the turn-taking for diagnosis.
        with (gpu_turn if os.environ.get("MIT_TEST_SERIALISE_RANKS") else contextlib.nullcontext()):
            recs = _records(weights, lo, hi)
            torch.cuda.synchronize()
        out = Dm.gather_pages(recs)
        if rank == 0:
            q.put(("records", out.reshape(N_PAGES, -1).cpu().numpy()))
        Dm.barrier()
        torch.distributed.destroy_process_group()
        if rank == 0:  # same process, second group: the RCCL smoke (a spawned interpreter costs minutes of imports on a cold box)
            q.put(("rccl", _rccl_smoke(_free_port())))
        q.put((rank, "ok"))
    except Exception as e:
        q.put((rank, repr(e)))
        raise


def test_page_records_do_not_depend_on_world_size_and_rccl_smoke(cuda):
    from manga_image_translator_amd import pipeline

    single = _records(pipeline.synthetic_weights(dict_size=D), 0, N_PAGES).cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    gpu_turn = ctx.Lock()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q, gpu_turn)) for r in range(2)]
    for p in procs:
        p.start()
    got, rccl, oks = None, None, []
    for _ in range(4):
        item = q.get(timeout=900)
        if item[0] == "records":
            got = item[1]
        elif item[0] == "rccl":
            rccl = item[1]
        else:
            oks.append(item)
    for p in procs:
        p.join(timeout=120)
    assert sorted(oks) == [(0, "ok"), (1, "ok")], oks
    assert got.shape == single.shape and got.dtype == np.uint8
    assert np.array_equal(got, single), f"{(got != single).sum()} bytes differ between world 1 and world 2"
    assert rccl == "ok", rccl  # world-size-1 RCCL group: uint8 broadcast, gather to rank 0, MAX all-reduce, barrier


def _rccl_smoke(port):
    """A world-size-1 RCCL group (backend "nccl") running the collectives dist.py uses, on device tensors."""
    import torch.distributed as dist

    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        arena = torch.arange(1 << 20, dtype=torch.int64, device=dev).to(torch.uint8)
        ref = arena.clone()
        dist.broadcast(arena, src=0)                                     # the weight arena (dist.broadcast_weights)
        packed = torch.arange(4096, dtype=torch.int32, device=dev).view(torch.uint8)
        out = torch.empty((1,) + tuple(packed.shape), dtype=torch.uint8, device=dev)
        dist.gather(packed, list(out.unbind(0)), dst=0)                  # the per-page records (dist.gather_pages)
        t = torch.tensor([3.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                          # max-over-ranks timing (dist.max_over_ranks)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        return "ok" if torch.equal(arena, ref) and torch.equal(out[0], packed) and float(t.item()) == 3.5 else "wrong results"
    finally:
        dist.destroy_process_group()


def test_bench_gpus_2_self_launch_two_ranks_share_the_gpu(cuda):
    """The driver's command shape, `python bench.py --gpus 2 …` with no launcher around it: bench.py starts its own two ranks
    (torch.distributed.run on 127.0.0.1), here on the gloo backend because both ranks sit on the box's one GPU.  The REAL engine runs on
    each rank's page block; the line must carry n_gpus = 2 and a gather whose blocks all passed their checksums."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MIT_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pages", "2",
                          "--no-cpu-baseline", "--no-roofline", "--no-dropin", "--no-fp32-leg", "--no-two-streams"],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 2 * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]       # pages of ALL ranks over the max-over-ranks time
    assert d["gather"]["verified_blocks"] == 2 * 3 and d["gather"]["bytes_per_step"] > 0 and "leg_errors" not in d
