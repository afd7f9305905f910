"""`python bench.py --gpus N` must start its own N ranks (the driver's command shape carries no torch.distributed.run): the launch path,
rendezvous on 127.0.0.1, weight-arena broadcast, checksummed result gather and max-over-ranks timing rehearsed on gloo without a GPU
(the partitioning the ranks follow is the reference's page loop, manga_translator.py:1491-1519)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):   # as the driver starts it: no launcher environment
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=e)


def test_bench_gpus_2_launches_its_own_ranks_and_prints_one_line():
    out = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--pages", "4", "--launch-rehearsal")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout           # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["rehearsal"] is True and d["value"] is None   # a rehearsal never carries a number
    assert d["gather"]["verified_blocks"] == 2 * 3 and d["gather"]["bytes_per_step"] == 2 * 4 * 1001
    assert d["weights_equal_on_all_ranks"] and d["last_blocks_equal_what_ranks_sent"]
    assert d["gather"]["wait_ms_rank0"] >= 0


def test_bench_gpus_n_refuses_early_without_enough_gpus():
    """No GPU here: the real (non-rehearsal) job must say why it cannot start and exit 2 at once, not hang in a rendezvous."""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    out = _run("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert out.returncode == 2
    assert "visible GPUs" in out.stderr and not out.stdout.strip()
