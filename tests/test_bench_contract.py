"""The bench line contract (driver-facing): checked on the committed evidence lines, and bench.py's CLI must parse without a GPU."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01[a-z]_bench.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_lines_follow_the_contract(path):
    d = json.load(open(path))
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict)):
        assert isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["unit"] == "pages/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    pages = d["config"]["pages_per_gpu"] * d["n_gpus"] * d["steps"]
    assert abs(d["value"] - pages / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)


def test_bench_cli_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
