"""The bench line contract (driver-facing): checked on the committed evidence lines, and bench.py's CLI must parse without a GPU."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9][a-z]_bench.json")))


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
    if os.path.basename(path) >= "r02a":  # round 2 lines: per-stage roofline, HBM kernel table, parity and drop-in legs
        assert set(r["stages"]) == {"ctd", "ocr48", "lama_mpe"} and all(0 < v["frac_of_fp32_mfma_peak"] < 1 for v in r["stages"].values())
        if os.path.basename(path) < "r03a":
            assert 0 < r["whole_step"]["frac_of_fp32_mfma_peak"] <= r["frac"]
        big = [v for v in r["hbm_kernels"].values() if v.get("alg_GB_per_launch", 0) > 0.5]
        assert big and all(0 < v["frac_of_hbm_peak"] < 1 for v in big)
        assert d["parity_checked"]["ok"] is True and d["parity_checked"]["ocr"]["lines_with_different_tokens"] == 0
        assert d["dropin"]["batch"] == 1 and d["dropin"]["unit"] == "pages/s" and d["dropin"]["value"] < d["value"]
        assert "thread" in c["sample"] and c["cores"] in (8, 16, 32, 64, 128)


    if os.path.basename(path) >= "r03a":  # round 3 lines: the GEMM mode is named, the fp32-MFMA figure is timed beside a split-mode headline,
        gm = d["gemm_mode"]               # split tiles are priced on the bf16 pipe, the coupled (glue-inclusive) path has its own numbers
        assert gm["mode"] in (0, 6, 9)
        if gm["mode"]:
            f = d["fp32_mfma"]
            assert f["unit"] == "pages/s" and 0 < f["value"] < 1.1 * d["value"] and f["gemm_mode"] == 0
            assert 0 < f["roofline"]["frac"] < 1 and f["roofline"]["peak"] == 157.3
            assert r["peak"] == 2500.0 and r["plane_pairs"] == gm["mode"] and r["kernel"].startswith("conv_gemm_split_kernel")
            assert abs(r["achieved"] - r["plane_pairs"] * r["fp32_equivalent_tflops"]) < 0.6
        assert all(0 < v["frac_of_mfma_roofline"] < 1 for v in r["stages"].values()) and 0 < r["whole_step"]["frac_of_mfma_roofline"] < 1
        c2 = d["coupled"]
        assert c2["batch"]["unit"] == "pages/s" and 0 < c2["batch"]["value"] < d["value"]
        assert c2["b1_plugins"]["value"] > 0 and min(c2["b1_plugins"]["detector_boxes_found_per_page"]) >= 28
        assert c2["batch"]["lines_per_page_after_ocr"]["min"] >= 24


@pytest.mark.parametrize("path", LINES[-6:], ids=[os.path.basename(p) for p in LINES[-6:]])
def test_headline_of_a_full_record_fits_the_drivers_tail(path):
    """The driver keeps an 8 KB tail of stdout and parses its last line (BENCH_r05.parsed was null: a 20 KB line).  bench.py prints a
    compact headline; here every recent committed full record is reduced the same way and must fit, parse from a tail, and keep
    ``roofline`` and ``cpu_baseline`` with the contract's fields."""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(path))
    line = json.dumps(bench.compact_line(full, "bench_details.json"))
    assert len(line) < bench.HEADLINE_MAX_BYTES == 4096
    stdout = "warning: something a library printed earlier\n" * 400 + line + "\n"
    d = json.loads(stdout[-8000:].strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and "model" not in d["config"]
    r, c = d["roofline"], d["cpu_baseline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c)


def test_emit_prints_the_headline_last_and_keeps_the_full_record(tmp_path, capsys):
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(LINES[-1]))
    bench.emit(full, str(tmp_path / "sub" / "details.json"))
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    assert len(cap.out) < 4200 and json.loads(last)["value"] == full["value"]
    assert json.load(open(tmp_path / "sub" / "details.json")) == full          # nothing measured is lost
    assert "conv_gemm_by_tile" in cap.err


def test_gpus_8_without_pages_is_baseline_config_4():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.parse(["--gpus", "8"]).pages == 128 and bench.parse(["--gpus", "1"]).pages == 64
    assert bench.parse(["--gpus", "8", "--pages", "64"]).pages == 64 and bench.parse(["--gpus", "4"]).pages == 64


def test_bench_cli_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_cpu_baseline_and_parity_legs_on_a_small_page(monkeypatch):
    """bench.py's CPU legs end to end without a GPU: thread sweep + warm-up + timed pages through the oracle, and the parity leg
    fed with a stand-in for the GPU results built from the oracle outputs themselves (must report ok) and with a corrupted
    copy (must not)."""
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from manga_image_translator_amd import pipeline, synth

    monkeypatch.setattr(bench, "H", 256)
    monkeypatch.setattr(bench, "W", 192)
    monkeypatch.setattr(bench, "N_BOXES", 4)
    monkeypatch.setattr(bench, "DECODE_STEPS", 4)
    weights = pipeline.synthetic_weights(dict_size=64)
    pages, quads, masks = zip(*[synth.synth_page(i, 256, 192, n_boxes=4) for i in range(2)])
    cpu, outs = bench.cpu_baseline_leg(weights, (pages, quads, masks), ("detect", "ocr", "inpaint"), 1, [1, 2])
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] in (1, 2) and len(outs) == 2
    assert set(cpu["seconds_per_stage"]) == {"detect", "ocr", "inpaint"} and set(cpu["thread_sweep_seconds"]["ocr"]) <= {"1", "2"}
    assert abs(cpu["value"] - 1.0 / sum(cpu["seconds_per_stage"].values())) < 1e-3 * cpu["value"]

    class Res:
        pass

    T = 4
    res = Res()
    res.det_mask = torch.from_numpy(np.stack([o["detect"][0] for o in outs]))
    res.det_shrink = torch.from_numpy(np.stack([(o["detect"][1][0, 0] > 0.3).astype(np.uint8) for o in outs]))
    res.inpainted = torch.from_numpy(np.stack([o["inpaint"] for o in outs]))
    order, toks, lens, probs = [], [], [], []
    for b, o in enumerate(outs):
        for indices, r in o["ocr"]:
            for j, i in enumerate(indices):
                order.append((b, i))
                t = r[j][0].numpy()
                toks.append(np.concatenate([[1], t, np.zeros(T - len(t), np.int64)]))
                lens.append(1 + len(t))
                probs.append(r[j][1])
    res.ocr_order, res.ocr_tokens = order, torch.tensor(np.stack(toks), dtype=torch.int32)
    res.ocr_length, res.ocr_prob = torch.tensor(lens, dtype=torch.int32), torch.tensor(probs, dtype=torch.float32)
    par = bench.parity_leg(res, [0, 1], outs, ("detect", "ocr", "inpaint"))
    assert par["ok"] and par["pages"] == 2 and par["ocr"]["lines"] == 8 and par["inpaint"]["max_abs_u8_diff"] == 0
    res.inpainted = res.inpainted.clone()
    res.inpainted[0, :8, :8] = res.inpainted[0, :8, :8] ^ 0x40
    res.ocr_tokens[0, 1] += 1
    par = bench.parity_leg(res, [0, 1], outs, ("detect", "ocr", "inpaint"))
    assert not par["ok"] and par["ocr"]["lines_with_different_tokens"] == 1 and par["inpaint"]["max_abs_u8_diff"] == 64


def test_split_tile_roofline_view():
    """The opt-in split-bf16 mode prices its dominant tile against the bf16 MFMA peak with the executed (6x / 9x) FLOPs."""
    import bench

    assert bench.split_tile_roofline("fast128x128x16w4c", 114.0) is None
    r = bench.split_tile_roofline("split128x128x16p6o", 161.8)
    assert r["plane_pairs"] == 6 and abs(r["executed_bf16_tflops"] - 970.8) < 0.1 and abs(r["frac_of_bf16_mfma_peak"] - 0.3883) < 1e-3
    assert bench.split_tile_roofline("split128x128x16p9m", 126.6)["plane_pairs"] == 9


HEADLINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9][a-z]_bench_headline.json")))


@pytest.mark.parametrize("path", HEADLINES, ids=[os.path.basename(p) for p in HEADLINES])
def test_committed_stdout_of_a_driver_style_run_parses_from_an_8k_tail(path):
    """What bench.py actually printed on the GPU box for the driver's command (round 6 on): the whole stdout, as the driver sees it."""
    out = open(path).read()
    assert len(out) < 4096
    d = json.loads(out[-8000:].strip().splitlines()[-1])
    assert d["metric"].startswith("pages/sec end-to-end") and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    full = os.path.join(ROOT, "profiles", os.path.basename(path).replace("_headline", ""))
    assert json.load(open(full))["value"] == d["value"]
