"""MFMA-busy per stage from rocprofv3 --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; counters-only runs).
usage: pmc_mfma.py <out.json> <pages> <stage>=<counter_collection.csv> ...
SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's 1024 SIMDs (256 CUs x 4), the cycles a SIMD's matrix pipe is busy
(MI355X_MICROARCH.md: = 32 x N for v_mfma_f32_32x32x16_bf16, 64 x N for v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE the cycles the
dispatch kept the GPU busy (summed over the 8 XCDs by the profiler when it exceeds duration x max clock: divided back here).
mfma_busy = MFMA_BUSY / (1024 x active cycles): the fraction of all matrix-pipe cycles of the stage's kernels that were used — the
counter-level counterpart of bench.py's frac_of_mfma_roofline (which prices the same kernels by FLOPs / peak)."""
import collections, csv, json, re, sys
csv.field_size_limit(1 << 30)
out_path, pages = sys.argv[1], int(sys.argv[2])
SIMDS, MAX_GHZ = 1024, 2.4
res = {"pages": pages, "stages": {}}
for arg in sys.argv[3:]:
    stage, path = arg.split("=", 1)
    if not path:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|mitcg::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0][:80]
        k = (name, r["Dispatch_Id"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        per[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = collections.defaultdict(float)
    by_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    for (name, _), c in per.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if c["ns"] > 0 and gui > 4.0 * c["ns"] * MAX_GHZ:   # reported as the sum over the 8 XCDs
            gui /= 8.0
        for dst in (tot, by_kernel[name]):
            dst["mfma_busy_cycles"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            dst["active_cycles"] += gui
            dst["ns"] += c["ns"]
            dst["launches"] += 1
    def summary(d):
        act = d["active_cycles"]
        return dict(launches=int(d["launches"]), ms=round(d["ns"] / 1e6, 3), mfma_busy=round(d["mfma_busy_cycles"] / (SIMDS * act), 4) if act else None,
                    effective_clock_GHz=round(act / d["ns"], 3) if d["ns"] else None)
    top = sorted(by_kernel.items(), key=lambda kv: -kv[1]["ns"])[:8]
    res["stages"][stage] = dict(**summary(tot), kernels={k: summary(v) for k, v in top})
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({s: {k: v for k, v in e.items() if k != "kernels"} for s, e in res["stages"].items()}))
