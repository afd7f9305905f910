"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters-only runs, no trace domains).
usage: pmc_traffic.py <out.json> <pages per step> <counter_collection.csv>...
Units per MI355X_MICROARCH.md (HBM section): both counters are in KB on gfx950 and FETCH_SIZE reports half the bytes of wide
coalesced reads; the x2 correction is applied in `hbm_read_MB_per_launch`, the raw averages are kept next to it.  Keys are the
kernels' names without namespace / return type / argument list, e.g. "conv_gemm_fast_kernel<128, 128, 16, 1, 4, 4, 4>" — the
same strings mit_conv_gemm_config_kernel() and the kernel probe report, which is how bench.py joins them."""
import collections, csv, json, re, sys
csv.field_size_limit(1 << 30)
pages = int(sys.argv[2])
out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|mitcg::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0][:80]
        a = out[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
res = {}
for k, cs in out.items():
    e = {"pages": pages}
    for c, (n, v, ns) in cs.items():
        e[c] = dict(launches=n, avg_KB_per_launch=round(v / n, 1), sum_MB=round(v / 1e3, 1), avg_us=round(ns / n / 1e3, 1))
    if "FETCH_SIZE" in e:
        e["hbm_read_MB_per_launch"] = round(2 * e["FETCH_SIZE"]["avg_KB_per_launch"] / 1e3, 1)
    if "WRITE_SIZE" in e:
        e["hbm_write_MB_per_launch"] = round(e["WRITE_SIZE"]["avg_KB_per_launch"] / 1e3, 1)
    res[k] = e
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("sum_MB", 0))[:16]:
    print(k, {c: (v if not isinstance(v, dict) else (v["launches"], v["avg_KB_per_launch"], v["avg_us"])) for c, v in e.items()})
