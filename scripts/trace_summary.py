"""Summarise a rocprofv3 kernel_trace CSV: per (kernel, grid, workgroup) launch count / total / mean duration."""
import csv, sys, re, collections
csv.field_size_limit(1 << 30)
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if pat and pat not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name)[:60]
    grid = (r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z")) if "Grid_Size_X" in r else (r.get("Grid_Size"),)
    wg = (r.get("Workgroup_Size_X"),) if "Workgroup_Size_X" in r else (r.get("Workgroup_Size"),)
    a = agg[(short, grid, wg)]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:60s} grid={k[1]} wg={k[2]} n={a[0]:6d} total={a[1]/1e3:9.2f}ms avg={a[1]/a[0]:9.1f}us {100*a[1]/tot:5.1f}%")
print(f"total {tot/1e3:.2f} ms")
