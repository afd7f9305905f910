"""GPU idle time of a rocprofv3 --kernel-trace run, attributed to the kernel that ended each gap (= whose launch came late).

usage: python scripts/dev/gap_report.py <kernel_trace.csv> [min_gap_us [after_last_kernel_substring | last=<ms>]]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if len(sys.argv) > 3 and sys.argv[3].startswith("last="):   # only the last N ms of the trace (the timed steps)
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - float(sys.argv[3][5:]) * 1e6]
elif len(sys.argv) > 3:   # only the part of the trace after the last launch of this kernel (engine construction: the weight packers)
    last = max(i for i, r in enumerate(rows) if sys.argv[3] in r[2])
    rows = rows[last + 1:]
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
busy_end = rows[0][1]
t0, gaps, tot_gap = rows[0][0], collections.defaultdict(lambda: [0, 0.0]), 0.0
prev_name = rows[0][2]
pairs = collections.defaultdict(lambda: [0, 0.0])
big = []
for s, e, n in rows[1:]:
    if s > busy_end:
        g = s - busy_end
        tot_gap += g
        if g >= min_gap:
            k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
            gaps[k][0] += 1
            gaps[k][1] += g
            pk = (prev_name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:], k[-40:])
            pairs[pk][0] += 1
            pairs[pk][1] += g
            big.append((g, (busy_end - t0) / 1e6, pk[0], k[-40:]))
    if e > busy_end:
        busy_end, prev_name = e, n
span = busy_end - t0
print(f"span {span / 1e6:.1f} ms, idle {tot_gap / 1e6:.1f} ms ({100 * tot_gap / span:.1f} %), kernels {len(rows)}")
print("idle time by the kernel that ended the gap (gaps >= %.0f us):" % (min_gap / 1e3))
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {g / 1e6:9.2f} ms  {c:6d} gaps  {k}")
print("by (kernel before, kernel after):")
for k, (c, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {g / 1e6:9.2f} ms  {c:6d} gaps  {k[0]}  ->  {k[1]}")
print("largest gaps:")
for g, t, a, b in sorted(big, reverse=True)[:25]:
    print(f"  {g / 1e6:8.3f} ms at t = {t:9.1f} ms   {a} -> {b}")
