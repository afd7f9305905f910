cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r11a}; mkdir -p $OUT
PROF_B1_NO_CPROFILE=1 python scripts/prof_b1.py ${2:-ocr} > $OUT/prof_b1.json 2> $OUT/prof_b1.err; echo rc=$?
cat $OUT/prof_b1.json
PROF_B1_NO_CPROFILE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o b1 -- python scripts/prof_b1.py ${2:-ocr} > $OUT/prof_b1_under_rocprof.json 2> $OUT/rocprof.err; echo rc=$?
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python - "$f" $OUT <<'P'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last OCR call = the last window of launches: find last embed-chain; simply take the last 1/8 of rows split by big gaps (> 2 ms)
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
calls = []; cur = [ts[0]]
for a, b in zip(ts, ts[1:]):
    if b[0] - a[1] > 1_500_000: calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
out = []
for c in calls[-4:]:
    span = (c[-1][1] - c[0][0]) / 1e6; busy = sum(e - s for s, e, _ in c) / 1e6
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(c, c[1:])]
    per = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in c:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:80]; per[k][0] += 1; per[k][1] += (e - s) / 1e3
    top = sorted(per.items(), key=lambda kv: -kv[1][1])[:40]
    out.append(dict(launches=len(c), span_ms=round(span, 2), busy_ms=round(busy, 2), gap_mean_us=round(sum(gaps) / max(1, len(gaps)), 2),
                    gap_gt20us=sum(1 for g in gaps if g > 20), gap_gt20us_ms=round(sum(g for g in gaps if g > 20) / 1e3, 2),
                    top=[(k, v[0], round(v[1] / 1e3, 3), round(v[1] / v[0], 1)) for k, v in top]))
json.dump(out, open(sys.argv[2] + "/b1_timeline.json", "w"), indent=1)
for o in out: print(o["launches"], o["span_ms"], o["busy_ms"], o["gap_mean_us"], o["gap_gt20us"], o["gap_gt20us_ms"])
for t in out[-1]["top"]: print(t)
import itertools
open(sys.argv[2] + "/b1_last_call_sequence.txt", "w").write("\n".join(f"{(s - calls[-1][0][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {n[:110]}" for s, e, n in calls[-1]))
P
rm -rf $OUT/prof
