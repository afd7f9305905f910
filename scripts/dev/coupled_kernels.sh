cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r12q; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c -- python bench.py --coupled-only --no-cpu-baseline > $OUT/coupled.json 2> $OUT/err.log; echo rc=$?
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -c 600 $OUT/coupled.json
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r12q/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6)
for r in rows[:45]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]
    print(f"{n:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}% avg {float(r['AverageNs'])/1e3:8.1f}")
P
