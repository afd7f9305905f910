"""Summarise rocprofv3 --pmc counter_collection CSVs: per (kernel substring, counter) mean value per dispatch."""
import csv, sys, json, collections
csv.field_size_limit(1 << 30)
pat = sys.argv[1]
out = collections.defaultdict(lambda: [0, 0.0])
dur = collections.defaultdict(lambda: [0, 0.0])
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if pat not in r["Kernel_Name"]:
            continue
        k = r["Counter_Name"]
        out[k][0] += 1
        out[k][1] += float(r["Counter_Value"])
        dur[k][0] += 1
        dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {k: dict(n=v[0], mean=v[1] / v[0], mean_ns=dur[k][1] / dur[k][0]) for k, v in out.items()}
print(json.dumps(res, indent=1))
