#!/bin/bash
# SQ counters of the split tiles on one split_check case (counters-only passes).  usage: pmc_split_tile.sh "<SC_CASE substring>" <outdir>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CASE=${1:-pp: long K}; OUT=${2:-gpurun_out/pmc_split}; mkdir -p $OUT
D=/tmp/pmc_split; rm -rf $D; mkdir -p $D
SC_CASE="$CASE" timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $D/a -o a -- scripts/split_check 3 > $D/a.log 2>&1
SC_CASE="$CASE" timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $D/b -o b -- scripts/split_check 3 > $D/b.log 2>&1
SC_CASE="$CASE" timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/c -o c -- scripts/split_check 3 > $D/c.log 2>&1
python - "$D" "$OUT" <<'PY'
import csv, glob, sys, collections, json
D, OUT = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(D + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"): n[(k, r["Counter_Name"])] += 1
out = {}
for k, c in agg.items():
    if "split" not in k: continue
    d = {m: v / max(n.get((k, "SQ_WAVE_CYCLES"), 1), 1) for m, v in c.items()}
    out[k] = d
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for m in sorted(d): print(f"   {m:28s} {d[m]:14.0f}  {d[m] / w:8.3f} of WAVE_CYCLES")
json.dump(out, open(OUT + "/pmc_split_tile.json", "w"), indent=1)
PY
