#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate counters-only passes: no trace domains beside --pmc) of one bench step.
# usage: pmc_stage.sh <stages> <pages> <out.json>      e.g.  pmc_stage.sh detect,ocr,inpaint 64 gpurun_out/r02_pmc_traffic.json
# The launch mix equals bench.py's (same --pages / micro-batches), so per-launch averages can be joined with its probe numbers.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STAGES=${1:-detect,ocr,inpaint}; PAGES=${2:-64}; OUT=${3:-gpurun_out/pmc_traffic.json}
D=/tmp/pmc_stage; rm -rf $D; mkdir -p $D $(dirname $OUT)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $D -o $c -- python bench.py --steps 1 --warmup 0 --pages $PAGES --stages $STAGES --no-cpu-baseline --no-roofline --no-dropin --no-fp32-leg --no-two-streams --no-other-configs > $D/$c.log 2>&1
done
python scripts/pmc_traffic.py $OUT $PAGES $(find $D -name "*counter_collection.csv")
