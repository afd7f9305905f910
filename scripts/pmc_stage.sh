#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate counters-only passes) of one bench stage.  usage: pmc_stage.sh <stage> <pages> <out.json>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STAGE=${1:-inpaint}; PAGES=${2:-16}; OUT=${3:-gpurun_out/pmc_stage.json}
D=/tmp/pmc_stage; rm -rf $D; mkdir -p $D $(dirname $OUT)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $D -o $c -- python bench.py --steps 1 --warmup 0 --pages $PAGES --distinct $PAGES --stages $STAGE --no-cpu-baseline --no-roofline > $D/$c.log 2>&1
done
python scripts/pmc_traffic.py $OUT $(find $D -name "*counter_collection.csv")
