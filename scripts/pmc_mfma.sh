#!/bin/bash
# MFMA-busy per stage: one counters-only rocprofv3 pass per stage (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; no trace domains beside
# --pmc), summarised by scripts/pmc_mfma.py into one JSON that bench.py joins into roofline.stages[*].mfma_busy.
# usage: pmc_mfma.sh <pages> <out.json>        e.g.  pmc_mfma.sh 16 gpurun_out/r03_mfma_busy.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PAGES=${1:-16}; OUT=${2:-gpurun_out/mfma_busy.json}
D=/tmp/pmc_mfma; rm -rf $D; mkdir -p $D $(dirname $OUT)
for s in detect ocr inpaint; do
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $D/$s -o $s -- \
    python bench.py --steps 1 --warmup 0 --pages $PAGES --stages $s --no-overlap --no-cpu-baseline --no-roofline --no-dropin --no-fp32-leg --no-two-streams --no-other-configs > $D/$s.log 2>&1
done
python scripts/pmc_mfma.py $OUT $PAGES detect=$(find $D/detect -name "*counter_collection.csv" | head -1) ocr=$(find $D/ocr -name "*counter_collection.csv" | head -1) \
  inpaint=$(find $D/inpaint -name "*counter_collection.csv" | head -1)
