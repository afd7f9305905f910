#!/bin/bash
# PMC passes over one conv shape (scripts/bench_conv.py ONLY=...), counters-only runs (no trace domains).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_conv; mkdir -p $OUT
export ONLY="${ONLY:-lama fused}" WIDE="${WIDE:-16}" NARROW="${NARROW:-9}" ROUNDS=1 B=8
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT -o p$i -- python scripts/bench_conv.py > $OUT/p$i.log 2>&1
done
python scripts/pmc_summarize.py "${KPAT:-conv_gemm_fast_kernel}" $(find $OUT -name "*counter_collection.csv") > $OUT/summary.json
cat $OUT/summary.json
find $OUT -name "*counter_collection.csv" -delete
