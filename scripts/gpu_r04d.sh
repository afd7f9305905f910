#!/bin/bash
# round 4, call d: SQ counters of mit_pgemm tiles on the long-K case (where the K loop is everything)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/manga_image_translator_amd:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
PG_CASE="long K" timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/r04d_pmc_sq -o pmc -- $R/scripts/pgemm_check 2 > $R/gpurun_out/r04d_pmc_sq.log 2>&1
PG_CASE="long K" timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/r04d_pmc_sq2 -o pmc -- $R/scripts/pgemm_check 2 > $R/gpurun_out/r04d_pmc_sq2.log 2>&1
tail -2 $R/gpurun_out/r04d_pmc_sq.log $R/gpurun_out/r04d_pmc_sq2.log
