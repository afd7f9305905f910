#!/bin/bash
# round 4, call b: tile order / grid of mit_pgemm (same process each: the knobs are read once), L2 hit counters, co-tenant check of the
# FFT rows kernels with 4-byte butterfly reads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/manga_image_translator_amd:$LD_LIBRARY_PATH
L=gpurun_out/r04b_pgemm_order.log
: > $L
for cfg in "1 -1" "0 -1" "1 0" "1 3" "1 4"; do
  set -- $cfg
  echo "#### MIT_PGEMM_ORDER=$1 MIT_PGEMM_WGS=$2" >> $L
  MIT_PGEMM_ORDER=$1 MIT_PGEMM_WGS=$2 PG_TILE=s3p6 timeout 300 scripts/pgemm_check 10 2>&1 | grep -v "planes:" >> $L
done
( timeout 120 scripts/cotenant_check 100 14 > gpurun_out/r04b_cotenant_check.log 2>&1; echo "exit $?" >> gpurun_out/r04b_cotenant_check.log )
cd /tmp && export TMPDIR=/tmp
for ord in 1 0; do
  MIT_PGEMM_ORDER=$ord PG_CASE="pw1 320" PG_TILE=128x128s3p6 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r04b_pmc_l2_ord$ord -o pmc -- $GRAFT_REPO_ROOT/scripts/pgemm_check 2 > $GRAFT_REPO_ROOT/gpurun_out/r04b_pmc_l2_ord$ord.log 2>&1
done
ls -R $GRAFT_REPO_ROOT/gpurun_out/r04b_pmc_l2_ord1 | head -20
