#!/bin/bash
# round 4, call a: first run of mit_pgemm (bit-identity with the split tiles + timings), the pending co-tenant variants of round 3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/manga_image_translator_amd:$LD_LIBRARY_PATH
( timeout 600 scripts/pgemm_check 10 > gpurun_out/r04a_pgemm_check.log 2>&1; echo "exit $?" >> gpurun_out/r04a_pgemm_check.log )
( timeout 120 scripts/cotenant_check 60 14 > gpurun_out/r04a_cotenant_check.log 2>&1; echo "exit $?" >> gpurun_out/r04a_cotenant_check.log )
tail -5 gpurun_out/r04a_pgemm_check.log
