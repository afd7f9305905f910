#!/bin/bash
# round 4, call c: timing ablations of mit_pgemm (MIT_CONV_EXPERIMENTS build) and the 256 x 256 planar tile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/manga_image_translator_amd:$LD_LIBRARY_PATH
PG_CASE="${PG_CASE:-}" timeout 600 scripts/pgemm_check 10 2>&1 | grep -v "planes:" > gpurun_out/r04c_pgemm_ablate.log
tail -3 gpurun_out/r04c_pgemm_ablate.log
