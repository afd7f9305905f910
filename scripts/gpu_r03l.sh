#!/bin/bash
# r03l: BK = 32 form of the 64x64 split tile on decoder-like shapes (split_check) + host profile of the mask refinement on the coupled page
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03l
timeout 300 scripts/split_check 30 > gpurun_out/r03l/split_check.log 2>&1; echo "split_check rc=$?" >> gpurun_out/r03l/split_check.log
timeout 400 python scripts/prof_coupled_maskref.py > gpurun_out/r03l/maskref.log 2>&1; echo "rc=$?" >> gpurun_out/r03l/maskref.log
grep -E "decoder-like|detector-like|split64|fast64|SPLIT CHECK" gpurun_out/r03l/split_check.log | tail -70
tail -45 gpurun_out/r03l/maskref.log
