#!/bin/bash
# experiments build: the parked tile variants (two tiles per workgroup, grid-strided multi-tile, persistent fp32 tile) + ablations
O=gpurun_out/r03d; mkdir -p $O
timeout 300 scripts/split_check 30 --ablate > $O/split_check_experiments.log 2>&1; echo "rc $?" >> $O/split_check_experiments.log
grep -v "\.\.\.$" $O/split_check_experiments.log | tail -150
