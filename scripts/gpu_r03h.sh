#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
python scripts/prof_b1.py ocr > $O/prof_ocr.json 2> $O/prof_ocr.err; cat $O/prof_ocr.json; grep -A26 "cProfile of one ocr" $O/prof_ocr.err | cut -c1-150 | tail -20
cd /tmp && export TMPDIR=/tmp
PROF_B1_NO_CPROFILE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $GRAFT_REPO_ROOT/scripts/prof_b1.py ocr > /dev/null 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
find /tmp/prof_b1 -name "*kernel_stats.csv" -exec cp {} $O/ocr_b1_kernel_stats.csv \;
head -22 $O/ocr_b1_kernel_stats.csv | cut -c1-200
