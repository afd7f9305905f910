#!/bin/bash
# GPU call r03a: full -m gpu suite in the new default GEMM mode, bench (split headline + fp32_mfma), B=1 / mask-refinement profile
set -x
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python scripts/prof_b1.py > $O/prof_b1.json 2> $O/prof_b1.err; cat $O/prof_b1.json
cd /tmp && export TMPDIR=/tmp
PROF_B1_NO_CPROFILE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $GRAFT_REPO_ROOT/scripts/prof_b1.py ocr,detect,maskref > $GRAFT_REPO_ROOT/$O/prof_b1_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_b1_rocprof.err
cd $GRAFT_REPO_ROOT
find /tmp/prof_b1 -name "*kernel_stats.csv" -exec cp {} $O/prof_b1_kernel_stats.csv \;
head -30 $O/prof_b1_kernel_stats.csv
