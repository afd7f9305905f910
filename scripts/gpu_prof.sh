#!/bin/bash
# usage: scripts/gpu_prof.sh <tag> [bench args...] — the headline step under rocprofv3 --kernel-trace --stats: kernel_stats.csv + the bench line
TAG=${1:-prof}; shift
OUT=gpurun_out/$TAG; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python bench.py --steps 3 --no-cpu-baseline --no-dropin --no-fp32-leg --no-two-streams --no-coupled --no-roofline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof rc=$?"
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
head -c 600 $OUT/bench_under_rocprof.json; echo; tail -2 $OUT/rocprof.err
