#!/bin/bash
# One GPU-box call that produces the evidence of a round: GPU tests, the bench line, a rocprofv3 kernel-trace of the same
# command, and the PMC traffic passes.  usage: scripts/gpu_evidence.sh <tag> [skip-tests]
set -u
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${2:-}" != "skip-tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
# the driver's command; stdout = the <= 4 KB headline (kept as it was printed), the full record -> bench.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --prof-dump $OUT/layers.csv --details $OUT/bench.json > $OUT/bench_headline.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python bench.py --no-cpu-baseline --no-dropin --no-fp32-leg --no-two-streams --no-other-configs --details $OUT/bench_under_rocprof.json > /dev/null 2> $OUT/rocprof.err; echo "rocprof rc=$?"
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
if [ "${3:-}" = "no-pmc" ]; then cat $OUT/bench_headline.json; exit 0; fi   # counters were taken by an earlier call of the round
bash scripts/pmc_stage.sh detect,ocr,inpaint 64 $OUT/pmc_traffic.json > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
bash scripts/pmc_mfma.sh 16 $OUT/mfma_busy.json > $OUT/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"; tail -1 $OUT/pmc_mfma.log
MIT_GEMM_SPLIT=0 bash scripts/pmc_mfma.sh 16 $OUT/mfma_busy_fp32.json > $OUT/pmc_mfma_fp32.log 2>&1; tail -1 $OUT/pmc_mfma_fp32.log
cat $OUT/bench_headline.json
