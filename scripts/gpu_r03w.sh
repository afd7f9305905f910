#!/bin/bash
# r03w: the 128x128 split tile without scratch spills — correctness, the concurrent-engine diagnostics, timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03w; mkdir -p $O
timeout 200 scripts/split_check 20 > $O/split_check.log 2>&1; grep -E "^(1x1 320|3x3 reflect|winograd|1x1 1280)|split128x128x16p6o|SPLIT CHECK" $O/split_check.log | head -24
(cd scripts; timeout 100 python diag_rfft_culprit.py gemm 2>&1 | grep -v amdgpu.ids | tail -1; timeout 200 python diag_concurrent2.py 2>&1 | grep -v amdgpu.ids | tail -6) | tee $O/diag.log
timeout 600 python -m pytest tests/test_gemm_split_gpu.py tests/test_pipeline_gpu.py tests/test_dist_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03w -- python bench.py --no-cpu-baseline --no-dropin --no-fp32-leg > $O/bench_under_rocprof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03w/bench_under_rocprof.json')); r=d['roofline']
print('value',d['value'],'dominant',r['tile_config'],r['achieved'],r['frac'],r['avg_launch_us'], {k:v['ms_per_page'] for k,v in r['stages'].items()})
PY
head -3 $O/kernel_stats.csv | cut -c1-160
