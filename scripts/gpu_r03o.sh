#!/bin/bash
# r03o: refine_mask CCL with pre-linked wave runs / reduced unions / flatten pass — parity + timing through the coupled leg
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_ctd_refine_gpu.py tests/test_coupled_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-roofline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03o/bench.json'))
c=d['coupled']; print('value',d['value'],'coupled batch',c['batch']['value'],c['batch']['host_ms_per_page_by_phase']); print('b1',c['b1_plugins']['value'],c['b1_plugins']['ms_per_stage'])
g=c['b1_plugins']['glue_kernels']
for k,v in g.items():
    print(k, v['probed_wall_ms'], v['gpu_kernel_ms'])
    for kk,vv in list(v['kernels'].items())[:6]: print('   ',kk,vv)
PY
