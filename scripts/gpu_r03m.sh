#!/bin/bash
# r03m: native mask assignment + early bilateral, BK = 32 small split tile rule, pipelined coupled batch path
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_split_gpu.py tests/test_mask_gpu.py tests/test_coupled_gpu.py tests/test_ocr_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python scripts/prof_coupled_maskref.py > $O/maskref.log 2>&1; grep -E "dispatch ms|lines" $O/maskref.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03m/bench.json'))
print('value',d['value'],'fp32',d['fp32_mfma']['value'],'dropin',d['dropin']['value'],d['dropin']['ms_per_stage'])
c=d['coupled']; print('coupled batch',c['batch']['value'],c['batch']['host_ms_per_page_by_phase']); print('b1',c['b1_plugins']['value'],c['b1_plugins']['ms_per_stage'])
g=c['b1_plugins']['glue_kernels']
for k,v in g.items():
    print(k, v['probed_wall_ms'], v['gpu_kernel_ms'])
    for kk,vv in list(v['kernels'].items())[:8]: print('   ',kk,vv)
print(d['parity_checked']['ok'])
PY
