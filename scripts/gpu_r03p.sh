#!/bin/bash
# r03p: attention_rows_kernel (encoder self-attention with K / V staged once per (head, row)): parity + stage time
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03p; mkdir -p $O
timeout 900 python -m pytest tests/test_ocr_gpu.py tests/test_ocr_ctc_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-dropin > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03p/bench.json'))
print('value',d['value'], {k:v['ms_per_page'] for k,v in d['roofline']['stages'].items()})
for k,v in d['roofline']['hbm_kernels'].items():
    if 'attention' in k or 'xpos' in k or 'layernorm' in k: print(k,v)
PY
