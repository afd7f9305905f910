#!/bin/bash
set -x
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests/test_coupled_gpu.py tests/test_mask_gpu.py tests/test_conv_gemm_gpu.py tests/test_gemm_split_gpu.py tests/test_pipeline_gpu.py tests/test_esrgan_gpu.py tests/test_ocr_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print('headline', d['value'], 'fp32', d['fp32_mfma']['value'], 'dropin', d['dropin']['value'], d['dropin']['ms_per_stage']); print(json.dumps(d['coupled'])[:3000]); print(d.get('leg_errors'))
for k,v in sorted(d['conv_gemm_by_tile'].items(), key=lambda kv:-kv[1]['ms']): print(' ',k,v['launches'],v['ms'],v['alg_tflops'])
"
