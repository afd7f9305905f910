#!/bin/bash
set -x
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_coupled_gpu.py tests/test_ocr_gpu.py tests/test_mask_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err
python bench.py --overlap --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-dropin > $O/bench_overlap.json 2> $O/bench_overlap.err
python -c "
import json
d=json.load(open('$O/bench.json')); print('plain', d['value']); print(json.dumps(d['coupled'])[:3000]); print(d.get('leg_errors'))
d=json.load(open('$O/bench_overlap.json')); print('overlap', d['value'])
"
