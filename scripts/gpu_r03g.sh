#!/bin/bash
set -x
O=gpurun_out/r03g; mkdir -p $O
python -m pytest tests/test_coupled_gpu.py tests/test_mask_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg > $O/bench.json 2> $O/bench.err; tail -c 1000 $O/bench.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-dropin --lama-mb 32 --ctd-mb 32 --group 32 > $O/bench_mb32.json 2> $O/bench_mb32.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-dropin --lama-mb 8 --ctd-mb 16 --group 16 > $O/bench_mb8.json 2> $O/bench_mb8.err
python -c "
import json
d=json.load(open('$O/bench.json')); print('headline', d['value'], 'dropin', d['dropin']['value'], d['dropin']['ms_per_stage']); print(json.dumps(d['coupled'])[:2500]); print(d.get('leg_errors'))
print('mb32', json.load(open('$O/bench_mb32.json'))['value'], 'mb8', json.load(open('$O/bench_mb8.json'))['value'])
"
