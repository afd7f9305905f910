#!/bin/bash
set -x
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -30 $O/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err
python bench.py --config5 --steps 2 --warmup 1 > $O/config5.json 2> $O/config5.err; tail -c 600 $O/config5.err
python bench.py --config1 --steps 2 --warmup 1 > $O/config1.json 2> $O/config1.err; tail -c 600 $O/config1.err
timeout 120 scripts/split_check 20 > $O/split_check.log 2>&1; grep -A6 "ESRGAN\|decoder-like" $O/split_check.log | grep -v "\.\.\.$"
python -c "
import json
d=json.load(open('$O/bench.json')); print('headline', d['value'], 'fp32', d['fp32_mfma']['value'], 'dropin', d['dropin']['value']); print(json.dumps(d['coupled'])[:3000]); print(d.get('leg_errors'))
for f in ('config5','config1'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], {k:(v['ms'], v['conv_exec_tflops'], v['frac_of_mfma_roofline']) for k,v in d['stages_ms_per_page'].items()})
"
bash scripts/pmc_mfma.sh 16 $O/mfma_busy.json
