#!/bin/bash
# GPU call r03b: full -m gpu suite (no -x), bench, B=1 profile after the always-split / 64x64 split tile / split-K / GPU mask tail changes
set -x
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
PROF_B1_NO_CPROFILE=1 python scripts/prof_b1.py > $O/prof_b1.json 2> $O/prof_b1.err; cat $O/prof_b1.json
MIT_GEMM_SPLIT_MIN_TILES=1280 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg > $O/bench_min1280.json 2> $O/bench_min1280.err
python -c "
import json
for f in ('bench','bench_min1280'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['dropin']['value'], d['dropin']['ms_per_stage'])
"
