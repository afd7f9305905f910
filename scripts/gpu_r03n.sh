#!/bin/bash
# r03n: same-box A/Bs — BK = 32 small split tile on the headline; coupled batch pipeline slot size / mask workers
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03n; mkdir -p $O
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-fp32-leg --no-coupled --no-roofline"
for i in 1 2; do
  python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bk32 on ', d['value'])"
  MIT_CONV_NO_SMALL_BK32=1 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bk32 off', d['value'])"
done
for gw in "16 4" "8 4" "8 8" "4 8"; do set -- $gw; g=$1; w=$2
  python bench.py --coupled-only --steps 2 --coupled-group $g --coupled-mask-workers $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('coupled group $g workers $w', d['value'], d['host_ms_per_page_by_phase'])"
done
python bench.py --coupled-only --steps 2 --coupled-group 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('coupled unpipelined', d['value'], d['host_ms_per_page_by_phase'])"
