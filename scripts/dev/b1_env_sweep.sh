#!/bin/bash
# B = 1 plugin calls (scripts/prof_b1.py) under a list of environment settings: usage  b1_env_sweep.sh <tag> "<stages>" "VAR=.. VAR2=.." "VAR=.." ...
TAG=$1; STAGES=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
for setting in "" "$@"; do
  echo "== [$setting]" | tee -a $OUT/sweep.log
  env $setting PROF_B1_NO_CPROFILE=1 python scripts/prof_b1.py $STAGES 2>/dev/null | tee -a $OUT/sweep.log
done
