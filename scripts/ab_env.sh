#!/bin/bash
# A/B of one environment knob inside ONE GPU-box call (boxes differ by a few percent, so never compare across calls):
# usage: scripts/ab_env.sh KNOB "<command>"   -> runs <command> with and without KNOB=1, twice each, interleaved
KNOB=$1; shift
for i in 1 2; do
  echo "--- $KNOB unset"; "$@" 2>&1 | tail -1
  echo "--- $KNOB=1";  env $KNOB=1 "$@" 2>&1 | tail -1
done
