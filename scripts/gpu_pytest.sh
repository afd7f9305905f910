#!/bin/bash
# usage: scripts/gpu_pytest.sh <tag> [pytest args...] — the GPU test suite (or a selection) on the GPU box, log under gpurun_out/
TAG=${1:-pytest}; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
if [ $# -eq 0 ]; then set -- tests; fi
timeout 2400 python -m pytest "$@" -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
