#!/bin/bash
# round 4, call e: new GPU tests (pgemm, co-tenancy, the un-serialised two-rank rehearsal) and the LaMa tests after the FFT rows change
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_pgemm_gpu.py tests/test_cotenant_gpu.py tests/test_dist_gpu.py tests/test_lama_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04e_pytest.log
cat gpurun_out/r04e_pytest.log
