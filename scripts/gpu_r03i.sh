#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
timeout 120 scripts/split_check 20 > $O/split_check.log 2>&1; grep -A8 "decoder-like" $O/split_check.log | grep -v "\.\.\.$"; tail -1 $O/split_check.log
python -m pytest tests/test_gemm_split_gpu.py tests/test_ocr_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_gpu.py tests/test_coupled_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
PROF_B1_NO_CPROFILE=1 python scripts/prof_b1.py ocr,detect 2>/dev/null
