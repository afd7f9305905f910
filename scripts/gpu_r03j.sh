#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests/test_ocr_gpu.py tests/test_pipeline_gpu.py tests/test_coupled_gpu.py tests/test_conv_gemm_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -25 $O/pytest.log | cut -c1-220
PROF_B1_NO_CPROFILE=1 python scripts/prof_b1.py ocr 2>/dev/null
MIT_OCR_DECODE_GRAPH=0 PROF_B1_NO_CPROFILE=1 python scripts/prof_b1.py ocr 2>/dev/null
