#!/bin/bash
cd "$GRAFT_REPO_ROOT/scripts"
for k in att142 att64 gemm rfft lama ocr ctd; do timeout 120 python diag_rfft_culprit.py $k 2>&1 | grep -v amdgpu.ids | tail -1; done
