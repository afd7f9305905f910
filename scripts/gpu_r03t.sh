#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== rows kernel on"; timeout 300 python scripts/diag_concurrent.py 12 2>&1 | grep -v amdgpu.ids | tail -6
echo "== MIT_ATT_NO_ROWS=1"; MIT_ATT_NO_ROWS=1 timeout 300 python scripts/diag_concurrent.py 12 2>&1 | grep -v amdgpu.ids | tail -6
