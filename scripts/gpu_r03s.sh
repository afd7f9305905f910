#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== rows kernel on"; python scripts/diag_world.py 2>&1 | grep -v amdgpu.ids | tail -24
echo "== MIT_ATT_NO_ROWS=1"; MIT_ATT_NO_ROWS=1 python scripts/diag_world.py 2>&1 | grep -v amdgpu.ids | tail -24
