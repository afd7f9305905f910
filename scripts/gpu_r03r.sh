#!/bin/bash
# r03r: bisect the world-size dependence of the page records (test_dist_gpu) over the two new dispatch rules
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03r; mkdir -p $O
T=tests/test_dist_gpu.py::test_page_records_do_not_depend_on_world_size_and_rccl_smoke
MIT_CONV_NO_SMALL_BK32=1 timeout 400 python -m pytest $T -m gpu -x -q > $O/no_bk32.log 2>&1; echo "no bk32: rc=$?"; grep -E "bytes differ|passed|failed" $O/no_bk32.log | tail -2
MIT_ATT_NO_ROWS=1 timeout 400 python -m pytest $T -m gpu -x -q > $O/no_rows.log 2>&1; echo "no rows: rc=$?"; grep -E "bytes differ|passed|failed" $O/no_rows.log | tail -2
