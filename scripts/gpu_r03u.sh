#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
