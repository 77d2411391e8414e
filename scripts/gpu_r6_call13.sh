#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call13; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "pair_layout" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest.log
