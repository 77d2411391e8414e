#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call3
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_samplers.py tests/test_gpu_variants.py -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 $OUT/pytest_gpu.log
