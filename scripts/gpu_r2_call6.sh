#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call6
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "16x16x32 or identity or swiglu" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest.log
timeout 300 python scripts/opbench.py gemm --rounds 5 --gemm-variants 0,14,15,16 > $OUT/opbench_gemm.log 2>&1; tail -22 $OUT/opbench_gemm.log
