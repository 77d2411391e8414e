#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call11
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or swiglu" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 300 python scripts/gemm_trace_w4q.py 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/opbench.py gemm --rounds 5 --gemm-variants 0,15,16 2>&1 | grep -v amdgpu.ids | tail -16
