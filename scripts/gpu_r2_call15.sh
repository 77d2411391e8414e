#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call15
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest.log
timeout 300 python scripts/opbench.py attn --rounds 7 --attn-variants 3,4,5 2>&1 | grep -v amdgpu.ids | tail -4
