#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call17
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fulldepth.py -m gpu -q -x -k "norm or golden or full_2b or switches" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
timeout 200 python scripts/opbench.py elem --rounds 7 2>&1 | grep "gated_residual\|rmsnorm"
LUMINA_DIT_LIB=$R/lumina-t2x_amd/lib/liblumina_dit_old.so timeout 200 python scripts/opbench.py elem --rounds 7 2>&1 | grep "gated_residual\|rmsnorm" | sed 's/^/OLD /'
bash scripts/gpu_bench_ab_lib.sh 2
