#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call18
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "qk_norm or qkv_post or golden or switches" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 200 python scripts/opbench.py elem --rounds 9 2>&1 | grep "qk_"
LUMINA_DIT_LIB=$R/lumina-t2x_amd/lib/liblumina_dit_old.so timeout 200 python scripts/opbench.py elem --rounds 9 2>&1 | grep "qk_" | sed 's/^/OLD /'
