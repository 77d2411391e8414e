#!/bin/bash
# round 6, session 3, call 10: pair layout for the MoE experts' grouped GEMMs - parity, then cfg5-1024 off / on
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call14; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "moe or grouped or pair" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest.log
for i in 1 2; do for v in 0 1; do echo "pair_layout=$v"; timeout 900 python scripts/bench_configs.py cfg5-1024 cfg5 --nfe 8 --opt pair_layout=$v 2>&1 | grep "ms/NFE" | cut -c1-120; done; done | tee $OUT/ab_pair_moe.log
