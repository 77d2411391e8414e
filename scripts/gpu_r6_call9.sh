#!/bin/bash
# round 6, session 3, call 5: full GPU suite with the pair layout on by default, then every other BASELINE config with the option off / on
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call9; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_full.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest_gpu_full.log
for v in 0 1; do echo "pair_layout=$v"; timeout 900 python scripts/bench_configs.py cfg1 cfg3 cfg4 cfg5 cfg5-1024 --nfe 8 --opt pair_layout=$v 2>&1 | grep "ms/NFE" | cut -c1-120; done | tee $OUT/configs_pair_layout.log
