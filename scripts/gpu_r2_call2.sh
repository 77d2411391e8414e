#!/bin/bash
# round 2, GPU call 2: the whole -m gpu suite on the refactored build + a default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call2
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_gpu.log; grep -E "^full_|engine vs reference" $OUT/pytest_gpu.log | head -12
timeout 300 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err; echo "bench exit $?"; tail -c 2500 $OUT/bench.json.log; tail -3 $OUT/bench.err
