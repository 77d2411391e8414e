#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call5
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" > $OUT/pytest_attn.log 2>&1; echo "pytest attn exit $?"; tail -6 $OUT/pytest_attn.log
timeout 300 python scripts/opbench.py attn96 --rounds 5 --attn-variants 1,3 > $OUT/opbench_attn96.log 2>&1; echo "opbench exit $?"; tail -3 $OUT/opbench_attn96.log
timeout 300 python scripts/opbench.py attn --rounds 5 --attn-variants 2,3 > $OUT/opbench_attn72.log 2>&1; tail -2 $OUT/opbench_attn72.log
timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fulldepth.py -m gpu -q -x -s -k "flag" > $OUT/pytest_flag.log 2>&1; echo "pytest flag exit $?"; tail -4 $OUT/pytest_flag.log; grep "engine vs reference" $OUT/pytest_flag.log
timeout 400 python scripts/bench_configs.py cfg3 > $OUT/bench_cfg3.log 2>&1; tail -1 $OUT/bench_cfg3.log
timeout 400 python scripts/bench_configs.py cfg1 cfg5 --pairs 4 > $OUT/bench_pairs4.log 2>&1; tail -2 $OUT/bench_pairs4.log
timeout 300 python scripts/opbench.py gemm_b4 --rounds 3 --gemm-variants 0,1,2,7,8 --cold 16 > $OUT/opbench_gemm_b4.log 2>&1; tail -22 $OUT/opbench_gemm_b4.log
