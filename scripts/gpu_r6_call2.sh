#!/bin/bash
# round 6, call 2: full GPU suite, headline profile set (rocprofv3 stats + PMC incl. the attention summary), wall-vs-sum traces of the
# 512-row configs, wave-level counters of the persistent GEMM at prefetch distance 3 / 4 and on the O shape, the bench line with its CPU leg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call2; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -s > $OUT/pytest_gpu_full_suite.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu_full_suite.log
PROF_OUT=$OUT/prof bash scripts/gpu_prof.sh > $OUT/prof.log 2>&1; tail -25 $OUT/prof.log | cut -c1-200
bash scripts/gpu_trace_wall.sh cfg1 cfg5 2>&1 | tail -30 | cut -c1-200; cp -r gpurun_out/trace_wall $OUT/
PMC_OUT=$OUT/pmc_w13_pd3 PMC_PASSES="p1 p4 p6" LT_PMC_VENDOR=1 bash scripts/gpu_pmc_gemm_stalls.sh; cat $OUT/pmc_w13_pd3/summary.txt | cut -c1-400
LUMINA_DIT_LIB=$R/lumina-t2x_amd/lib/pd4/liblumina_dit.so PMC_OUT=$OUT/pmc_w13_pd4 PMC_PASSES="p1 p4 p6" LT_PMC_VENDOR=0 bash scripts/gpu_pmc_gemm_stalls.sh; cat $OUT/pmc_w13_pd4/summary.txt | cut -c1-400
LT_PMC_SHAPE=8192,2304,2304,16 PMC_OUT=$OUT/pmc_o_pd3 PMC_PASSES="p1 p4 p6" LT_PMC_VENDOR=0 bash scripts/gpu_pmc_gemm_stalls.sh; cat $OUT/pmc_o_pd3/summary.txt | cut -c1-400
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench exit $?"; cut -c1-600 $OUT/bench_full.json
du -sh $OUT
