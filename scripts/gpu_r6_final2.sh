#!/bin/bash
# round 6, final measurement set at HEAD (pair layout on): headline profile (rocprofv3 stats + PMC summaries), per-config kernel stats,
# every other BASELINE config, the bench line twice, the full GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/final2; mkdir -p $OUT
cd $R
PROF_OUT=$OUT/prof bash scripts/gpu_prof.sh > $OUT/prof.log 2>&1; grep -E "exit|^\"void (lt_|\(anon)" $OUT/prof.log | head -12 | cut -c1-150
timeout 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench exit $?"; cut -c1-330 $OUT/bench_final.json
timeout 1500 python scripts/bench_configs.py cfg1 cfg3 cfg4 cfg5 cfg5-1024 --nfe 8 > $OUT/bench_configs_all.log 2>&1; grep "ms/NFE" $OUT/bench_configs_all.log | cut -c1-170
for c in cfg1 cfg3 cfg4 cfg5 cfg5-1024; do PROF_EXTRA="" bash scripts/gpu_prof_cfg.sh $c > $OUT/prof_$c.log 2>&1; cp gpurun_out/prof_$c/kernel_stats.csv $OUT/kernel_stats_$c.csv; done
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_final_2.json 2> $OUT/bench_final_2.err; cut -c1-330 $OUT/bench_final_2.json
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_full.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu_full.log
