#!/bin/bash
# rocprofv3 per-dispatch kernel trace of bench_configs.py configs -> sum of kernel durations vs wall per NFE (scripts/trace_wall_vs_sum.py)
#   scripts/gpu_trace_wall.sh cfg1 cfg5        (output: gpurun_out/trace_wall/<cfg>_{wall_vs_sum.txt,kernel_stats.csv})
set -u
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_wall
mkdir -p $OUT
cd $R
for CFG in "$@"; do
  rm -rf /tmp/tw_$CFG
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tw_$CFG -o run -- python scripts/bench_configs.py $CFG --nfe 12 > $OUT/${CFG}_run.log 2>&1
  echo "$CFG exit $?"; tail -1 $OUT/${CFG}_run.log | cut -c1-200
  cp $(find /tmp/tw_$CFG -name "*kernel_stats.csv" | head -1) $OUT/${CFG}_kernel_stats.csv
  python scripts/trace_wall_vs_sum.py /tmp/tw_$CFG ${MARKER:-ode_combine} > $OUT/${CFG}_wall_vs_sum.txt 2>&1
  tail -12 $OUT/${CFG}_wall_vs_sum.txt
done
