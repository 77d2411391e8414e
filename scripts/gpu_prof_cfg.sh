#!/bin/bash
# rocprofv3 kernel-trace summary of one of the small BASELINE configs (scripts/bench_configs.py): usage gpu_prof_cfg.sh cfg1
set -u
CFG=${1:-cfg1}
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$CFG
rm -rf $OUT /tmp/prof_$CFG; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$CFG -o run -- python scripts/bench_configs.py $CFG --nfe 16 ${PROF_EXTRA:-} > $OUT/run.log 2>&1
echo "exit $?"; tail -2 $OUT/run.log
for f in $(find /tmp/prof_$CFG -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -24 $f | cut -c1-160; done
