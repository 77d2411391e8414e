#!/bin/bash
# round 6, session 3, call 8: rocprofv3 kernel stats + PMC passes of the headline with the pair layout on (scripts/gpu_prof.sh)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call12; mkdir -p $OUT
cd $R
PROF_OUT=$OUT/prof bash scripts/gpu_prof.sh > $OUT/prof.log 2>&1; grep -E "exit|^\"void (lt_|\(anon)" $OUT/prof.log | head -14 | cut -c1-170
