#!/bin/bash
# rocprofv3 per-dispatch kernel trace of one bench_configs.py config: per-kernel stats + the durations of one kernel by call order
#   scripts/gpu_trace_cfg.sh <cfg> <kernel substring>
set -u
CFG=${1:-cfg5-1024}; PAT=${2:-moe_plan}
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_$CFG
rm -rf $OUT /tmp/trace_$CFG; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$CFG -o run -- python scripts/bench_configs.py $CFG --nfe 4 > $OUT/run.log 2>&1
echo "exit $?"; tail -1 $OUT/run.log
f=$(find /tmp/trace_$CFG -name "*kernel_trace.csv" | head -1)
s=$(find /tmp/trace_$CFG -name "*kernel_stats.csv" | head -1)
cp $s $OUT/kernel_stats.csv
python - "$f" "$PAT" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "dispatches of", sys.argv[2])
tail = d[-64:]
print("last 64 (us):", " ".join(f"{x:.0f}" for x in tail))
print("even calls avg %.1f  odd calls avg %.1f" % (sum(tail[0::2]) / len(tail[0::2]), sum(tail[1::2]) / len(tail[1::2])))
PY
