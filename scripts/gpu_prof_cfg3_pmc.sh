#!/bin/bash
# rocprofv3 kernel stats + one PMC pass (matrix-pipe duty) of cfg 3 (Flag-DiT 5B): the hd-96 attention kernel's counters
set -u
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_cfg3
rm -rf $OUT /tmp/prof_cfg3 /tmp/pmc_cfg3; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg3 -o run -- python scripts/bench_configs.py cfg3 --nfe 4 > $OUT/run.log 2>&1
echo "trace exit $?"; tail -1 $OUT/run.log | cut -c1-200
for f in $(find /tmp/prof_cfg3 -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -12 $f | cut -c1-150; done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_cfg3 -o run -- python scripts/bench_configs.py cfg3 --nfe 2 > $OUT/pmc.log 2>&1
echo "pmc exit $?"
python scripts/summarize_pmc.py /tmp/pmc_cfg3 > $OUT/pmc_summary.txt 2>&1; grep -E "attn|w4q" $OUT/pmc_summary.txt | cut -c1-220
