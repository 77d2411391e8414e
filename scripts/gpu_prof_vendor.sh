#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_vendor
rm -rf $OUT /tmp/prof_vendor; mkdir -p $OUT
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vendor -o run -- python scripts/vendor_gemm_probe.py > $OUT/run.log 2>&1
echo "exit $?"; tail -3 $OUT/run.log
for f in $(find /tmp/prof_vendor -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -12 $f | cut -c1-400; done
