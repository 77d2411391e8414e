#!/bin/bash
# fabric-side read / write traffic per launch of every kernel of one small BASELINE config (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE; counters in
# their own run, as the guide prescribes):   usage gpu_pmc_cfg.sh cfg1
set -u
CFG=${1:-cfg1}
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$CFG
rm -rf $OUT /tmp/pmc_$CFG; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmc_$CFG -o run -- python scripts/bench_configs.py $CFG --nfe 4 > $OUT/run.log 2>&1
echo "exit $?"; tail -2 $OUT/run.log
python scripts/summarize_pmc.py /tmp/pmc_$CFG --json $OUT/pmc_gemm.json > $OUT/pmc_summary.txt 2>&1
grep -E "gemm_bf16|attn_small|gated_residual" $OUT/pmc_summary.txt | cut -c1-220 | head -20
