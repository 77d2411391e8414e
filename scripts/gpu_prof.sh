#!/bin/bash
# rocprofv3 kernel-trace summary + PMC passes of the headline bench (separate runs, as the guide prescribes)
set -u
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1  # the engine's own HIP events off: rocprofv3's trace is the measurement here
R=$GRAFT_REPO_ROOT
OUT=${PROF_OUT:-$R/gpurun_out/prof}
rm -rf $OUT; mkdir -p $OUT
cd $R
STEPS=${PROF_STEPS:-6}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o bench -- python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace exit $?"; grep -E '^\{"metric' $OUT/trace.log | cut -c1-400; tail -2 $OUT/trace.log
find /tmp/prof_trace -type f | head -20
for f in $(find /tmp/prof_trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -14 $f | cut -c1-200; done
for f in $(find /tmp/prof_trace -name "*domain_stats.csv"); do cp $f $OUT/domain_stats.csv; done
timeout 900 rocprofv3 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/prof_pmc1 -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
echo "pmc1 exit $?"; tail -2 $OUT/pmc1.log
timeout 900 rocprofv3 --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_pmc2 -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1
echo "pmc2 exit $?"; tail -2 $OUT/pmc2.log
find /tmp/prof_pmc1 /tmp/prof_pmc2 -type f | head
python scripts/summarize_pmc.py /tmp/prof_pmc1 /tmp/prof_pmc2 --json $OUT/pmc_gemm.json --json-attn $OUT/pmc_attn.json > $OUT/pmc_summary.txt 2>&1; head -40 $OUT/pmc_summary.txt | cut -c1-220
du -sh $OUT
