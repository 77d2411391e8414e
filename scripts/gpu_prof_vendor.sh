#!/bin/bash
# kernel trace + two PMC passes of scripts/vendor_gemm_probe.py (vendor GEMM next to the engine's variants on the same operands)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_vendor
rm -rf $OUT /tmp/prof_vendor /tmp/prof_vendor_pmc1 /tmp/prof_vendor_pmc2; mkdir -p $OUT
cd $R
export PROBE_W13_ONLY=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vendor -o run -- python scripts/vendor_gemm_probe.py > $OUT/run.log 2>&1
echo "exit $?"
for f in $(find /tmp/prof_vendor -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -6 $f | cut -c1-170; done
timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/prof_vendor_pmc1 -o run -- python scripts/vendor_gemm_probe.py > $OUT/pmc1.log 2>&1
echo "pmc1 exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_vendor_pmc2 -o run -- python scripts/vendor_gemm_probe.py > $OUT/pmc2.log 2>&1
echo "pmc2 exit $?"
python scripts/summarize_pmc.py /tmp/prof_vendor_pmc1 /tmp/prof_vendor_pmc2 > $OUT/pmc_summary.txt 2>&1; grep -i "gemm\|Cijk" $OUT/pmc_summary.txt | cut -c1-260
