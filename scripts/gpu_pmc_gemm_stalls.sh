#!/bin/bash
# Where the persistent GEMM's wave cycles go: rocprofv3 PMC passes (wait / busy / instruction-class counters) over the op-level GEMM loop
# of the W1|W3 shape (scripts/opbench.py gemm ... variant 15), and the list of counters this box offers.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${PMC_OUT:-$R/gpurun_out/r6/pmc_stalls}; mkdir -p $OUT   # LUMINA_DIT_LIB / LT_PMC_SHAPE / LT_PMC_VENDOR select build, shape, vendor leg
cd $R
run() {  # tag, counters...
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o g -- python scripts/pmc_gemm_loop.py > $OUT/$tag.log 2>&1
  echo "$tag exit $?"
  python - "$tag" >> $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gemm" in k or "Cijk" in k:
        print(tag, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
  rm -rf /tmp/pmc_$tag
}
PASSES=${PMC_PASSES:-p1 p2 p3 p4 p5 p6 p7}
want() { [[ " $PASSES " == *" $1 "* ]]; }
want p1 && run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
want p2 && run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
want p3 && run p3 SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
want p4 && run p4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
want p5 && run p5 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
want p6 && run p6 GRBM_GUI_ACTIVE SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
want p7 && run p7 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
