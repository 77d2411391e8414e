#!/bin/bash
# wave-level wait / busy counters of the one-wave-per-SIMD attention kernels (hd 72, hd 96) - the attention twin of gpu_pmc_gemm_stalls.sh
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/pmc_attn; mkdir -p $OUT
cd $R
run() {
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmca_$tag -o g -- python scripts/pmc_attn_loop.py > $OUT/$tag.log 2>&1
  echo "$tag exit $?"
  python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmca_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "attn" in k:
        print(tag, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run p3 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run p4 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32
run p5 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_LDS
