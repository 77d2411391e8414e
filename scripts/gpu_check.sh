#!/bin/bash
# One gpurun call worth of checks: op parity, model parity, smoke, short bench (+ optional A/B).  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== nproc $(nproc)"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; }
for f in test_gpu_ops test_gpu_model; do
  echo "== pytest $f"
  timeout 1500 python -m pytest tests/$f.py -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/$f.log 2>&1
  echo "exit $?"; tail -${TAILN:-30} gpurun_out/$f.log
done
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "exit $?"; tail -3 gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 > gpurun_out/bench.log 2>&1; echo "exit $?"; tail -2 gpurun_out/bench.log
for ab in ${BENCH_AB:-}; do
  echo "== bench A/B $ab"
  timeout 900 python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline $(echo $ab | tr ',' ' ') > gpurun_out/bench_$(echo $ab | tr -d ',-' ).log 2>&1; echo "exit $?"
  tail -1 gpurun_out/bench_$(echo $ab | tr -d ',-').log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','kernel_time_ms_per_step','attention_tflops_per_s')}, d['roofline']['achieved'])"
done
