#!/bin/bash
# ONE parameterised recipe script for a `gpurun` call (replaces the numbered lab-notebook scripts of round 2, which stay in git
# history).  Runs ON THE GPU BOX from the repo snapshot; everything it writes goes under gpurun_out/<round>/<tag>/.
#
#   scripts/gpu_call.sh <tag> <step> [<step> ...]
#
# steps (run in the order given):
#   tests[:<pytest -k expr>]   python -m pytest tests -m gpu -q -x [-k expr]          -> pytest_gpu.log
#   bench[:<extra args>]       python bench.py <extra>                                 -> bench.json (+ one-line digest)
#   ab:<opt>=<v0>,<v1>[,...]   interleaved same-box A/B of one lt_set_option knob, 2 rounds  -> ab_<opt>.log
#   ablib:<path>[,<path>...]   interleaved A/B of the default and other builds of the library (LUMINA_DIT_LIB)  -> ab_lib.log
#   configs:<names>            scripts/bench_configs.py <names: cfg1 cfg3 cfg4 cfg5, space separated by '+'>
#   opbench:<args>             scripts/opbench.py <args with '+' for spaces>
#   prof                       rocprofv3 --kernel-trace --stats of bench.py + two --pmc passes (scripts/gpu_prof.sh)
#   power:<variants>           scripts/power_probe.py <variants separated by '+'>: watts and J/TFLOP of GEMM loops
#   sh:<command>               anything else ('+' for spaces)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${LT_ROUND:-r5}
TAG=$1; shift
OUT=$R/gpurun_out/$ROUND/$TAG
mkdir -p "$OUT"
cd "$R"
digest() {  # one line per bench JSON
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "no JSON line:", e); sys.exit(0)
r = d["roofline"]
kt = {k: round(v, 2) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}
pw = d.get("power") or {}
print(sys.argv[2], {k: round(d[k], 2) for k in ("value", "ms_per_step")}, kt, "gemm TF/s", round(r["achieved"], 1), "attn TF/s",
      round(d["attention_tflops_per_s"], 1), "W", round(pw.get("avg_w") or 0, 0), "J/TF", round(pw.get("joule_per_tflop") or 0, 3),
      "MHz", round(pw.get("gfx_clock_mhz_avg") or 0))
PY
}
for step in "$@"; do
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}; arg=${arg//+/ }
  echo "=== $step"
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout 2400 python -m pytest tests -m gpu -q -x -s -k "$arg" > $OUT/pytest_gpu.log 2>&1
      else timeout 2400 python -m pytest tests -m gpu -q -x -s > $OUT/pytest_gpu.log 2>&1; fi
      echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log; grep -E "engine vs reference|routing" $OUT/pytest_gpu.log | cut -c1-330 ;;
    bench)
      n=$(ls $OUT/bench*.json 2>/dev/null | wc -l)
      timeout 900 python bench.py $arg > $OUT/bench$n.json 2> $OUT/bench$n.err; echo "bench exit $?"; digest $OUT/bench$n.json "bench $arg" ;;
    ab)
      opt=${arg%%=*}; vals=${arg#*=}
      for i in 1 2; do for v in ${vals//,/ }; do
        timeout 600 python bench.py --no-cpu-baseline --opt $opt=$v > $OUT/ab.tmp 2>/dev/null; digest $OUT/ab.tmp "$opt=$v" | tee -a $OUT/ab_$opt.log
      done; done ;;
    ablib)
      for i in 1 2; do for lib in "" ${arg//,/ }; do
        LUMINA_DIT_LIB=$lib timeout 600 python bench.py --no-cpu-baseline > $OUT/ab.tmp 2>/dev/null; digest $OUT/ab.tmp "lib=${lib:-default}" | tee -a $OUT/ab_lib.log
      done; done ;;
    configs) timeout 1200 python scripts/bench_configs.py $arg > $OUT/bench_configs.log 2>&1; echo "exit $?"; tail -12 $OUT/bench_configs.log ;;
    opbench) timeout 1200 python scripts/opbench.py $arg > $OUT/opbench_$(echo $arg | tr ' /' '__' | cut -c1-60).log 2>&1; echo "exit $?"; tail -30 $OUT/opbench_*.log | cut -c1-220 ;;
    prof) PROF_OUT=$OUT/prof bash scripts/gpu_prof.sh 2>&1 | tail -60 ;;
    power) timeout 900 python scripts/power_probe.py $arg > $OUT/power_probe.log 2>&1; echo "exit $?"; cat $OUT/power_probe.log | cut -c1-220 ;;
    sh) bash -c "$arg" > $OUT/sh_$(date +%s).log 2>&1; echo "exit $?"; tail -30 $OUT/sh_*.log | cut -c1-220 ;;
    *) echo "unknown step $step" ;;
  esac
done
du -sh $OUT
