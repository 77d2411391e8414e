# interleaved same-box A/B of the default build against another build of the library on the small configurations:
#   ALT=lumina-t2x_amd/lib/alt1/liblumina_dit.so CFGS="cfg1 cfg5" bash scripts/ab_lib_cfg.sh
for i in 1 2; do for lib in "" "${ALT}"; do echo "lib=${lib:-default}"; LUMINA_DIT_LIB=$lib python scripts/bench_configs.py ${CFGS:-cfg1 cfg5} 2>&1 | grep -E "ms/NFE" | tail -4; done; done
