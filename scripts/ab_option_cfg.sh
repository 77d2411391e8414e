# interleaved same-box A/B of one engine option on the small configurations:  OPT=attn_small_fused VALUES="0 1" CFGS="cfg1 cfg5" bash scripts/ab_option_cfg.sh
for i in 1 2; do for v in ${VALUES:-0 1}; do echo "${OPT}=$v"; python scripts/bench_configs.py ${CFGS:-cfg1 cfg5} --opt ${OPT}=$v 2>&1 | grep -E "ms/NFE" | tail -4; done; done
