# the other BASELINE configs at HEAD + the attention-prologue form of q_norm / RoPE at 16 384 tokens (cfg 4), same box
python scripts/bench_configs.py cfg1 cfg3 cfg4 cfg5 cfg5-1024 2>&1 | grep -E "ms/NFE"
for v in 0 1 0 1; do echo "attn_q_fused=$v"; python scripts/bench_configs.py cfg4 --nfe 4 --opt attn_q_fused=$v 2>&1 | grep -E "ms/NFE"; done
python scripts/bench_configs.py cfg1 cfg5 --pairs 4 2>&1 | grep -E "ms/NFE"
