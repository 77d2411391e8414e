# interleaved same-box A/B of the weight-panel prefetch forms on the 512-row configurations (option gemm_prefetch: 0 off, 2 side stream, 3 riders)
for i in 1 2; do for v in ${PF_VALUES:-0 3}; do echo "gemm_prefetch=$v"; python scripts/bench_configs.py cfg1 cfg5 --opt gemm_prefetch=$v 2>&1 | grep -E "ms/NFE" | tail -4; done; done
