#!/usr/bin/env python
"""Watts and joules per TFLOP of the GEMM kernels (not part of the product): VERDICT r2 asked for the "power-bound" claim of
DESIGN.md 5.1 to be measured in watts, not inferred from clocks.  Each case runs ONE kernel back to back for ~1.5 s on the
W1|W3 shape of cfg 2 (8192 x 12288 x 2304, plain epilogue so that the vendor library runs the same work) while a host thread
samples the socket power (bench.PowerSampler: amdsmi, 5 ms period); cases are interleaved twice.

    python scripts/power_probe.py [w4q] [w4p] [classic] [pingpong] [vendor] [zeros] [attn] [norm] [idle]
"""
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from bench import PowerSampler  # noqa: E402
from gpu_util import P, lib, ok, stream  # noqa: E402

M, N, K = 8192, 12288, 2304


def loop(fn, seconds=1.5):
    """run fn back to back for ~seconds; returns (launches, wall seconds, PowerSampler)"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        fn()
    en.record()
    torch.cuda.synchronize()
    per = st.elapsed_time(en) / 20 * 1e-3
    n = max(50, int(seconds / per))
    ps = PowerSampler(0)
    with ps:
        time.sleep(0.05)
        ps.samples.clear(); ps.clocks.clear()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return n, dt, ps


def main():
    want = sys.argv[1:] or ["idle", "w4q", "pingpong", "classic", "vendor", "zeros", "attn", "norm"]  # w4p: EXPERIMENTAL=1 builds only
    L = lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    A0, W0 = torch.zeros_like(A), torch.zeros_like(W)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    Wt = W.t()

    def gemm(variant, a=A, w=W):
        return lambda: ok(L.lt_op_gemm_bf16(P(a), P(w), P(None), 1, P(out), M, N, K, 0, variant, stream()), "gemm")

    cases = {}
    fl = 2.0 * M * N * K / 1e12
    if "w4q" in want: cases["w4q 16x16x32 persistent (variant 15, product default)"] = (gemm(15), fl)
    if "w4p" in want: cases["w4p 32x32x16 persistent (variant 13)"] = (gemm(13), fl)
    if "pingpong" in want: cases["8-wave ping-pong (variant 3)"] = (gemm(3), fl)
    if "classic" in want: cases["12-wave classic (variant 1)"] = (gemm(1), fl)
    if "vendor" in want: cases["hipBLASLt via torch.matmul"] = (lambda: torch.matmul(A, Wt, out=out), fl)
    if "zeros" in want: cases["w4q, all-zero operands"] = (gemm(15, A0, W0), fl)
    if "attn" in want:
        B, H, Nq, hd = 2, 32, 4096, 72
        q = torch.randn(B, H, Nq, hd, device="cuda", generator=g).to(torch.bfloat16)
        k = (torch.randn(B, H, Nq, hd, device="cuda", generator=g) * 0.2).to(torch.bfloat16)
        vt = torch.randn(B, H, hd, Nq, device="cuda", generator=g).to(torch.bfloat16)
        o = torch.empty(B, Nq, H * hd, device="cuda", dtype=torch.bfloat16)
        cases["attention hd 72 (default variant), 4096 keys"] = (
            lambda: ok(L.lt_op_attention(P(q), P(k), P(vt), P(None), P(o), P(None), 0, B, H, H, Nq, Nq, Nq, hd, 1.0, 1, stream()), "attn"),
            4.0 * B * H * Nq * Nq * hd / 1e12)
    if "norm" in want:
        x = torch.randn(M, 2304, device="cuda", generator=g).to(torch.bfloat16)
        y = torch.empty_like(x)
        cases["HBM copy 2 x 37.7 MB (torch)"] = (lambda: y.copy_(x), 0.0)
    print(f"# power probe: {M} x {N} x {K} bf16 GEMM (plain epilogue), one kernel back to back, socket power sampled every 5 ms")
    if "idle" in want:
        ps = PowerSampler(0)
        with ps:
            time.sleep(1.0)
        r = ps.report(1.0, 0.0)
        print(f"idle: {r['avg_w']:.0f} W (source: {r['source']})" if r else "idle: no power source readable")
    for rnd in range(2):
        for name, (fn, tf) in cases.items():
            n, dt, ps = loop(fn)
            r = ps.report(dt, tf * n) or {}
            tfs = tf * n / dt
            print(f"round {rnd} {name:58s} {dt / n * 1e6:8.1f} us  {tfs:7.1f} TF/s  {r.get('avg_w', 0):6.0f} W avg {r.get('max_w', 0):6.0f} max "
                  f"{(r.get('joule_per_tflop') or 0):6.3f} J/TFLOP  gfx {r.get('gfx_clock_mhz_avg', 0):5.0f} MHz  ({r.get('samples', 0)} samples)", flush=True)


if __name__ == "__main__":
    main()
