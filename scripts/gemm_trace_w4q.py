#!/usr/bin/env python
"""Diagnostics (GPU box, `make EXPERIMENTAL=1` build): where a launch of the persistent 16x16x32 GEMM spends its time.

Per workgroup the trace build records s_memrealtime (100 MHz) at entry / prologue done / last main loop done / last epilogue
issued / exit, and the shader clocks of the first tile's main loop.  Printed: the launch's duration by events, the span from the
first entry to the last exit, and the distribution over workgroups of each phase - dispatch skew, prologue, main loop per slab,
epilogue issue, store drain, and how long finished workgroups wait for the slowest one."""
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, lib, ok, stream  # noqa: E402

L = lib()
CASES = [("wo", 8192, 2304, 2304, 16), ("w2", 8192, 2304, 6144, 16), ("qk", 8192, 4608, 2304, 16), ("w13 plain", 8192, 12288, 2304, 15),
         ("w13 swiglu", 8192, 12288, 2304, 115)]


def pct(x):
    return " ".join(f"{np.percentile(x, q):7.2f}" for q in (0, 10, 50, 90, 100))


for name, M, N, K, variant in CASES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tr = torch.zeros(256, 8, device="cuda", dtype=torch.int64)
    call = lambda: ok(L.lt_op_gemm_trace(P(A), P(W), P(Cc), M, N, K, variant, P(tr), stream()))
    plain = lambda: ok(L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(Cc), M, N, K, 1 if variant >= 100 else 0, variant % 100, stream()))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    walls = {}
    for nm, fn in (("traced", call), ("product", plain)):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s, e in evs:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        walls[nm] = float(np.median([s.elapsed_time(e) for s, e in evs])) * 1e3
    call()
    torch.cuda.synchronize()
    t = tr.cpu().numpy().astype(np.float64)
    live = t[:, 7] > 0
    t = t[live]
    us = lambda c: c / 100.0  # 100 MHz ticks -> us
    t0 = t[:, 0].min()
    entry, pro, loop, epi, ex = (us(t[:, i] - t0) for i in range(5))
    tiles = t[:, 7]
    ns = K // 32
    span = ex.max()
    clk = t[:, 5] / (us(t[:, 2] - t[:, 1]) / tiles) / 1e3 if True else 0  # shader clocks of tile 0's loop / its wall share
    print(f"== {name}: M{M} N{N} K{K} variant {variant}: {int(live.sum())} workgroups x {int(tiles.max())} tiles, {ns} slabs per tile; "
          f"event duration traced {walls['traced']:.1f} us, product {walls['product']:.1f} us; first entry -> last exit {span:.1f} us")
    print(f"   percentiles over workgroups            min     p10     p50     p90     max")
    print(f"   entry after first entry        us: {pct(entry)}")
    print(f"   prologue (entry -> loop start)  us: {pct(pro - entry)}")
    print(f"   all tiles' loops + epilogues    us: {pct(loop - pro)}   (per slab: {np.median((loop - pro) / (tiles * ns)):.3f} us)")
    print(f"   last epilogue issue             us: {pct(epi - loop)}")
    print(f"   store drain (issue -> exit)     us: {pct(ex - epi)}")
    print(f"   exit time                       us: {pct(ex)}")
    print(f"   idle after exit (span - exit)   us: {pct(span - ex)}   mean {np.mean(span - ex):.2f}")
    print(f"   shader clock in tile 0's loop  GHz: {pct(t[:, 5] / ((loop - pro) / tiles * 1e3))}  (upper bound when tiles > 1: the loop share includes epilogues)")
    xcc = (t[:, 6].astype(np.int64) >> 32) & 0xf
    per = []
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        per.append(f"xcc{x}: n {int(m.sum())} loop {np.median((loop - pro)[m]):.1f} us clk {np.median(t[m, 5] / ((loop - pro)[m] / tiles[m] * 1e3)):.2f} exit {np.median(ex[m]):.1f}")
    print("   per XCD (median): " + " | ".join(per))
