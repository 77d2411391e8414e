#!/usr/bin/env python
"""Diagnostics (GPU box): where a launch of attention variant 4 (hd 72, one wave per SIMD) spends its time - per workgroup
s_memrealtime stamps at entry / loop start / loop end / exit and the shader clocks of the tile loop."""
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, lib, ok, set_option, stream  # noqa: E402

L = lib()
VARIANT = int(os.environ.get("LT_ATTN_VARIANT", "4"))  # 4, or 5 (PV on 16x16x32 MFMAs: 1280 matrix-pipe cycles per tile)
set_option("attention_variant", VARIANT)
B, H, N, hd = 2, 32, 4096, 72
q = torch.randn(B, H, N, hd, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, N, hd, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, hd, N, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * hd, device="cuda", dtype=torch.bfloat16)
nwg = B * H * (N // 256)
tr = torch.zeros(nwg, 8, device="cuda", dtype=torch.int64)
sc = 1 / math.sqrt(hd)
call = lambda: ok(L.lt_op_attention_trace(P(q), P(k), P(vt), P(out), B, H, H, N, N, N, hd, sc, P(tr), stream()))
for _ in range(5):
    call()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, e in evs:
    s.record(); call(); e.record()
torch.cuda.synchronize()
wall = float(np.median([s.elapsed_time(e) for s, e in evs])) * 1e3
t = tr.cpu().numpy().astype(np.float64)
t = t[t[:, 5] > 0]  # persistent kernel: one record per workgroup (its first item's phases, its own exit)
t0 = t[:, 0].min()
us = lambda c: c / 100.0
pct = lambda x: " ".join(f"{np.percentile(x, p):8.2f}" for p in (0, 10, 50, 90, 100))
ntile = t[:, 5]
loop = us(t[:, 2] - t[:, 1])
print(f"attention variant {VARIANT} B{B} H{H} N{N} hd{hd}: {nwg} items on {len(t)} persistent workgroups, event duration {wall:.1f} us, first entry -> last exit {us(t[:, 3].max() - t0):.1f} us")
print("   percentiles over workgroups          min      p10      p50      p90      max")
print(f"   entry after first entry       us: {pct(us(t[:, 0] - t0))}")
print(f"   prologue (entry -> loop)      us: {pct(us(t[:, 1] - t[:, 0]))}")
print(f"   tile loop                     us: {pct(loop)}   per tile {np.median(loop / ntile):.3f} us")
print(f"   first loop end -> exit        us: {pct(us(t[:, 3] - t[:, 2]))}   (the remaining items of the workgroup)")
print(f"   whole workgroup               us: {pct(us(t[:, 3] - t[:, 0]))}")
print(f"   shader clocks per tile          : {pct(t[:, 4] / ntile)}   (44 MFMAs = 1408 matrix-pipe cycles)")
print(f"   shader clock                 GHz: {pct(t[:, 4] / (loop * 1e3))}")
xcc = t[:, 6].astype(np.int64) & 0xf
print("   per XCD (median): " + " | ".join(f"xcc{x}: loop {np.median(loop[xcc == x]):.1f} us clk {np.median((t[:, 4] / (loop * 1e3))[xcc == x]):.2f}" for x in sorted(set(xcc.tolist()))))
