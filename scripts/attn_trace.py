#!/usr/bin/env python
"""Diagnostics (GPU box): per-wave cycle breakdown of the hd-72 ping-pong attention kernel (s_memtime build)."""
import ctypes as C
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, lib, ok, set_option, stream  # noqa: E402

L = lib()
set_option("attention_variant", 3)
B, H, N, hd = 2, 32, 4096, 72
q = torch.randn(B, H, N, hd, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, N, hd, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, hd, N, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * hd, device="cuda", dtype=torch.bfloat16)
tr = torch.zeros(64, 8, 8, device="cuda", dtype=torch.int64)
sc = 1 / math.sqrt(hd)
for _ in range(2):
    ok(L.lt_op_attention_trace(P(q), P(k), P(vt), P(out), B, H, H, N, N, N, hd, sc, P(tr), stream()))
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(5):
    ok(L.lt_op_attention_trace(P(q), P(k), P(vt), P(out), B, H, H, N, N, N, hd, sc, P(tr), stream()))
en.record()
torch.cuda.synchronize()
wall = st.elapsed_time(en) / 5 * 1e3
t = tr.cpu()
names = ["X", "vmwait", "bar1", "Y", "bar2"]
tot = None
for blk in range(4):
    for w in range(8):
        nt = float(t[blk, w, 5])
        if nt == 0:
            continue
        per = [float(t[blk, w, i]) / nt for i in range(5)]
        tot = sum(per)
        print(f"blk {blk*64+5:4d} wave {w} grp {w//4}: " + " ".join(f"{n} {v:6.0f}" for n, v in zip(names, per)) + f" | tile {tot:6.0f} cyc")
rounds = (B * H * (N // 256) + 255) // 256
print(f"wall {wall:.1f} us traced; {rounds} rounds x 64 tiles x {tot:.0f} cyc -> implied clock >= {rounds*64*tot/wall/1e3:.2f} GHz")
