#!/usr/bin/env python
"""Probe (not part of the product): what would split-K buy the 512-row GEMMs?  A split-K workgroup does exactly what a workgroup of
the problem (M, s N, K / s) does - same tile, same bytes staged per workgroup, s times the workgroups - so timing that problem on
the round-1 small tiles (HBM-cold weights) prices split-K before any of it is built (the fp32 partial traffic comes on top)."""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, lib, ok, stream  # noqa: E402
from opbench import ab  # noqa: E402

L = lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, v, splits in (("qkv", 512, 4608, 1536, 7, (1, 2, 4)), ("wo", 512, 1536, 1536, 8, (1, 2, 4)), ("w2", 512, 1536, 4096, 8, (1, 2, 4, 8))):
    for s in splits:
        n, k = N * s, K // s
        A = torch.randn(M, k, device="cuda", generator=g).to(torch.bfloat16)
        Ws = [(torch.randn(n, k, device="cuda", generator=g) / math.sqrt(k)).to(torch.bfloat16) for _ in range(max(2, int(400e6 / (n * k * 2))))]
        out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        rot = [0]

        def fn():
            rot[0] = (rot[0] + 1) % len(Ws)
            ok(L.lt_op_gemm_bf16(P(A), P(Ws[rot[0]]), P(None), 1, P(out), M, n, k, 0, v, stream()))
        med, mn = ab({"x": fn}, 5)["x"]
        tiles = ((M + (127 if v == 7 else 63)) // (128 if v == 7 else 64)) * ((n + 127) // 128)
        print(f"{name:4s} split {s}: as M{M} N{n} K{k} variant {v} ({tiles} workgroups): median {med * 1e3:7.1f} us (best {mn * 1e3:6.1f})", flush=True)
