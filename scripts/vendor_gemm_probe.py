#!/usr/bin/env python
"""Which vendor kernels (names carry the tile configuration) serve the four GEMM shapes of the headline workload, how long do
they take, and - under the PMC passes of scripts/gpu_prof_vendor.sh - how do their LDS / fabric / matrix-pipe counters compare
with the engine's kernels on the widest shape (W1||W3: M 8192, N 12288, K 2304)?"""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, lib, ok, stream  # noqa: E402

L = lib()
g = torch.Generator(device="cuda").manual_seed(0)
shapes = (("qkv", 8192, 6912, 2304), ("wo", 8192, 2304, 2304), ("w13", 8192, 12288, 2304), ("w2", 8192, 2304, 6144))
if os.environ.get("PROBE_W13_ONLY"):
    shapes = shapes[2:3]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(6):
        torch.nn.functional.linear(A, W)
    # classic 256x256 (8 waves), ping-pong 256x256 (8 waves), persistent 4 waves x (128 x 128) on 32x32x16 MFMAs, persistent 4 waves on
    # 16x16x32 MFMAs (the engine's kernel) - all in the product library (round 1 probed the experimental variants 10 / 12 here)
    for variant in (1, 3, 14, 15):
        for _ in range(6):
            ok(L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(out), M, N, K, 0, variant, stream()))
    torch.cuda.synchronize()
    print("done", name, flush=True)
