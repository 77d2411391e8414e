#!/usr/bin/env python
"""a few launches of the engine's self-attention at the bench shapes (hd 72: 4096 tokens, hd 96: 4160 tokens), for PMC passes"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import _lib

L = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B, H, N, hd in ((2, 32, 4096, 72), (2, 32, 4160, 96)):
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(B, H, N, hd, device="cuda", generator=g).to(torch.bfloat16)
    k = (torch.randn(B, H, N, hd, device="cuda", generator=g) * (1.4427 / math.sqrt(hd))).to(torch.bfloat16)
    vt = torch.randn(B, H, hd, N, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(B, N, H * hd, device="cuda", dtype=torch.bfloat16)
    for _ in range(6):
        _lib.check(L.lt_op_attention(P(q), P(k), P(vt), None, P(out), P(None), 0, B, H, H, N, N, N, hd, C.c_float(1.0 / math.sqrt(hd)), 0, s), "attn")
torch.cuda.synchronize()
print("done")
