#!/usr/bin/env python
"""a few launches of the engine's W1|W3 GEMM (plain epilogue) and of the vendor GEMM on the same operands, for PMC passes"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import _lib
import ctypes as C

lib = _lib.load()
# LT_PMC_SHAPE="M,N,K,variant" (default: the W1|W3 shape on 256-wide tiles, variant 15; the O projection: 8192,2304,2304,16); LT_PMC_VENDOR=0 skips hipBLASLt
M, N, K, VAR = (int(v) for v in os.environ.get("LT_PMC_SHAPE", "8192,12288,2304,15").split(","))
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
PAIR = os.environ.get("LT_PMC_PAIR", "0") == "1"  # round 6: A and W in the row-pair-interleaved layout (whole-line requests), same kernel
if PAIR:
    _lib.check(lib.lt_op_pair_layout(C.c_void_p(A.data_ptr()), M, K, 1, C.c_void_p(s)), "pair A")
    _lib.check(lib.lt_op_pair_layout(C.c_void_p(W.data_ptr()), N, K, 1, C.c_void_p(s)), "pair W")
for _ in range(6 if PAIR else 0):
    _lib.check(lib.lt_op_gemm_bf16_pair(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), M, N, K, 0, 0, C.c_void_p(s)), "lt_op_gemm_bf16_pair")
if PAIR:
    _lib.check(lib.lt_op_pair_layout(C.c_void_p(A.data_ptr()), M, K, 0, C.c_void_p(s)), "unpair A")
    _lib.check(lib.lt_op_pair_layout(C.c_void_p(W.data_ptr()), N, K, 0, C.c_void_p(s)), "unpair W")
for _ in range(0 if PAIR else 6):
    _lib.check(lib.lt_op_gemm_bf16(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(None), 1, C.c_void_p(out.data_ptr()), M, N, K, 0, VAR, C.c_void_p(s)), "lt_op_gemm_bf16")
for _ in range(6 if os.environ.get("LT_PMC_VENDOR", "1") == "1" else 0):
    torch.matmul(A, W.t(), out=out)
torch.cuda.synchronize()
print("done")
