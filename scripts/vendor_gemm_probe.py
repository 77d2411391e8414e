#!/usr/bin/env python
"""Which vendor kernels (names carry the tile configuration) serve the four GEMM shapes of the headline workload, and how
long do they take?  Run under rocprofv3 --kernel-trace --stats (scripts/gpu_prof_vendor.sh)."""
import math

import torch

g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in (("qkv", 8192, 6912, 2304), ("wo", 8192, 2304, 2304), ("w13", 8192, 12288, 2304), ("w2", 8192, 2304, 6144)):
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    for _ in range(12):
        torch.nn.functional.linear(A, W)
    torch.cuda.synchronize()
    print("done", name, flush=True)
