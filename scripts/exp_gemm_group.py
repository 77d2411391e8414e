import sys, math, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/scripts")
from gpu_util import P, lib, ok, set_option, stream
import opbench
L = lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, epi in [("w13", 8192, 12288, 2304, 1), ("qk", 8192, 4608, 2304, 0), ("w2", 8192, 2304, 6144, 0), ("w13_cfg3", 8320, 16384, 3072, 1)]:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=torch.bfloat16)
    cases = {}
    for G in (1, 2, 4, 8, 16, 32):
        def fn(G=G):
            set_option("gemm_group", G)
            ok(L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(out), M, N, K, epi, 0, stream()))
        cases[G] = fn
    r = opbench.ab(cases, 5)
    fl = 2.0 * M * N * K
    print(name, {G: f"{med*1e3:.1f}us {fl/med/1e9:.0f}TF" for G, (med, mn) in r.items()}, flush=True)
