#!/usr/bin/env python
"""Diagnostics (GPU box): per-wave cycle breakdown of the ping-pong GEMM from its s_memtime-instrumented build."""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, lib, ok, set_option, stream  # noqa: E402

L = lib()
set_option("gemm_pp_tail", int(os.environ.get("LT_PP_TAIL", "1")))
print("gemm_pp_tail =", os.environ.get("LT_PP_TAIL", "1"))
NAMES_PP = ["ds_issue", "vm_wait", "lgkm_wait", "bar1", "mfma", "bar2"]
# variant 12 (4 waves, VGPR-staged): 8 MFMA + 8 fragment reads | vmcnt wait | 8 MFMA + 8 ds_write | lgkmcnt(0) | barrier | 16 MFMA half
NAMES_W4S = ["h0_reads", "vm_wait", "h0_writes", "lgkm_wait", "barrier", "h1"]
CASES = [(8192, 6912, 2304, 3, 8), (8192, 2304, 6144, 4, 12), (8192, 6912, 2304, 4, 12), (8192, 6912, 2304, 12, 4),
         (8192, 12288, 2304, 12, 4)]
if os.environ.get("LT_TRACE_VARIANT"):  # e.g. 5 = single-barrier rendezvous kernel, 256x256 tile; 12 = 4-wave VGPR-staged
    v = int(os.environ["LT_TRACE_VARIANT"])
    CASES = [(8192, 6912, 2304, v, 4 if v == 12 else 8)]
for (M, N, K, variant, nw) in CASES:
    names = NAMES_W4S if variant == 12 else NAMES_PP
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tr = torch.zeros(64, nw, 8, device="cuda", dtype=torch.int64)
    for _ in range(2):
        ok(L.lt_op_gemm_trace(P(A), P(W), P(Cc), M, N, K, variant, P(tr), stream()))
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    walls = {}
    for nm, fn in (("traced", lambda: L.lt_op_gemm_trace(P(A), P(W), P(Cc), M, N, K, variant, P(tr), stream())),
                   ("untraced", lambda: L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(Cc), M, N, K, 0, variant, stream())),
                   ("classic", lambda: L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(Cc), M, N, K, 0, 3 if variant == 12 else 2 - variant % 2, stream()))):
        for _ in range(3):
            fn()
        st.record()
        for _ in range(10):
            fn()
        en.record()
        torch.cuda.synchronize()
        walls[nm] = st.elapsed_time(en) / 10 * 1e3
    t = tr.cpu()
    bm, bn = (256, 256) if variant in (3, 5, 12) else (256, 288)
    tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    rounds = (tiles + 255) // 256
    v6, v7 = int(t[0, 0, 6]), int(t[0, 0, 7])
    ns, pro, loop, epi = v6 >> 32, v6 & 0xffffffff, v7 >> 20, v7 & 0xfffff
    blk_cyc = float(pro + loop + epi)
    print(f"== M{M} N{N} K{K} variant {variant}: slabs {ns} tiles {tiles} rounds {rounds}; wall us {walls}; wave 0 of one traced "
          f"block: prologue {pro} + main loop {loop} + epilogue {epi} cycles -> implied clock >= {rounds * blk_cyc / walls['traced'] / 1e3:.2f} GHz")
    for blk in range(0, min(4, t.shape[0])):
        if int(t[blk, 0, 6]) == 0:
            continue
        for w in range(nw):
            nsl = float(int(t[blk, w, 6]) >> 32)
            per = [float(t[blk, w, i]) / nsl for i in range(6)]
            print(f"  blk {blk*64+5:4d} wave {w:2d} grp {w//4}: " + " ".join(f"{n} {v:6.0f}" for n, v in zip(names, per)) + f" | step {sum(per):6.0f} cyc")
