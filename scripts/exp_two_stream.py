#!/usr/bin/env python
"""Feasibility probe (not part of the product): would running the two samples of a CFG pair on two HIP streams, half a phase apart,
let the HBM-bound row kernels of one sample hide under the MFMA kernels of the other - and does the power the row phase leaves
unused come back as clock to the other stream's GEMM?  (every MFMA kernel sits at the board's power cap, the row
kernels do not.)

Sequence per sample and "layer": gated_residual_norm (4096 rows) -> O-shaped GEMM (4096 x 2304 x 2304 = 128 tiles = 128 workgroups:
half the chip) -> gated_residual_norm -> W2-shaped GEMM (4096 x 2304 x 6144, 128 workgroups).
  A  one stream, both samples per kernel (8192 rows, 256 workgroups): the engine's structure
  B  two streams, one sample each, started together (lock step)
  C  two streams, stream 1 delayed by one row kernel + half a GEMM (phases interleave)
"""
import ctypes as C
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from bench import PowerSampler  # noqa: E402
from gpu_util import P, lib, ok  # noqa: E402

d, F, N = 2304, 6144, 4096
L = lib()


def sp(s):
    return C.c_void_p(s.cuda_stream)


def make(rows, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    B = rows // N
    return dict(x=r(rows, d), y=r(rows, d), h=torch.empty(rows, d, device="cuda", dtype=torch.bfloat16), u=r(rows, F),
                w=(1 + 0.02 * torch.randn(d, device="cuda", generator=g)).to(torch.bfloat16),
                mod=(0.1 * torch.randn(B, 4 * d, device="cuda", generator=g)).to(torch.bfloat16), B=B, rows=rows,
                wo=(torch.randn(d, d, device="cuda", generator=g) / math.sqrt(d)).to(torch.bfloat16),
                w2=(torch.randn(d, F, device="cuda", generator=g) / math.sqrt(F)).to(torch.bfloat16))


def layer(t, s):
    ok(L.lt_op_gated_residual_norm(P(t["x"]), P(t["y"]), P(t["w"]), P(t["mod"]), 1, 0, P(t["w"]), P(t["mod"][:, d:]), P(None), 1, 4 * d,
                                   P(t["h"]), t["B"], N, d, C.c_float(1e-5), C.c_float(1e-6), 1, sp(s)))
    ok(L.lt_op_gemm_bf16(P(t["h"]), P(t["wo"]), P(None), 1, P(t["y"]), t["rows"], d, d, 0, 0, sp(s)))
    ok(L.lt_op_gated_residual_norm(P(t["x"]), P(t["y"]), P(t["w"]), P(t["mod"]), 1, 0, P(t["w"]), P(t["mod"][:, d:]), P(None), 1, 4 * d,
                                   P(t["h"]), t["B"], N, d, C.c_float(1e-5), C.c_float(1e-6), 1, sp(s)))
    ok(L.lt_op_gemm_bf16(P(t["u"]), P(t["w2"]), P(None), 1, P(t["y"]), t["rows"], d, F, 0, 0, sp(s)))


def run(mode, layers=96):
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    both, a, b = make(2 * N, 1), make(N, 2), make(N, 3)

    def go():
        if mode == "A":
            for _ in range(layers):
                layer(both, s0)
        else:
            if mode == "C":  # stagger: stream 1 starts with half a layer of its own
                ok(L.lt_op_gated_residual_norm(P(b["x"]), P(b["y"]), P(b["w"]), P(b["mod"]), 1, 0, P(b["w"]), P(b["mod"][:, d:]), P(None), 1, 4 * d,
                                               P(b["h"]), 1, N, d, C.c_float(1e-5), C.c_float(1e-6), 1, sp(s1)))
                ok(L.lt_op_gemm_bf16(P(b["h"]), P(b["wo"]), P(None), 1, P(b["y"]), N, d, d, 0, 0, sp(s1)))
            for _ in range(layers):
                layer(a, s0)
                layer(b, s1)
    go()
    torch.cuda.synchronize()
    ps = PowerSampler(0)
    with ps:
        t0 = time.perf_counter()
        go()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    r = ps.report(dt, 1.0) or {}
    fl = layers * 2 * (2.0 * N * d * d + 2.0 * N * d * F) / 1e12
    print(f"mode {mode}: {dt / layers * 1e3:8.3f} ms per layer pair  ({fl / dt:7.1f} TF/s over the whole sequence)  {r.get('avg_w', 0):6.0f} W avg  "
          f"gfx {r.get('gfx_clock_mhz_avg', 0):5.0f} MHz", flush=True)


if __name__ == "__main__":
    print(__doc__.split("Sequence")[0].strip().splitlines()[0])
    for rnd in range(2):
        for mode in ("A", "B", "C"):
            run(mode)
