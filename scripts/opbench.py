#!/usr/bin/env python
"""Operator-level A/B bench on the GPU box (not part of the product): times each hot kernel through the C ABI at
the cfg-2 shapes, variants interleaved in one process (guide 5.4 rule 24), random data (rule 25), median + min.

    python scripts/opbench.py gemm attn elem [--rounds 7]
"""
import argparse
import ctypes as C
import math
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import P, bf, lib, ok, rel_l2, set_option, stream  # noqa: E402


def timeit(fn, iters=5):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def ab(cases, rounds):
    """cases: {name: fn}; interleaved rounds -> {name: (median_ms, min_ms)}"""
    for fn in cases.values():
        fn()
    torch.cuda.synchronize()
    res = {k: [] for k in cases}
    for _ in range(rounds):
        for k, fn in cases.items():
            res[k].append(timeit(fn))
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


SHAPES_CFG2 = [("qkv", 8192, 6912, 2304, 0), ("wo", 8192, 2304, 2304, 0), ("w13", 8192, 12288, 2304, 1),
               ("w2", 8192, 2304, 6144, 0)]
# BASELINE cfg 1 (Next-DiT-ImageNet 600M, 256 tokens, cond + null row): 512 rows - the small-M regime
# BASELINE cfg 3 (Flag-DiT 5B at 1024^2): 2 x 4160 = 8320 rows = 32.5 row tiles of 256
SHAPES_CFG3 = [("qkv", 8320, 9216, 3072, 0), ("wo", 8320, 3072, 3072, 0), ("w13", 8320, 16384, 3072, 1), ("w2", 8320, 3072, 8192, 0)]
SHAPES_CFG1 = [("qkv", 512, 4608, 1536, 0), ("wo", 512, 1536, 1536, 0), ("w13", 512, 8192, 1536, 1), ("w2", 512, 1536, 4096, 0)]


def bench_gemm(rounds, variants, zeros=False, shapes=SHAPES_CFG2, cold=0):
    """cold > 0: rotate through `cold` different weight buffers (> 256 MiB of MALL in total) so every launch streams its
    weights from HBM like the engine does (one GEMM per layer, 16-32 layers of distinct weights)."""
    L = lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {v: 0.0 for v in variants}
    for name, M, N, K, epi in shapes:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
        if zeros:  # power probe: same instruction stream, no operand toggling (clock is power-managed under MFMA load)
            A.zero_()
            W.zero_()
        Ws = [W] + [W.clone() for _ in range(max(0, cold - 1))]
        rot = [0]
        outs = {}
        cases = {}
        for v in variants:
            out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=torch.bfloat16)
            outs[v] = out

            def fn(v=v, out=out):
                rot[0] = (rot[0] + 1) % len(Ws)
                W = Ws[rot[0]]  # "3t0" = variant 3 with the ping-pong tail overlap switched off
                # "7p3" = variant 7 under gemm_pipeline 3 (single-barrier loop)
                if isinstance(v, str) and "p" in v:
                    vi, pipe, tail = int(v.split("p")[0]), int(v.split("p")[1]), 0
                else:
                    vi, tail = (int(v[:-2]), int(v[-1])) if isinstance(v, str) and "t" in v else (int(v), 0)
                    pipe = 0
                assert pipe == 0 and tail == 0, "gemm_pipeline / gemm_pp_tail were round-1 knobs (csrc/experimental, explicit variants)"
                if vi == 18 and epi == 1:
                    vi = 17  # 64 x 128 deep-ring tiles have no SwiGLU form either
                if vi == 16 and epi == 1:
                    vi = 15  # the 288-wide tile has no SwiGLU form (32-column pairing): time the 256-wide one in its place
                ok(L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(out), M, N, K, epi, vi, stream()), "gemm")
            cases[v] = fn
        r = ab(cases, rounds)
        ref = outs[variants[0]].float()
        fl = 2.0 * M * N * K
        for v in variants:
            med, mn = r[v]
            err = rel_l2(outs[v], ref) if v != variants[0] else 0.0
            tot[v] += med
            print(f"gemm {name:4s} M{M} N{N} K{K} epi{epi} variant {v:>3}: median {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF/s"
                  f"  (best {fl/mn/1e9:7.1f})  rel-vs-v{variants[0]} {err:.2e}", flush=True)
    fl_layer = sum(2.0 * M * N * K for _, M, N, K, _ in shapes)
    for v in variants:
        print(f"gemm per-layer total variant {v}: {tot[v]*1e3:8.1f} us  -> {fl_layer/tot[v]/1e9:7.1f} TF/s")


def bench_gemm_vendor(rounds):
    """The same four GEMM shapes and random operands through the vendor library (torch.matmul -> hipBLASLt / rocBLAS), next to
    the engine's default kernels: where the practical ceiling of this part sits for bf16 GEMMs on non-trivial data.  Plain
    GEMMs only (the SwiGLU GEMM is timed without its epilogue on both sides here)."""
    L = lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, M, N, K, _ in SHAPES_CFG2:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        Wt = W.t()

        def ours():
            ok(L.lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(out), M, N, K, 0, 0, stream()))

        def vendor():
            torch.matmul(A, Wt, out=out2)

        def vendor_linear():
            torch.nn.functional.linear(A, W)

        r = ab({"engine (auto)": ours, "torch.matmul(A, W^T)": vendor, "F.linear(A, W)": vendor_linear}, rounds)
        fl = 2.0 * M * N * K
        for k, (med, mn) in r.items():
            print(f"vendor-cmp {name:4s} M{M} N{N} K{K} {k:22s}: median {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF/s (best {fl/mn/1e9:7.1f})", flush=True)
        print(f"   max |engine - vendor| = {float((out.float() - out2.float()).abs().max()):.4f}", flush=True)


def bench_gemm_moe(rounds, variants, cold=0):
    """BASELINE cfg 5 expert GEMMs: 512 tokens x top-2 over 4 experts = 8 segments of 256 expert-sorted rows (two per expert,
    the worst case of the plan), each streaming its own expert's weights; dense GEMMs of the same shape for comparison."""
    L = lib()
    E, M, d, F_ = 4, 2048, 1536, 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    te = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], dtype=torch.int32, device="cuda")
    for name, N, K, epi in (("w13", 2 * F_, d, 1), ("w2", d, F_, 0)):
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W0 = (torch.randn(E, N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
        Ws = [W0] + [W0.clone() for _ in range(max(0, cold - 1))]
        rot = [0]

        def nextw():
            rot[0] = (rot[0] + 1) % len(Ws)
            return Ws[rot[0]]
        out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=torch.bfloat16)
        cases = {}
        for v in variants:
            cases[f"grouped v{v}"] = lambda v=v: ok(L.lt_op_gemm_grouped(P(A), P(nextw()), P(te), N * K, P(out), M, N, K, epi, v, stream()))
            cases[f"dense   v{v}"] = lambda v=v: ok(L.lt_op_gemm_bf16(P(A), P(nextw()), P(None), 1, P(out), M, N, K, epi, v, stream()))
        r = ab(cases, rounds)
        fl = 2.0 * M * N * K
        for k, (med, mn) in r.items():
            print(f"moe gemm {name} M{M} N{N} K{K} epi{epi} {k}: median {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF/s  "
                  f"weights {E * N * K * 2 / med / 1e9 if k.startswith('grouped') else N * K * 2 / med / 1e9:6.2f} TB/s", flush=True)


def bench_attn(rounds, variants, shape=(2, 32, 4096, 72)):
    L = lib()
    B, H, N, hd = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(B * N, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
    q = torch.empty(B, H, N, hd, device="cuda", dtype=torch.bfloat16)
    k = torch.empty_like(q)
    vt = torch.empty(B, H, hd, N, device="cuda", dtype=torch.bfloat16)
    ok(L.lt_op_qk_norm_rope(P(qkv), 3 * H * hd, 0, P(None), P(None), C.c_float(1e-5), P(q), B, N, H, hd, 0, P(None), 64, 1.0, stream()))
    ok(L.lt_op_qk_norm_rope(P(qkv), 3 * H * hd, H * hd, P(None), P(None), C.c_float(1e-5), P(k), B, N, H, hd, 0, P(None), 64, 1.0, stream()))
    ok(L.lt_op_v_transpose(P(qkv), 3 * H * hd, 2 * H * hd, P(vt), B, N, N, H, hd, stream()))
    scale = 1.0 / math.sqrt(hd)
    variants = [v for v in variants if not (hd == 96 and v == 2)]  # v2 needs a spare O^T row (hd % 32 != 0)
    outs, cases = {}, {}
    for v in variants:
        out = torch.empty(B, N, H * hd, device="cuda", dtype=torch.bfloat16)
        outs[v] = out

        def fn(v=v, out=out):
            set_option("attention_variant", v)
            ok(L.lt_op_attention(P(q), P(k), P(vt), None, P(out), P(None), 0, B, H, H, N, N, N, hd, C.c_float(scale), 0, stream()))
        cases[v] = fn
    r = ab(cases, rounds)
    fl = 4.0 * B * H * N * N * hd
    # fp32 reference for one head
    qh = q[0, 0].float()
    kh = k[0, 0].float()
    vh = qkv.view(B, N, 3, H, hd)[0, :, 2, 0].float()
    ref = torch.softmax(qh @ kh.t() * scale, -1) @ vh
    for v in variants:
        med, mn = r[v]
        err = rel_l2(outs[v].view(B, N, H, hd)[0, :, 0], ref)
        print(f"attn B{B} H{H} N{N} hd{hd} variant {v}: median {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF/s (best {fl/mn/1e9:7.1f})"
              f"  rel-L2 head0 vs fp32 {err:.2e}", flush=True)
    set_option("attention_variant", 4)


def bench_attn_vendor(rounds):
    """What does the library attention reach on this chip?  torch's scaled_dot_product_attention (flash / mem-efficient / math
    back ends as the build offers them) on the engine's shape - head_dim 72 as is and zero-padded to 80 / 96 / 128 (what a
    generic kernel would run), plus head_dim 64 and 128 at the same token count for scale - next to the engine's kernel.
    FLOPs are counted on the TRUE head_dim (72) for the padded runs, so the TF/s column is comparable with the engine's."""
    import torch.nn.functional as F

    L = lib()
    B, H, N = 2, 32, 4096
    g = torch.Generator(device="cuda").manual_seed(3)

    def sdpa_case(hd_true, hd_run):
        q = torch.randn(B, H, N, hd_true, device="cuda", generator=g).to(torch.bfloat16)
        k = torch.randn(B, H, N, hd_true, device="cuda", generator=g).to(torch.bfloat16)
        v = torch.randn(B, H, N, hd_true, device="cuda", generator=g).to(torch.bfloat16)
        if hd_run != hd_true:
            q, k, v = (F.pad(t, (0, hd_run - hd_true)) for t in (q, k, v))
        scale = 1.0 / math.sqrt(hd_true)
        return lambda: F.scaled_dot_product_attention(q, k, v, scale=scale)

    cases = {}
    for hd_true, hd_run in ((72, 72), (72, 80), (72, 96), (72, 128), (64, 64), (128, 128)):
        fn = sdpa_case(hd_true, hd_run)
        try:
            fn()
            torch.cuda.synchronize()
            cases[f"sdpa hd{hd_true} run as {hd_run}"] = (fn, hd_true)
        except Exception as exc:  # a back end may refuse a head_dim
            print(f"sdpa hd{hd_true} run as {hd_run}: not available ({type(exc).__name__}: {str(exc)[:80]})", flush=True)
    hd = 72
    qkv = torch.randn(B * N, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
    q = torch.empty(B, H, N, hd, device="cuda", dtype=torch.bfloat16)
    k = torch.empty_like(q)
    vt = torch.empty(B, H, hd, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(B, N, H * hd, device="cuda", dtype=torch.bfloat16)
    ok(L.lt_op_qk_norm_rope(P(qkv), 3 * H * hd, 0, P(None), P(None), C.c_float(1e-5), P(q), B, N, H, hd, 0, P(None), 64, 1.0, stream()))
    ok(L.lt_op_qk_norm_rope(P(qkv), 3 * H * hd, H * hd, P(None), P(None), C.c_float(1e-5), P(k), B, N, H, hd, 0, P(None), 64, 1.0, stream()))
    ok(L.lt_op_v_transpose(P(qkv), 3 * H * hd, 2 * H * hd, P(vt), B, N, N, H, hd, stream()))

    def ours():
        ok(L.lt_op_attention(P(q), P(k), P(vt), None, P(out), P(None), 0, B, H, H, N, N, N, hd, C.c_float(1.0 / math.sqrt(hd)), 0, stream()))
    cases["engine hd72 (attention_variant 3)"] = (ours, 72)
    r = ab({name: fn for name, (fn, _) in cases.items()}, rounds)
    for name, (_, hd_true) in cases.items():
        med, mn = r[name]
        fl = 4.0 * B * H * N * N * hd_true
        print(f"attn B{B} H{H} N{N} {name}: median {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF/s (best {fl/mn/1e9:7.1f})", flush=True)


def bench_elem(rounds):
    L = lib()
    B, N, d = 2, 4096, 2304
    M = B * N
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
    h = torch.empty_like(x)
    w = torch.ones(d, device="cuda", dtype=torch.bfloat16)
    mod = (0.1 * torch.randn(B, 4 * d, device="cuda", generator=g)).to(torch.bfloat16)
    qkv = torch.randn(M, 3 * d, device="cuda", generator=g).to(torch.bfloat16)
    q = torch.empty(B, 32, N, 72, device="cuda", dtype=torch.bfloat16)
    vt = torch.empty(B, 32, 72, N, device="cuda", dtype=torch.bfloat16)
    tab = torch.zeros(2, 384, 18, 2, device="cuda", dtype=torch.float32)
    ok(L.lt_op_rope_table_2d(P(tab), 384, 72, C.c_float(10000.0), C.c_float(1.0), stream()))

    def grn():
        ok(L.lt_op_gated_residual_norm(P(x), P(y), P(w), P(mod), 1, 1, P(w), P(mod[:, d:]), P(None), 1, 4 * d, P(h), B, N, d,
                                       C.c_float(1e-5), C.c_float(1e-6), 0, stream()))

    def grn_pre():  # engine form: gates / scales prepared once per NFE
        ok(L.lt_op_gated_residual_norm(P(x), P(y), P(w), P(mod), 1, 0, P(w), P(mod[:, d:]), P(None), 1, 4 * d, P(h), B, N, d,
                                       C.c_float(1e-5), C.c_float(1e-6), 1, stream()))

    def grn_pre_spec():  # the same call on the mode-specialised instantiation (norm_specialize knob; must be bit-identical)
        set_option("norm_specialize", 1)
        try:
            grn_pre()
        finally:
            set_option("norm_specialize", 0)

    def qkn():
        ok(L.lt_op_qk_norm_rope(P(qkv), 3 * d, 0, P(w), P(w), C.c_float(1e-5), P(q), B, N, 32, 72, 1, P(tab), 64, 1.0, stream()))

    def qkn_copy():  # no LayerNorm, no rotary: the pure layout change
        ok(L.lt_op_qk_norm_rope(P(qkv), 3 * d, 0, P(None), P(None), C.c_float(1e-5), P(q), B, N, 32, 72, 0, P(None), 64, 1.0, stream()))

    def qkn_ln():
        ok(L.lt_op_qk_norm_rope(P(qkv), 3 * d, 0, P(w), P(w), C.c_float(1e-5), P(q), B, N, 32, 72, 0, P(None), 64, 1.0, stream()))

    def qkn_rope():
        ok(L.lt_op_qk_norm_rope(P(qkv), 3 * d, 0, P(None), P(None), C.c_float(1e-5), P(q), B, N, 32, 72, 1, P(tab), 64, 1.0, stream()))

    def vtr():
        ok(L.lt_op_v_transpose(P(qkv), 3 * d, 2 * d, P(vt), B, N, N, 32, 72, stream()))

    r = ab({"gated_residual_norm": grn, "gated_residual_norm_pre": grn_pre, "gated_residual_norm_pre_spec": grn_pre_spec, "qk_norm_rope": qkn, "qk_copy_only": qkn_copy, "qk_ln_only": qkn_ln, "qk_rope_only": qkn_rope, "v_transpose": vtr}, rounds)
    bytes_ = {"gated_residual_norm": 4 * M * d * 2, "gated_residual_norm_pre": 4 * M * d * 2, "gated_residual_norm_pre_spec": 4 * M * d * 2, "qk_norm_rope": 2 * M * d * 2, "qk_copy_only": 2 * M * d * 2, "qk_ln_only": 2 * M * d * 2, "qk_rope_only": 2 * M * d * 2,
              "v_transpose": 2 * M * d * 2}
    for kname, (med, mn) in r.items():
        print(f"elem {kname:28s}: median {med*1e3:7.1f} us  {bytes_[kname]/med/1e9:6.2f} TB/s algorithmic", flush=True)


def bench_insitu(rounds):
    """Does an HBM-bound row kernel cost more right after an MFMA-bound GEMM (power-managed clocks) than on its own?
    time([gemm, grn] x n) - time([gemm] x n)  vs  time([grn] x n)."""
    L = lib()
    B, N, d, F_ = 2, 4096, 2304, 6144
    M = B * N
    g = torch.Generator(device="cuda").manual_seed(4)
    u = torch.randn(M, F_, device="cuda", generator=g).to(torch.bfloat16)
    w2 = (torch.randn(d, F_, device="cuda", generator=g) / math.sqrt(F_)).to(torch.bfloat16)
    x = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    h = torch.empty_like(x)
    w = torch.ones(d, device="cuda", dtype=torch.bfloat16)
    mod = (0.1 * torch.randn(B, 4 * d, device="cuda", generator=g)).to(torch.bfloat16)

    def gemm():
        ok(L.lt_op_gemm_bf16(P(u), P(w2), P(None), 1, P(y), M, d, F_, 0, 0, stream()))

    def grn():
        ok(L.lt_op_gated_residual_norm(P(x), P(y), P(w), P(mod), 1, 0, P(w), P(mod[:, d:]), P(None), 1, 4 * d, P(h), B, N, d,
                                       C.c_float(1e-5), C.c_float(1e-6), 1, stream()))

    def both():
        gemm()
        grn()

    r = ab({"gemm": gemm, "gemm+grn": both, "grn": grn}, rounds)
    print(f"insitu: gemm {r['gemm'][0]*1e3:.1f} us, gemm+grn {r['gemm+grn'][0]*1e3:.1f} us -> grn after gemm "
          f"{(r['gemm+grn'][0]-r['gemm'][0])*1e3:.1f} us ; grn alone {r['grn'][0]*1e3:.1f} us", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["gemm", "attn", "elem"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--gemm-variants", type=str, default="1,2,3,4")
    ap.add_argument("--attn-variants", type=str, default="1,2")
    ap.add_argument("--cold", type=int, default=0, help="gemm_small: rotate through this many weight copies (HBM-cold weights)")
    ap.add_argument("--gemm-zeros", action="store_true", help="all-zero operands (power / clock probe)")
    ap.add_argument("--gemm-stagger", type=int, default=0, help="4-wave GEMM kernels (variants 10, 13, 14): start-phase spread per XCD in "
                    "units of ~1024 cycles (lt_set_option gemm_stagger)")
    a = ap.parse_args()
    print("device:", torch.cuda.get_device_name(0), flush=True)
    if a.gemm_stagger:
        set_option("gemm_stagger", a.gemm_stagger)
        print("gemm_stagger =", a.gemm_stagger, flush=True)
    if "gemm" in a.what:
        bench_gemm(a.rounds, [v if ("t" in v or "p" in v) else int(v) for v in a.gemm_variants.split(",")], zeros=a.gemm_zeros)
    if "gemm_small" in a.what:
        bench_gemm(a.rounds, [v if ("t" in v or "p" in v) else int(v) for v in a.gemm_variants.split(",")], shapes=SHAPES_CFG1,
                   cold=a.cold)
    if "gemm_cfg3" in a.what:
        bench_gemm(a.rounds, [int(v) for v in a.gemm_variants.split(",")], shapes=SHAPES_CFG3)
    if "gemm_vendor" in a.what:
        bench_gemm_vendor(a.rounds)
    if "gemm_moe" in a.what:
        bench_gemm_moe(a.rounds, [int(v) for v in a.gemm_variants.split(",")], cold=a.cold)
    if "insitu" in a.what:
        bench_insitu(a.rounds)
    if "attn" in a.what:
        bench_attn(a.rounds, [int(v) for v in a.attn_variants.split(",")])
    if "gemm_probe" in a.what:  # per-tile fixed cost vs per-slab cost: same N, two K; plain vs SwiGLU epilogue on the same shape
        probe = [("n12k_k2304_e0", 8192, 12288, 2304, 0), ("n12k_k2304_e1", 8192, 12288, 2304, 1), ("n12k_k4608_e0", 8192, 12288, 4608, 0),
                 ("n12k_k4608_e1", 8192, 12288, 4608, 1), ("n6912_k2304", 8192, 6912, 2304, 0), ("n6912_k4608", 8192, 6912, 4608, 0),
                 ("n2304_k2304", 8192, 2304, 2304, 0), ("n2304_k4608", 8192, 2304, 4608, 0)]
        bench_gemm(a.rounds, [int(v) for v in a.gemm_variants.split(",")], shapes=probe)
    if "attn96" in a.what:  # cfg 3 (Flag-DiT 5B): 64 rows x 65 tokens, hd 96
        bench_attn(a.rounds, [int(v) for v in a.attn_variants.split(",")], shape=(2, 32, 4160, 96))
    if "gemm_b4" in a.what:  # cfg 1 at four image pairs per call (2048 rows)
        bench_gemm(a.rounds, [v if ("t" in v or "p" in v) else int(v) for v in a.gemm_variants.split(",")],
                   shapes=[(n, 4 * m, nn, k, e) for n, m, nn, k, e in SHAPES_CFG1], cold=a.cold)
    if "attn48" in a.what:  # cfg 5 at 1024^2 / cfg 1: the 600M models, hd 48
        bench_attn(a.rounds, [int(v) for v in a.attn_variants.split(",")], shape=(2, 32, 4096, 48))
        bench_attn(a.rounds, [int(v) for v in a.attn_variants.split(",")], shape=(2, 32, 256, 48))
    if "attn_vendor" in a.what:
        bench_attn_vendor(a.rounds)
    if "elem" in a.what:
        bench_elem(a.rounds)
