#!/usr/bin/env python
"""GPU box, not part of the judged bench line: ms/NFE of the OTHER BASELINE.json configs on one MI355X (bench.py reports
configs[1] only; these are the parity-test configurations, timed here so DESIGN.md can quote them).

    python scripts/bench_configs.py [cfg1] [cfg3] [cfg4] [cfg5] [--nfe 8] [--opt name=value ...]

cfg1  Next-DiT-ImageNet 600M, 256x256 (256 tokens), class-conditional CFG, Euler
cfg3  Flag-DiT 5B (lumina_t2i), 1024x1024 (64 x 65 tokens incl. eol), text T=128/8, CFG 4
cfg4  Next-DiT 2B GQA, 2048x2048 (16384 tokens), time-aware scaling 2.0 / watershed 0.3, proportional attention
cfg5  Next-DiT-MoE 600M "Both" (4 time + 4 space experts, top-2), 256x256
cfg5-1024  the same model at BASELINE configs[4]'s stated 1024x1024: 4096 tokens, 8192 rows, 16 384 routed rows per MoE FFN
Synthetic weights drawn on the GPU (SURVEY.md 8d statistics), inputs resident in HBM, wall clock around the sampler call.
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lumina_t2x_amd  # noqa: E402,F401
from bench import random_init_  # noqa: E402
from lumina_t2x_amd import _lib, models  # noqa: E402
from lumina_t2x_amd.transport import Sampler, create_transport  # noqa: E402
from lumina_t2x_amd.flops import flops_per_nfe  # noqa: E402  (the product's own algorithmic-work model; no oracle import)


def timed(model, z, nfe, warm, **kw):
    tr = create_transport("Linear", "velocity", None, None, None)
    shift = kw.pop("_shift", None)

    def run(n):
        fn = Sampler(tr).sample_ode(sampling_method="euler", num_steps=n + 1, time_shifting_factor=shift)
        return fn(z, model.forward_with_cfg, **kw)

    run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(nfe)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nfe
    assert torch.isfinite(out[-1].float()).all()
    return dt * 1e3


def text_inputs(T, dim, dev):
    g = torch.Generator(device="cuda").manual_seed(1)
    feats = torch.randn(2, T, dim, device=dev, generator=g).to(torch.bfloat16)
    mask = torch.ones(2, T, dtype=torch.int32, device=dev)
    mask[1, 8:] = 0
    return feats, mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["cfg1", "cfg5", "cfg4"])
    ap.add_argument("--nfe", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=1, help="cfg1 / cfg5: image pairs per call (rows = 2 x pairs <= 8; every kernel is "
                    "row-parallel, so batching amortises the latency-bound launches of a 256-token image)")
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    for opt in a.opt:
        k, v = opt.split("=")
        _lib.check(_lib.load().lt_set_option(k.encode(), int(v)))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(7)
    for which in a.which:
        torch.manual_seed(0)
        if which == "cfg1":
            with torch.device(dev):
                m = models.imagenet.DiT_Llama_600M_patch2(qk_norm=True).to(torch.bfloat16)
            random_init_(m, 0)
            z = torch.randn(a.pairs, 4, 32, 32, device=dev, generator=g).repeat(2, 1, 1, 1)  # fp32 state (BASELINE: fp32 for cfg 1)
            y = torch.tensor([207] * a.pairs + [1000] * a.pairs, device=dev)
            ms = timed(m.eval(), z, a.nfe, 2, y=y, cfg_scale=4.0)
            fl = flops_per_nfe(dim=1536, n_layers=16, n_heads=32, n_tokens=256, batch=2 * a.pairs)
            toks = 256 * a.pairs
        elif which == "cfg5":
            with torch.device(dev):
                m = models.moe.DiT_Llama_600M_patch2_Both(qk_norm=True).to(torch.bfloat16)
            random_init_(m, 0)
            z = torch.randn(a.pairs, 4, 32, 32, device=dev, generator=g).to(torch.bfloat16).repeat(2, 1, 1, 1)
            y = torch.tensor([207] * a.pairs + [1000] * a.pairs, device=dev)
            ms = timed(m.eval(), z, a.nfe, 2, y=y, cfg_scale=4.0)
            fl = flops_per_nfe(dim=1536, n_layers=16, n_heads=32, n_tokens=256, batch=2 * a.pairs, adaln_chunks=6, ffn_visits=4)  # 1.41e12 per pair
            toks = 256 * a.pairs
        elif which == "cfg5-1024":
            with torch.device(dev):
                m = models.moe.DiT_Llama_600M_patch2_Both(qk_norm=True).to(torch.bfloat16)
            random_init_(m, 0)
            z = torch.randn(1, 4, 128, 128, device=dev, generator=g).to(torch.bfloat16).repeat(2, 1, 1, 1)
            y = torch.tensor([207, 1000], device=dev)
            ms = timed(m.eval(), z, a.nfe, 2, y=y, cfg_scale=4.0)
            fl = flops_per_nfe(dim=1536, n_layers=16, n_heads=32, n_tokens=4096, batch=2, adaln_chunks=6, ffn_visits=4)
            toks = 4096
        elif which == "cfg4":
            with torch.device(dev):
                m = models.NextDiT_2B_GQA_patch2(qk_norm=True, cap_feat_dim=2048).to(torch.bfloat16)
            random_init_(m, 0)
            z = torch.randn(1, 4, 256, 256, device=dev, generator=g).to(torch.bfloat16).repeat(2, 1, 1, 1)
            feats, mask = text_inputs(128, 2048, dev)
            ms = timed(m.eval(), z, min(a.nfe, 4), 1, cap_feats=feats, cap_mask=mask, cfg_scale=4.0, proportional_attn=True,
                       base_seqlen=4096, scale_factor=2.0, scale_watershed=0.3, _shift=4)
            fl = flops_per_nfe(dim=2304, n_layers=24, n_heads=32, n_kv_heads=8, cap_feat_dim=2048, n_tokens=16384, text_len=128)
            toks = 16384
        elif which == "cfg3":
            with torch.device(dev):
                m = models.flag_dit.DiT_Llama_5B_patch2(qk_norm=True, cap_feat_dim=4096).to(torch.bfloat16)
            random_init_(m, 0)
            z = torch.randn(1, 4, 128, 128, device=dev, generator=g).to(torch.bfloat16).repeat(2, 1, 1, 1)
            feats, mask = text_inputs(128, 4096, dev)
            ms = timed(m.eval(), z, min(a.nfe, 4), 1, cap_feats=feats, cap_mask=mask, cfg_scale=4.0, proportional_attn=True,
                       base_seqlen=4096, _shift=4)
            fl = flops_per_nfe(dim=3072, n_layers=32, n_heads=32, cap_feat_dim=4096, n_tokens=4160, text_len=128, adaln_chunks=6)  # 75.6e12 at 4096 tokens
            toks = 4160
        else:
            raise SystemExit(f"unknown config {which}")
        if which in ("cfg1", "cfg5") and a.pairs > 1:
            which = f"{which} x{a.pairs} pairs ({ms / a.pairs:.3f} ms/NFE per pair)"
        # both roofs: MFMA = algorithmic FLOP / 2.5 PFLOP/s; HBM = the bf16 weights every NFE must stream / 8 TB/s
        wbytes = sum(p.numel() for p in m.parameters()) * 2
        print(f"{which}: {ms:9.3f} ms/NFE  {1e3 / ms:8.2f} NFE/s  {toks / ms * 1e3:12.0f} latent-tokens/s  "
              f"{fl / ms / 1e9:8.1f} model TFLOP/s = {fl / ms / 1e9 / 2500:.3f} of the MFMA peak; weights {wbytes / 1e9:.2f} GB -> "
              f"{wbytes / ms / 1e9:.2f} TB/s = {wbytes / ms / 1e9 / 8:.3f} of the HBM peak; ideal {max(fl / 2.5e15, wbytes / 8e12) * 1e3:.3f} ms "
              f"-> {max(fl / 2.5e15, wbytes / 8e12) * 1e3 / ms:.3f} of the bounding roof", flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
