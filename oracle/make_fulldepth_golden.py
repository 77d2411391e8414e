"""Full-depth, full-width fixtures for the BASELINE configs (tests/golden/full_*.npz) - TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_fulldepth_golden [case ...]      (authoring container: needs /root/reference, ~62 GB RAM)

Each fixture holds, for one seeded (weights, inputs) draw of a COMPLETE model (every layer, every width of the BASELINE
config, reduced token count where the judge's CPU budget asks for it):
  ref     forward_with_cfg of the UNMODIFIED reference module (imported from /root/reference behind oracle/stubs), CPU fp32
  oracle  the same call through the functional restatement (oracle/*.py), fp32
  floor   the restatement with the reference's bf16 rounding points switched on: its distance to `ref` is the
          reference's own bf16-vs-fp32 noise floor on this draw, the yardstick the GPU tests gate against (1.5 x floor)
Weights and inputs are NOT stored: oracle.synth regenerates them from the seeds (numpy PCG64); `wsum` / `wprobe` are
checksums of the draw so that a silent RNG mismatch is detected instead of showing up as a parity failure.
The -m "not gpu" suite pins oracle to ref on these (tests/test_oracle_golden.py); the -m gpu suite runs the engine on the
regenerated draw and compares with ref / floor (tests/test_gpu_fulldepth.py) - no CPU oracle time on the GPU box.
"""
import gc
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import nextdit_oracle as O  # noqa: E402
from oracle import ref_harness as R  # noqa: E402
from oracle import synth  # noqa: E402
from oracle import variants_oracle as V  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")

# name -> (config, package, module, class, latent_hw, text_len, uncond_len, seed_w, seed_x, [(tag, t_value, kwargs)])
CASES = {
    # BASELINE configs[1]: Lumina-Next-T2I 2B at 1024^2 (the bench workload), all 24 layers, 4096 tokens
    "full_2b": dict(cfg=synth.NEXT_2B, pkg="lumina_next_t2i", module="models.model", cls="NextDiT", latent_hw=(128, 128), text_len=128,
                    uncond_len=8, seed_w=61, seed_x=62,
                    calls=[("cfg4", 0.5, dict(cfg_scale=4.0, base_seqlen=4096, proportional_attn=True))]),
    # BASELINE configs[3]: Lumina-Next-SFT 2B (GQA 32/8), time-aware RoPE scaling (scale_factor 2, watershed 0.3): both branches.
    # The fp32 reference at the config's 16 384 tokens needs an explicit [B, H, N, N] mask (> 62 GB); 4096 tokens as a 32 x 128
    # patch grid keeps the config's RoPE range (column positions 0..127 under NTK / linear scaling) and its 16-tile key loops
    "full_2b_gqa_ntk": dict(cfg=synth.NextDiTConfig(n_kv_heads=8), pkg="lumina_next_t2i_mini", module="models.nextdit", cls="NextDiT",
                            latent_hw=(64, 256), text_len=128, uncond_len=8, seed_w=71, seed_x=72,
                            calls=[("ntk", 0.6, dict(cfg_scale=4.0, scale_factor=2.0, scale_watershed=0.3, base_seqlen=4096, proportional_attn=True)),
                                   ("lin", 0.1, dict(cfg_scale=4.0, scale_factor=2.0, scale_watershed=0.3, base_seqlen=4096, proportional_attn=True))]),
    # BASELINE configs[2]: Lumina-T2I 5B (Flag-DiT, 32 layers, d 3072, hd 96) at the config's own 1024^2: 64 rows x 65 tokens incl. eol
    "full_flag5b": dict(cfg=synth.FLAG_5B, pkg="lumina_t2i", module="models.model", cls="DiT_Llama", latent_hw=(128, 128), text_len=128,
                        uncond_len=8, seed_w=81, seed_x=82,
                        calls=[("cfg4", 0.5, dict(cfg_scale=4.0, base_seqlen=4096, proportional_attn=True))]),
    # BASELINE configs[4]: Next-DiT-MoE 600M "Both" (4 time + 4 space experts per block, 16 layers), 1024 tokens
    "full_moe600m": dict(cfg=synth.NextDiTConfig(dim=1536, n_layers=16, n_heads=32, family="moe"), pkg="Next-DiT-MoE", module="models.models2",
                         cls="DiT_Llama", latent_hw=(64, 64), text_len=0, uncond_len=0, seed_w=91, seed_x=92,
                         calls=[("cfg4", 0.5, dict(cfg_scale=4.0))]),
    # the same model at the config's own 256^2 (latent 32 x 32 -> 256 tokens, 512 rows with the CFG pair): the small-M kernels
    "full_moe600m_256": dict(cfg=synth.NextDiTConfig(dim=1536, n_layers=16, n_heads=32, family="moe"), pkg="Next-DiT-MoE", module="models.models2",
                             cls="DiT_Llama", latent_hw=(32, 32), text_len=0, uncond_len=0, seed_w=91, seed_x=93,
                             calls=[("cfg4", 0.5, dict(cfg_scale=4.0))]),
    # BASELINE configs[4] AS WORDED: "Next-DiT-MoE, 1024x1024" - latent 128 x 128 -> 4096 tokens, 8192 rows with the CFG pair, top-2 of
    # 4 + 4 experts = 16 384 routed rows per MoE layer: the MFMA-bound grouped-GEMM regime (round 4).  Same weights.
    "full_moe600m_4096": dict(cfg=synth.NextDiTConfig(dim=1536, n_layers=16, n_heads=32, family="moe"), pkg="Next-DiT-MoE", module="models.models2",
                              cls="DiT_Llama", latent_hw=(128, 128), text_len=0, uncond_len=0, seed_w=91, seed_x=94,
                              calls=[("cfg4", 0.5, dict(cfg_scale=4.0))]),
    # BASELINE configs[0]: Next-DiT-ImageNet 600M, class-conditional 256^2 (latent 32 x 32 -> 256 tokens), all 16 layers, pinned to the
    # unmodified Next-DiT-ImageNet/models/models.py (a13 had only the live oracle at full depth)
    "full_imagenet600m": dict(cfg=synth.IMAGENET_600M, pkg="Next-DiT-ImageNet", module="models.models", cls="DiT_Llama", latent_hw=(32, 32),
                              text_len=0, uncond_len=0, seed_w=51, seed_x=52, calls=[("cfg4", 0.5, dict(cfg_scale=4.0))]),
    # BASELINE configs[3] AT ITS OWN SIZE: 2048^2 -> latent 256 x 256 -> 128 x 128 patches = 16 384 tokens, 24 layers, GQA 32 / 8, both
    # branches of the time-aware RoPE (positions 0..127 in BOTH axes, 256 key tiles per query block).  The reference module cannot
    # run this on the authoring host (its SDPA fallback materialises an fp32 [B, H, N, N] mask: 2 x 32 x 16384^2 x 4 B = 69 GB,
    # lumina_next_t2i_mini/models/nextdit.py:358-361), so `reference: False`: the fixture holds the RESTATEMENT's outputs - the same code,
    # unchanged, that equals the reference bit for bit on full_2b_gqa_ntk (same weights: seed_w 71) - and says so in `pinned_by`.
    # The oracle's self-attention passes no mask to SDPA (the sampling mask is all ones) and therefore takes the CPU flash path:
    # no N x N tensor, no query chunking needed.
    "full_2b_gqa_16k": dict(cfg=synth.NextDiTConfig(n_kv_heads=8), pkg="lumina_next_t2i_mini", module="models.nextdit", cls="NextDiT",
                            latent_hw=(256, 256), text_len=128, uncond_len=8, seed_w=71, seed_x=73, reference=False,
                            calls=[("ntk", 0.6, dict(cfg_scale=4.0, scale_factor=2.0, scale_watershed=0.3, base_seqlen=4096, proportional_attn=True)),
                                   ("lin", 0.1, dict(cfg_scale=4.0, scale_factor=2.0, scale_watershed=0.3, base_seqlen=4096, proportional_attn=True))]),
}


def weight_checksum(sd):
    """(float64 sum of |w| over three tensors spread through the draw, first 8 values of the last matrix)"""
    keys = list(sd.keys())
    pick = [keys[0], keys[len(keys) // 2], keys[-1]]
    wsum = np.array([float(sd[k].double().abs().sum()) for k in pick], dtype=np.float64)
    mats = [k for k in keys if sd[k].ndim == 2]
    return wsum, sd[mats[-1]].flatten()[:8].double().numpy().copy(), pick + [mats[-1]]


def oracle_call(cfg, sd, ins, kw, bf16, moe_hook=None):
    if cfg.family == "next_t2i":
        z, t, cap, mask = ins
        return O.forward_with_cfg(sd, cfg, z, t, cap, mask, bf16=bf16, **kw)
    if cfg.family == "flag_t2i":
        z, t, cap, mask = ins
        return V.flag_forward_with_cfg(sd, cfg, z, t, cap, mask, bf16=bf16, **kw)
    z, t, y = ins
    return V.imagenet_forward_with_cfg(sd, cfg, z, t, y, bf16=bf16, moe_hook=moe_hook, **kw)


def _fresh_import(pkg, module):
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    root = os.path.join(R.REFERENCE_ROOT, pkg)
    sys.path = [q for q in sys.path if not q.startswith(R.REFERENCE_ROOT)]
    stubs = os.path.join(HERE, "stubs")
    if stubs not in sys.path:
        sys.path.insert(0, stubs)
    sys.path.insert(0, root)
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # model.py:952 hard-codes .cuda()
    return importlib.import_module(module)


def reference_calls(case, sd, make_inputs):
    os.environ["TORCHDYNAMO_DISABLE"] = "1"
    cfg = case["cfg"]
    mod = _fresh_import(case["pkg"], case["module"])
    cls = getattr(mod, case["cls"])
    kw = cfg.ctor_kwargs()
    if "use_flash_attn" in cls.__init__.__code__.co_varnames:
        kw["use_flash_attn"] = False
    t0 = time.time()
    model = cls(**kw).eval()
    res = model.load_state_dict(sd, strict=True, assign=True)  # assign: the module aliases the draw instead of copying 8-20 GB
    assert not res.missing_keys and not res.unexpected_keys
    gc.collect()
    print(f"  reference {case['pkg']}:{case['cls']} built in {time.time() - t0:.0f} s", flush=True)
    outs = {}
    with torch.no_grad():
        for tag, tv, ckw in case["calls"]:
            ins = make_inputs(tv)
            t0 = time.time()
            ckw = dict(ckw)
            scale = ckw.pop("cfg_scale")
            if cfg.has_text:
                z, t, cap, mask = ins
                outs[tag] = model.forward_with_cfg(z, t, cap, mask, scale, **ckw).float().numpy().copy()
            else:
                z, t, y = ins
                outs[tag] = model.forward_with_cfg(z, t, y, scale, **ckw).float().numpy().copy()
            print(f"  reference call {tag}: {time.time() - t0:.0f} s", flush=True)
    del model
    gc.collect()
    return outs


def _record_routing(model, n_layers):
    """forward hooks on the routers of the reference's TimeMoeLayer / SpaceMoeLayer (Next-DiT-MoE/models/models2.py:459-506): the top-k
    ids each MoE layer selects, in oracle.variants_oracle.MoeRouting's table format ([L][time, space][rows][k], ascending per row)."""
    rec, hooks = {}, []
    for name, m in model.named_modules():
        kind = type(m).__name__
        if kind not in ("TimeMoeLayer", "SpaceMoeLayer"):
            continue
        layer = int(name.split(".")[1])
        where = (layer, 0 if kind == "TimeMoeLayer" else 1)

        def hook(mod, args, out, where=where, k=m.num_experts_per_tok):
            rec[where] = torch.sort(torch.topk(out, k).indices, dim=1).values.to(torch.int32).numpy().copy()

        hooks.append(m.gate.register_forward_hook(hook))
    return rec, hooks


def reference_bf16_calls(case, sd, make_inputs):
    """The yardstick's pin: the UNMODIFIED reference module moved to bfloat16 - `model.eval().to(dtype)` as
    lumina_next_t2i/sample.py:129 does - with flash_attn replaced by oracle/stubs/flash_attn (fp32-accumulate SDPA, one cast back),
    run once plain and once under torch.autocast (sample.py:173 wraps the sampling loop in it; on CPU: autocast("cpu", bfloat16)).
    The mini fork's `use_flash_attn=True` branch (lumina_next_t2i_mini/models/nextdit.py:328-357) never builds the N x N mask, so the
    16 384-token case runs here too.  Returns {tag: (plain, autocast, routing-or-None)}."""
    os.environ["TORCHDYNAMO_DISABLE"] = "1"
    cfg = case["cfg"]
    mod = _fresh_import(case["pkg"], case["module"])
    cls = getattr(mod, case["cls"])
    kw = cfg.ctor_kwargs()
    if "use_flash_attn" in cls.__init__.__code__.co_varnames:
        kw["use_flash_attn"] = True
    model = cls(**kw).eval()
    res = model.load_state_dict(sd, strict=True, assign=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(torch.bfloat16)   # new bf16 parameters; the fp32 draw `sd` is untouched
    gc.collect()
    outs = {}
    with torch.no_grad():
        for tag, tv, ckw in case["calls"]:
            got = []
            routing = None
            for ac in (False, True):
                ins = make_inputs(tv)
                ckw2 = dict(ckw)
                scale = ckw2.pop("cfg_scale")
                rec, hooks = _record_routing(model, cfg.n_layers)
                t0 = time.time()
                with torch.autocast("cpu", torch.bfloat16, enabled=ac):
                    if cfg.has_text:
                        z, t, cap, mask = ins
                        o = model.forward_with_cfg(z.to(torch.bfloat16), t, cap.to(torch.bfloat16), mask, scale, **ckw2)
                    else:
                        z, t, y = ins
                        o = model.forward_with_cfg(z.to(torch.bfloat16), t, y, scale, **ckw2)
                for h in hooks:
                    h.remove()
                assert o.dtype == torch.bfloat16, o.dtype
                got.append(o.float().numpy().copy())
                if rec and not ac:
                    rows = max(v.shape[0] for v in rec.values())
                    routing = np.full((cfg.n_layers, 2, rows, 2), -1, dtype=np.int32)
                    for (l, b), v in rec.items():
                        routing[l, b] = v if v.shape[0] == rows else np.repeat(v, rows // v.shape[0], axis=0)  # time router: one row per sample
                print(f"  reference in bf16 ({'autocast' if ac else 'plain'}) call {tag}: {time.time() - t0:.0f} s", flush=True)
            outs[tag] = (got[0], got[1], routing)
    del model
    gc.collect()
    return outs


def add_refbf16(name):
    """`--refbf16 case`: adds refbf16_* / refbf16ac_* (+ refbf16_agree_* for MoE) to an EXISTING fixture; nothing else is recomputed."""
    case = CASES[name]
    cfg = case["cfg"]
    path = os.path.join(OUT, f"{name}.npz")
    old = dict(np.load(path, allow_pickle=False))
    sd = synth.synth_state_dict(cfg, seed=case["seed_w"], streams=True)
    wsum, wprobe, _ = weight_checksum(sd)
    assert np.array_equal(wsum, old["wsum"]) and np.array_equal(wprobe, old["wprobe"]), "the draw is not the fixture's"

    def make_inputs(tv):
        if cfg.has_text:
            ins = list(synth.synth_inputs(cfg, latent_hw=case["latent_hw"], text_len=case["text_len"], uncond_len=case["uncond_len"],
                                          seed=case["seed_x"], t_value=tv))
            ins[2] = ins[2].to(torch.bfloat16).float()
        else:
            ins = list(synth.synth_inputs(cfg, latent_hw=case["latent_hw"], seed=case["seed_x"], t_value=tv))
        ins[0] = ins[0].to(torch.bfloat16).float()
        return tuple(ins)

    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
    for tag, (plain, ac, routing) in reference_bf16_calls(case, sd, make_inputs).items():
        old[f"refbf16_{tag}"], old[f"refbf16ac_{tag}"] = plain, ac
        base = old[f"ref_{tag}"] if f"ref_{tag}" in old else old[f"oracle_{tag}"]
        fl = old[f"floor_{tag}"]
        print(f"[{name}] {tag}: vs {'reference' if f'ref_{tag}' in old else 'restatement'} fp32: reference-in-bf16 {rel(plain, base):.3e} "
              f"(autocast {rel(ac, base):.3e}; ch3 {rel(plain[:, 3], base[:, 3]):.3e} / {rel(ac[:, 3], base[:, 3]):.3e}), "
              f"floor {rel(fl, base):.3e} (ch3 {rel(fl[:, 3], base[:, 3]):.3e}); floor vs reference-in-bf16 {rel(fl, plain):.3e}", flush=True)
        if routing is not None:
            table = old[f"route_{tag}"].astype(np.int32)
            ran = table >= 0
            agree = float((((routing == table) | ~ran).all(axis=-1)).mean())
            old[f"refbf16_agree_{tag}"] = np.float64(agree)
            print(f"[{name}] {tag}: the reference module in bf16 selects the fp32 run's experts on {agree * 100:.2f} % of (layer, branch, "
                  f"row) slots; the bf16 choreography: {float(old[f'floor_agree_{tag}']) * 100:.2f} %", flush=True)
    if not case.get("reference", True):
        old["pinned_by"] = np.array("refbf16_*: output of the UNMODIFIED reference module in bf16 (flash branch, oracle/stubs/flash_attn) at this "
                                    "size; oracle_* / floor_*: the restatement, pinned bit for bit on the same weights at 4096 tokens by "
                                    "full_2b_gqa_ntk (the fp32 reference cannot run this size)")
    np.savez_compressed(path, **old)
    print(f"[{name}] refbf16 added", flush=True)


def run_case(name):
    case = CASES[name]
    cfg = case["cfg"]
    print(f"[{name}] drawing weights ...", flush=True)
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=case["seed_w"], streams=True)
    nparam = sum(v.numel() for v in sd.values())
    print(f"[{name}] {nparam / 1e9:.2f} B parameters in {time.time() - t0:.0f} s", flush=True)
    wsum, wprobe, wkeys = weight_checksum(sd)

    def make_inputs(tv):
        if cfg.has_text:
            return synth.synth_inputs(cfg, latent_hw=case["latent_hw"], text_len=case["text_len"], uncond_len=case["uncond_len"],
                                      seed=case["seed_x"], t_value=tv)
        return synth.synth_inputs(cfg, latent_hw=case["latent_hw"], seed=case["seed_x"], t_value=tv)

    out = {"config": np.array(json.dumps(cfg.to_dict())), "seed_w": case["seed_w"], "seed_x": case["seed_x"],
           "latent_hw": np.array(case["latent_hw"]), "text_len": case["text_len"], "uncond_len": case["uncond_len"],
           "package": np.array(case["pkg"]), "wsum": wsum, "wprobe": wprobe, "wkeys": np.array(json.dumps(wkeys)),
           "calls": np.array(json.dumps([[tag, tv, kw] for tag, tv, kw in case["calls"]]))}
    # the engine's inputs are bf16: round z / cap once so that reference, oracle and engine all see identical values
    def rounded(ins):
        ins = list(ins)
        ins[0] = ins[0].to(torch.bfloat16).float()
        if cfg.has_text:
            ins[2] = ins[2].to(torch.bfloat16).float()
        return tuple(ins)

    run_ref = R.available() and case.get("reference", True)
    out["pinned_by"] = np.array("reference module output stored as ref_*" if case.get("reference", True) else
                                "restatement only (reference cannot run this size); the restatement is pinned bit for bit on the same "
                                "weights at 4096 tokens by full_2b_gqa_ntk")
    ref = reference_calls(case, sd, lambda tv: rounded(make_inputs(tv))) if run_ref else {}
    with torch.no_grad():
        for tag, tv, ckw in case["calls"]:
            ins = rounded(make_inputs(tv))
            t0 = time.time()
            want = oracle_call(cfg, sd, ins, ckw, False)
            t1 = time.time()
            floor = oracle_call(cfg, sd, ins, ckw, True)
            print(f"[{name}] {tag}: oracle fp32 {t1 - t0:.0f} s, bf16-choreography {time.time() - t1:.0f} s", flush=True)
            out[f"oracle_{tag}"] = want.float().numpy()
            out[f"floor_{tag}"] = floor.float().numpy()
            if cfg.family.startswith("moe"):
                # routing-pinned yardstick: the experts the fp32 run selects (the restatement equals the
                # reference module bit for bit on this draw, so these ARE the reference's selections), the bf16 choreography
                # re-run with that discrete choice held equal, and how often the free bf16 run agreed with it
                rec32 = V.MoeRouting(cfg.n_layers)
                again = oracle_call(cfg, sd, ins, ckw, False, rec32)
                assert torch.equal(again, want)
                table = rec32.table()
                rec16 = V.MoeRouting(cfg.n_layers)
                free16 = oracle_call(cfg, sd, ins, ckw, True, rec16)
                assert torch.equal(free16.float(), floor.float())
                forced = oracle_call(cfg, sd, ins, ckw, True, V.MoeRouting(cfg.n_layers, force=table))
                t16 = rec16.table()
                ran = table >= 0
                agree = float((((t16 == table) | ~ran).all(axis=-1)).mean())
                assert table.max() < 128
                out[f"route_{tag}"] = table.astype(np.int8)
                out[f"floor_forced_{tag}"] = forced.float().numpy()
                out[f"floor_agree_{tag}"] = np.float64(agree)
                r = torch.from_numpy(ref[tag]) if tag in ref else want
                relf = float((forced.float() - r).norm() / r.norm())
                print(f"[{name}] {tag}: bf16 choreography with the fp32 routing forced vs reference {relf:.3e}; free bf16 routing agrees "
                      f"with fp32 on {agree * 100:.2f} % of (layer, branch, row) selections", flush=True)
            if tag in ref:
                out[f"ref_{tag}"] = ref[tag]
                r = torch.from_numpy(ref[tag])
                rel = lambda a, b: float((a - b).norm() / b.norm())
                print(f"[{name}] {tag}: oracle vs reference {rel(want, r):.3e}; bf16 floor vs reference {rel(floor.float(), r):.3e} "
                      f"(unguided channel {rel(floor.float()[:, 3], r[:, 3]):.3e})", flush=True)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"[{name}] written", flush=True)
    del sd
    gc.collect()


def main():
    torch.set_grad_enabled(False)
    args = sys.argv[1:]
    if args and args[0] == "--refbf16":
        for n in args[1:]:
            add_refbf16(n)
        return
    for n in args or list(CASES):
        run_case(n)


if __name__ == "__main__":
    main()
