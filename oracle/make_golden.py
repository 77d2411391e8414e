"""Generates tests/golden/*.npz by running the UNMODIFIED reference (behind oracle/stubs) in the authoring
container:  python -m oracle.make_golden        (needs /root/reference; TEST INFRASTRUCTURE ONLY)

Fixtures store seeds + config + inputs + the reference's outputs; weights are regenerated from the seed by
oracle.synth (numpy PCG64, bit-reproducible for the pinned numpy).  See tests/test_oracle_golden.py.
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_harness as R  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def kats(models_mod, transport_mod):
    """known answers of the reference's own helper functions"""
    M = importlib.import_module("models.model")
    integ = importlib.import_module("transport.integrators")
    d = {}
    o = integ.ode(drift=None, t0=0, t1=1, sampler_type="euler", num_steps=30, atol=1e-6, rtol=1e-3, time_shifting_factor=4)
    d["tgrid_30_shift4"] = _np(o.t)
    o = integ.ode(drift=None, t0=0, t1=1, sampler_type="euler", num_steps=5, atol=1e-6, rtol=1e-3)
    d["tgrid_5"] = _np(o.t)
    d["rope72_default"] = _np(torch.view_as_real(M.NextDiT.precompute_freqs_cis(72, 16)))
    d["rope72_ntk2"] = _np(torch.view_as_real(M.NextDiT.precompute_freqs_cis(72, 16, scale_factor=2.0, scale_watershed=0.3, timestep=0.5)))
    d["rope72_lin2"] = _np(torch.view_as_real(M.NextDiT.precompute_freqs_cis(72, 16, scale_factor=2.0, scale_watershed=0.3, timestep=0.1)))
    x = (torch.arange(16, dtype=torch.float32) / 10).view(1, 2, 1, 8)
    f = M.NextDiT.precompute_freqs_cis(8, 4)[:1, :2].flatten(0, 1).unsqueeze(0)
    d["rotary_in"] = _np(x)
    d["rotary_out"] = _np(M.Attention.apply_rotary_emb(x, f))
    d["temb_half"] = _np(M.ParallelTimestepEmbedder.timestep_embedding(torch.tensor([0.5, 0.037, 1.0]), 256))
    tr = transport_mod.create_transport("Linear", "velocity", None, None, None)
    d["interval"] = np.array(tr.check_interval(tr.train_eps, tr.sample_eps, sde=False, eval=True, reverse=False, last_step_size=0.0), dtype=np.float64)
    for meth in ("euler", "midpoint", "rk4"):
        fn = transport_mod.Sampler(tr).sample_ode(sampling_method=meth, num_steps=5)
        d[f"wiring_{meth}"] = _np(fn(torch.ones(2, 4, 8, 8), lambda x, t, **kw: -x))
    ffn = {}
    for dim in (1536, 2304, 3072, 4096, 576):
        ffn[str(dim)] = M.FeedForward(dim, 4 * dim, 256, None).w1.weight.shape[0]
    d["ffn_widths"] = np.array(json.dumps(ffn))
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **d)
    print("kat.npz:", sorted(d))


def model_case(name, cfg, pkg, latent_hw, text_len, uncond_len, seed_w, seed_x):
    sd = synth.synth_state_dict(cfg, seed=seed_w)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=latent_hw, text_len=text_len, uncond_len=uncond_len, seed=seed_x)
    if pkg == "lumina_next_t2i":
        model = R.build_reference_model(cfg, sd)
        transport_mod = importlib.import_module("transport")
    else:
        for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
            del sys.modules[m]
        sys.path.insert(0, os.path.join(R.REFERENCE_ROOT, pkg))
        nd = importlib.import_module("models.nextdit")
        model = nd.NextDiT(use_flash_attn=False, **cfg.ctor_kwargs()).eval() if "use_flash_attn" in nd.NextDiT.__init__.__code__.co_varnames \
            else nd.NextDiT(**cfg.ctor_kwargs()).eval()
        model.load_state_dict(sd, strict=True)
        transport_mod = None
    N = (latent_hw[0] // 2) * (latent_hw[1] // 2)
    out = {"config": np.array(json.dumps(cfg.to_dict())), "seed_w": seed_w, "seed_x": seed_x,
           "z": _np(z), "t": _np(t), "cap": _np(cap), "mask": _np(mask), "package": np.array(pkg)}
    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(_np(o))) for l in model.layers]
    with torch.no_grad():
        out["forward"] = _np(model(z, t, cap, mask))  # before any forward_with_cfg: default RoPE table
        out["hidden"] = np.stack(hidden)
        for h in hooks:
            h.remove()
        out["cfg4_prop"] = _np(model.forward_with_cfg(z, t, cap, mask, 4.0, scale_factor=1.0, scale_watershed=1.0,
                                                      base_seqlen=16, proportional_attn=True))
        t_lo = torch.full((2,), 0.1)
        out["cfg4_lin2"] = _np(model.forward_with_cfg(z, t_lo, cap, mask, 4.0, scale_factor=2.0, scale_watershed=0.3,
                                                      base_seqlen=16, proportional_attn=True))
        t_hi = torch.full((2,), 0.8)
        out["cfg4_ntk2"] = _np(model.forward_with_cfg(z, t_hi, cap, mask, 4.0, scale_factor=2.0, scale_watershed=0.3,
                                                      base_seqlen=16, proportional_attn=True))
        out["cfg1_plain"] = _np(model.forward_with_cfg(z, t, cap, mask, 1.0))
        if transport_mod is not None:
            tr = transport_mod.create_transport("Linear", "velocity", None, None, None)
            for meth in ("euler", "midpoint"):
                fn = transport_mod.Sampler(tr).sample_ode(sampling_method=meth, num_steps=5, time_shifting_factor=4)
                traj = fn(z, model.forward_with_cfg, cap_feats=cap, cap_mask=mask, cfg_scale=4.0, proportional_attn=True,
                          base_seqlen=16, scale_factor=1.0, scale_watershed=1.0)
                out[f"traj_{meth}"] = _np(traj)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz: N={N} T={text_len}", {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim > 0})


def packed_case(name, cfg, sizes, text_len, seed_w, seed_x):
    """NextDiT.forward on a LIST of differently sized latents (patchify_and_embed list branch, model.py:789-834) - the
    unmodified reference, CPU fp32 (SDPA branch with the key mask, model.py:407-415)."""
    sd = synth.synth_state_dict(cfg, seed=seed_w)
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    R.load_reference("lumina_next_t2i")
    model = R.build_reference_model(cfg, sd)
    rng = np.random.default_rng(seed_x)
    B = len(sizes)
    xs = [torch.from_numpy(rng.standard_normal((cfg.in_channels, h, w), dtype=np.float32)) for h, w in sizes]
    t = torch.from_numpy(rng.uniform(0.1, 0.9, size=B).astype(np.float32))
    cap = torch.from_numpy(rng.standard_normal((B, text_len, cfg.cap_feat_dim), dtype=np.float32))
    mask = torch.ones(B, text_len, dtype=torch.int32)
    for b in range(B):
        mask[b, text_len - 3 * b:] = 0
    out = {"config": np.array(json.dumps(cfg.to_dict())), "seed_w": seed_w, "seed_x": seed_x, "sizes": np.array(sizes, dtype=np.int32),
           "t": _np(t), "cap": _np(cap), "mask": _np(mask)}
    with torch.no_grad():
        ys = model(xs, t, cap, mask)
        assert isinstance(ys, list) and all(tuple(y.shape) == (cfg.in_channels, h, w) for y, (h, w) in zip(ys, sizes))
        # proportional attention as forward_with_cfg would leave it on the layers (model.py:891-899)
        for layer in model.layers:
            layer.attention.proportional_attn, layer.attention.base_seqlen = True, 16
        yp = model(xs, t, cap, mask)
        # the same samples one by one as [1, C, H, W] tensors: only the proportional scale (log of the PADDED length) may differ
        solo = [model(x[None], t[b:b + 1], cap[b:b + 1], mask[b:b + 1])[0] for b, x in enumerate(xs)]
    for b in range(B):
        out[f"x{b}"], out[f"y{b}"], out[f"yprop{b}"], out[f"solo_prop{b}"] = _np(xs[b]), _np(ys[b]), _np(yp[b]), _np(solo[b])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz:", sizes, {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim > 1})


def compositional_case(name, cfg, latent_hw, splits, text_len, seed_w, seed_x):
    """lumina_next_compositional_generation NextDiT (regional cross-attention, models/model.py:422-446, :852-955): the
    unmodified reference on CPU fp32.  Y = regions + 1 captions, global caption separate (demo.py:208-228)."""
    sd = synth.synth_state_dict(cfg, seed=seed_w)
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    R.load_reference("lumina_next_compositional_generation")
    M = importlib.import_module("models.model")
    assert "compositional" in M.__file__
    model = M.NextDiT(**cfg.ctor_kwargs()).eval()
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(seed_x)
    h_split, w_split = splits
    n_reg = h_split * w_split
    Y = n_reg + 1
    H, W = latent_hw
    z = torch.from_numpy(rng.standard_normal((1, cfg.in_channels, H, W), dtype=np.float32)).repeat(2, 1, 1, 1)
    t = torch.full((2,), 0.45, dtype=torch.float32)
    cap = torch.from_numpy(rng.standard_normal((Y, text_len, cfg.cap_feat_dim), dtype=np.float32))
    mask = torch.ones(Y, text_len, dtype=torch.int32)
    for r in range(Y):
        mask[r, text_len - (2 * r) % 7:] = 0 if (2 * r) % 7 else 1
    mask[Y - 1, 8:] = 0  # negative / empty caption
    gcap = torch.from_numpy(rng.standard_normal((1, text_len + 8, cfg.cap_feat_dim), dtype=np.float32))
    gmask = torch.ones(1, text_len + 8, dtype=torch.int32)
    gmask[0, -5:] = 0
    out = {"config": np.array(json.dumps(cfg.to_dict())), "seed_w": seed_w, "seed_x": seed_x, "splits": np.array(splits, dtype=np.int32),
           "z": _np(z), "t": _np(t), "cap": _np(cap), "mask": _np(mask), "gcap": _np(gcap), "gmask": _np(gmask)}
    kw = dict(global_cap_feats=gcap, global_cap_mask=gmask.bool(), h_split_num=h_split, w_split_num=w_split)
    with torch.no_grad():
        out["cfg4_prop"] = _np(model.forward_with_cfg(z, t, cap, mask.bool(), 4.0, scale_factor=1.0, scale_watershed=1.0,
                                                      base_seqlen=16, proportional_attn=True, **kw))
        out["cfg1_plain"] = _np(model.forward_with_cfg(z, t, cap, mask.bool(), 1.0, **kw))
        out["forward"] = _np(model(z, t, cap, mask.bool(), gcap, gmask.bool(), h_split, w_split))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz:", {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim > 1})


def transport_kats():
    """Everything in transport/ that is the reference's own code (no torchdiffeq inside): SDE samplers (Euler-Maruyama / Heun,
    every diffusion form and last-step rule, integrators.py:27-76, transport.py:197-344), the three path plans (path.py) and
    the training loss (transport.py:95-164), run from the unmodified modules with a fixed torch seed and a toy model callable."""
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    _, T = R.load_reference("lumina_next_t2i")
    d = {}

    def model(x, t, **kw):  # smooth, time dependent, sample dependent
        return -x * (0.5 + t.view(-1, 1, 1, 1)) + 0.1 * torch.cos(3.0 * x)

    x0 = torch.linspace(-1.0, 1.0, 2 * 3 * 4 * 4).view(2, 3, 4, 4).clone()
    cases = []
    # explicit eps where the reference's own defaulting leaves sample_eps = None (it tests `train_eps is None` AFTER assigning
    # train_eps, __init__.py:47-57) and then fails in check_interval
    for path_type, prediction, eps in (("Linear", "velocity", (None, None)), ("GVP", "velocity", (None, None)),
                                       ("VP", "score", (1e-5, 1e-3)), ("Linear", "noise", (1e-3, 1e-3))):
        tr = T.create_transport(path_type, prediction, None, eps[0], eps[1])
        for method in ("Euler", "Heun"):
            # ("constant" returns a Python float and makes th.sqrt fail inside the reference's own step - not a usable form;
            #  "inccreasing-decreasing" is the reference's spelling, path.py:62)
            for form in ("SBDM", "sigma", "linear", "decreasing", "inccreasing-decreasing"):
                for last in ("Mean", "Tweedie", "Euler", None):
                    if (form, last) not in (("SBDM", "Mean"), ("sigma", "Tweedie"), ("linear", "Euler"), ("decreasing", None),
                                            ("inccreasing-decreasing", "Mean")):
                        continue
                    key = f"sde_{path_type}_{prediction}_{method}_{form}_{last}"
                    torch.manual_seed(1234)
                    fn = T.Sampler(tr).sample_sde(sampling_method=method, diffusion_form=form, diffusion_norm=0.7, last_step=last,
                                                  last_step_size=0.04, num_steps=7)
                    xs = fn(x0.clone(), model)
                    d[key] = _np(torch.stack(xs))
                    cases.append([key, path_type, prediction, method, form, last, eps[0], eps[1]])
        # training loss with fixed noise / time draws
        torch.manual_seed(99)
        try:
            terms = tr.training_losses(model, x0.clone())
            d[f"loss_{path_type}_{prediction}"] = _np(terms["loss"])
        except AssertionError:  # the reference's Transport.sample is "not implemented" off the (0, 1) interval (transport.py:107)
            pass
        # path plan functions on a fixed (t, x)
        tt = torch.tensor([0.13, 0.77])
        plan = tr.path_sampler
        xt = x0 * 0.3
        vel = torch.sin(x0)
        tag = f"plan_{path_type}_{prediction}"
        drift_mean, drift_var = plan.compute_drift(xt, tt)
        for nm, val in (("drift_mean", drift_mean), ("drift_var", drift_var), ("score", plan.get_score_from_velocity(vel, xt, tt)),
                        ("noise", plan.get_noise_from_velocity(vel, xt, tt)), ("vel_from_score", plan.get_velocity_from_score(vel, xt, tt)),
                        ("diff_sbdm", plan.compute_diffusion(xt, tt, form="SBDM", norm=0.7)),
                        ("diff_dec", plan.compute_diffusion(xt, tt, form="decreasing", norm=0.7))):
            d[f"{tag}_{nm}"] = _np(val * torch.ones(1))
    d["x0"] = _np(x0)
    d["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(OUT, "transport_kat.npz"), **d)
    print("transport_kat.npz:", len(cases), "sde cases")


def likelihood_kats():
    """``Sampler.sample_ode_likelihood`` (transport.py:393-450) run from the UNMODIFIED reference modules: the Hutchinson probe
    draws (``th.randint`` per function call), the autograd divergence ``eps^T d(drift . eps)/dx``, the reversed time ``1 - t``,
    the sign of the returned drift, ``prior_logp`` and the final ``prior - delta`` - everything that is the reference's own
    code.  The fixed-grid stepping of the (x, logp) TUPLE state goes through the torchdiffeq stub (oracle/odeint_oracle.py:
    flatten, step, split - parity unpinned like every torchdiffeq piece).  Toy velocity fields with known divergence so the
    fixture can be checked by hand: `lin` has div = -D (0.5 + t') exactly for every probe, `cos` adds a state dependent term."""
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    _, T = R.load_reference("lumina_next_t2i")
    d = {}
    fields = {"lin": lambda x, t, **kw: -x * (0.5 + t.view(-1, 1, 1, 1)),
              "cos": lambda x, t, **kw: -x * (0.5 + t.view(-1, 1, 1, 1)) + 0.1 * torch.cos(3.0 * x)}
    x0 = torch.linspace(-1.0, 1.0, 2 * 3 * 4 * 4).view(2, 3, 4, 4).clone() * 0.8
    cases = []
    for path_type, eps in (("Linear", (None, None)), ("GVP", (None, None))):
        tr = T.create_transport(path_type, "velocity", None, eps[0], eps[1])
        for fname, model in fields.items():
            for method, n in (("euler", 6), ("midpoint", 5), ("rk4", 4)):
                key = f"lik_{path_type}_{fname}_{method}"
                torch.manual_seed(4321)
                fn = T.Sampler(tr).sample_ode_likelihood(sampling_method=method, num_steps=n)
                with torch.no_grad():
                    logp, z = fn(x0.clone(), model)
                d[key + "_logp"], d[key + "_z"] = _np(logp), _np(z)
                cases.append([key, path_type, fname, method, n])
    d["x0"] = _np(x0)
    d["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(OUT, "likelihood_kat.npz"), **d)
    print("likelihood_kat.npz:", len(cases), "cases", {k: v.tolist() for k, v in d.items() if k.endswith("euler_logp")})


def _fresh_import(pkg, module):
    """import <pkg>/<module> of the reference with a clean `models` namespace (every sub-project calls its package `models`)"""
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    root = os.path.join(R.REFERENCE_ROOT, pkg)
    sys.path = [q for q in sys.path if not q.startswith(R.REFERENCE_ROOT)]
    sys.path.insert(0, root)
    return importlib.import_module(module)


def family_case(name, cfg, latent_hw, seed_w, seed_x, text_len=16, uncond_len=8):
    """imagenet / moe / flag_t2i fixtures from the unmodified reference modules (fp32, CPU, eager)."""
    os.environ["TORCHDYNAMO_DISABLE"] = "1"  # models.py:463 decorates the SwiGLU with torch.compile; eager is the same maths
    sd = synth.synth_state_dict(cfg, seed=seed_w)
    ins = synth.synth_inputs(cfg, latent_hw=latent_hw, text_len=text_len, uncond_len=uncond_len, seed=seed_x)
    pkg, module = {"imagenet": ("Next-DiT-ImageNet", "models.models"), "moe": ("Next-DiT-MoE", "models.models2"),
                   "moe_time": ("Next-DiT-MoE", "models.models"), "moe_space": ("Next-DiT-MoE", "models.models1"),
                   "flag_t2i": ("lumina_t2i", "models.model")}[cfg.family]
    mod = _fresh_import(pkg, module)
    kw = cfg.ctor_kwargs()
    model = mod.DiT_Llama(**kw).eval()
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    out = {"config": np.array(json.dumps(cfg.to_dict())), "seed_w": seed_w, "seed_x": seed_x, "package": np.array(pkg),
           "text_len": text_len, "uncond_len": uncond_len, "latent_hw": np.array(latent_hw)}
    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(_np(o))) for l in model.layers]
    with torch.no_grad():
        if cfg.has_text:
            z, t, cap, mask = ins
            out.update(z=_np(z), t=_np(t), cap=_np(cap), mask=_np(mask))
            out["forward"] = _np(model(z, t, cap, mask))
            out["hidden"] = np.stack(hidden)
            for h in hooks:
                h.remove()
            out["cfg4_prop"] = _np(model.forward_with_cfg(z, t, cap, mask, 4.0, base_seqlen=16, proportional_attn=True))
            out["cfg4_rope"] = _np(model.forward_with_cfg(z, t, cap, mask, 4.0, rope_scaling_factor=2.0, ntk_factor=1.5,
                                                          base_seqlen=16, proportional_attn=True))
            model.forward_with_cfg(z, t, cap, mask, 1.0, rope_scaling_factor=1.0, ntk_factor=1.0)  # restore the table
            out["cfg1_plain"] = _np(model.forward_with_cfg(z, t, cap, mask, 1.0))
        else:
            z, t, y = ins
            out.update(z=_np(z), t=_np(t), y=_np(y))
            out["forward"] = _np(model(z, t, y))
            out["hidden"] = np.stack(hidden)
            for h in hooks:
                h.remove()
            out["cfg4"] = _np(model.forward_with_cfg(z, t, y, 4.0))
            out["cfg4_rope"] = _np(model.forward_with_cfg(z, t, y, 4.0, rope_scaling_factor=2.0, ntk_factor=1.5))
            model.forward_with_cfg(z, t, y, 1.0, rope_scaling_factor=1.0, ntk_factor=1.0)
            out["cfg1_plain"] = _np(model.forward_with_cfg(z, t, y, 1.0))
            # BASELINE configs[0]: 4-step Euler ODE driven through the reference's own transport package
            tmod = _fresh_import("Next-DiT-ImageNet", "transport")
            tr = tmod.create_transport("Linear", "velocity", None, None, None)
            fn = tmod.Sampler(tr).sample_ode(sampling_method="euler", num_steps=5)
            if cfg.family == "imagenet":
                traj = fn(z, model.forward_with_cfg, y=y, cfg_scale=4.0)
                out["traj_euler"] = _np(traj)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz:", {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim > 0})


def mini_ode_kats():
    """time grids of the mini package's flat ODE class incl. the img2img `strength` cut
    (lumina_next_t2i_mini/transport.py:57-83); the module is loaded by path (its package has no __init__ we could import)"""
    import importlib.util
    path = os.path.join(R.REFERENCE_ROOT, "lumina_next_t2i_mini", "transport.py")
    spec = importlib.util.spec_from_file_location("ref_mini_transport", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # needs only the torchdiffeq stub (oracle/stubs) for `from torchdiffeq import odeint`
    d = {}
    cases = {"n30_s4": dict(num_steps=30, time_shifting_factor=4), "n30_s4_str06": dict(num_steps=30, time_shifting_factor=4, strength=0.6),
             "n10_str03": dict(num_steps=10, strength=0.3), "n17_s6_str085": dict(num_steps=17, time_shifting_factor=6.0, strength=0.85),
             "n5": dict(num_steps=5)}
    for k, kw in cases.items():
        d[k] = _np(mod.ODE(sampler_type="euler", **kw).t)
    d["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(OUT, "mini_ode.npz"), **d)
    print("mini_ode.npz:", {k: v.shape for k, v in d.items() if v.ndim > 0})


def solver_kats():
    """The reference's own in-tree ODE step, ``midpoint_solver`` (visual_anagrams/generate.py:212-219), driven exactly as its
    caller does (generate.py:385-414: ``timesteps.tolist()`` floats, ``noisy_img = noisy_img - (-midpoint_solver(...))``) over
    seeded grids with a closed-form drift and the unmodified tiny NextDiT's ``forward_with_cfg`` as drift.  Every call the
    solver makes to ``func`` is logged: the second call's state argument is ``y0 + f0 * dt/2`` = one explicit Euler step of
    size dt/2 written by the reference itself.  tests/test_oracle_golden.py holds oracle/odeint_oracle.py (euler, midpoint) to
    these; tests/test_gpu_model.py holds lt_sample_ode to odeint_oracle bit for bit and to the tiny-model trajectory here."""
    midpoint_solver = R.load_anagrams_solvers()
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "transport" or k.startswith("transport.")]:
        del sys.modules[m]
    sys.path = [q for q in sys.path if not q.startswith(R.REFERENCE_ROOT)]
    _, T = R.load_reference("lumina_next_t2i")
    integ = importlib.import_module("transport.integrators")
    cfg = synth.TINY
    model = R.build_reference_model(cfg, synth.synth_state_dict(cfg, seed=31))
    z, _, cap, mask = synth.synth_inputs(cfg, latent_hw=(16, 16), text_len=16, uncond_len=8, seed=32)
    model_kw = dict(cap_feats=cap, cap_mask=mask, cfg_scale=4.0, proportional_attn=True, base_seqlen=16, scale_factor=1.0,
                    scale_watershed=1.0)

    def closed(y, t):  # smooth, time dependent, state dependent (nonlinear)
        return -y * (0.5 + t.view(-1, 1, 1, 1)) + 0.1 * torch.cos(3.0 * y)

    from functools import partial
    drifts = {"closed": closed, "tiny": partial(model.forward_with_cfg, **model_kw)}  # generate.py:400: partial(model.forward_with_cfg, **model_kwargs)
    grids = {"uniform5": integ.ode(drift=None, t0=0, t1=1, sampler_type="midpoint", num_steps=5, atol=1e-6, rtol=1e-3).t,
             "shift4_6": integ.ode(drift=None, t0=0, t1=1, sampler_type="midpoint", num_steps=6, atol=1e-6, rtol=1e-3,
                                   time_shifting_factor=4).t}
    y0 = {"closed": torch.linspace(-1.0, 1.0, 2 * 3 * 4 * 4).view(2, 3, 4, 4).clone(), "tiny": z}
    d = {"config": np.array(json.dumps(cfg.to_dict())), "seed_w": 31, "seed_x": 32, "cap": _np(cap), "mask": _np(mask),
         "model_kw": np.array(json.dumps({k: v for k, v in model_kw.items() if not torch.is_tensor(v)}))}
    for gname, grid in grids.items():
        d[f"grid_{gname}"] = _np(grid)
        timesteps = grid.tolist()  # generate.py:385
        for dname, func in drifts.items():
            log_y, log_t = [], []

            def logged(y, t, _f=func):
                log_y.append(_np(y).copy())
                log_t.append(_np(t).copy())
                return _f(y, t)

            y = y0[dname].clone()
            traj, incs = [_np(y).copy()], []
            with torch.no_grad():
                for i in range(len(timesteps) - 1):
                    noise = -midpoint_solver(logged, timesteps[i], timesteps[i + 1], y)  # generate.py:402-404
                    incs.append(_np(-noise).copy())
                    y = y - noise                                                         # generate.py:414
                    traj.append(_np(y).copy())
            tag = f"{gname}_{dname}"
            d[f"traj_{tag}"], d[f"inc_{tag}"] = np.stack(traj), np.stack(incs)
            d[f"calls_y_{tag}"], d[f"calls_t_{tag}"] = np.stack(log_y), np.stack(log_t)
    np.savez_compressed(os.path.join(OUT, "solver_kat.npz"), **d)
    print("solver_kat.npz:", {k: v.shape for k, v in d.items() if hasattr(v, "shape") and v.ndim > 1})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    if sys.argv[1:] == ["solver"]:  # python -m oracle.make_golden solver: only tests/golden/solver_kat.npz
        return solver_kats()
    if sys.argv[1:] == ["likelihood"]:  # ... only tests/golden/likelihood_kat.npz
        return likelihood_kats()
    models_mod, transport_mod = R.load_reference()
    kats(models_mod, transport_mod)
    model_case("nextdit_tiny", synth.TINY, "lumina_next_t2i", (16, 16), 16, 8, 0, 1)
    model_case("nextdit_tiny_rect", synth.TINY, "lumina_next_t2i", (12, 20), 13, 5, 3, 4)
    model_case("nextdit_tiny_gqa", synth.TINY_GQA, "lumina_next_t2i_mini", (16, 16), 16, 8, 5, 6)
    family_case("imagenet_tiny", synth.TINY_IMAGENET, (16, 16), 7, 8)
    family_case("moe_tiny", synth.TINY_MOE, (16, 16), 9, 10)
    family_case("moe_time_tiny", synth.TINY_MOE_TIME, (16, 16), 19, 20)
    family_case("moe_space_tiny", synth.TINY_MOE_SPACE, (16, 16), 23, 24)
    family_case("flag_tiny", synth.TINY_FLAG, (16, 24), 11, 12)
    mini_ode_kats()
    transport_kats()
    packed_case("nextdit_tiny_packed", synth.TINY, [(16, 16), (12, 20), (8, 24), (16, 16)], 16, 13, 14)
    compositional_case("compositional_tiny", synth.TINY, (16, 24), (2, 2), 16, 15, 16)
    compositional_case("compositional_tiny_1x3", synth.TINY, (12, 24), (1, 3), 13, 17, 18)
    solver_kats()
    likelihood_kats()


if __name__ == "__main__":
    main()
