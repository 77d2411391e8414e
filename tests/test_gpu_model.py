"""-m gpu: the engine (NextDiT.forward / forward_with_cfg / sample_ode through the C ABI) against the CPU
oracle and the committed golden vectors of the reference.

Tolerance policy (bf16 engine vs fp32 reference): the reference's own bf16 path differs
from its fp32 path by rel-L2 ~1.4e-2 (plain forward) and ~4e-2 (forward_with_cfg at cfg 4, guidance amplifies
cond-uncond error ~2.8x) on synthetic weights.  Gates: plain forward <= 2.5e-2, cfg-4 <= 6e-2, i.e. 1.5 x the
measured floor, and the bf16-emulating oracle (same rounding choreography) must be matched 3x tighter.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import models
from lumina_t2x_amd.transport import Sampler, create_transport
from oracle import nextdit_oracle as O
from oracle import odeint_oracle as OD
from oracle import synth

from gpu_util import max_abs, rel_l2, set_option

pytestmark = pytest.mark.gpu

TOL_FWD, TOL_CFG4 = 2.5e-2, 6e-2


def _golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    cfg = synth.NextDiTConfig(**json.loads(str(g["config"])))
    return g, cfg


def _model(cfg, seed, dtype=torch.bfloat16):
    m = models.NextDiT(**cfg.ctor_kwargs())
    m.load_state_dict(synth.synth_state_dict(cfg, seed=seed), strict=True)
    return m.eval().to("cuda", dtype)


def _inputs(g, dtype=torch.bfloat16):
    z = torch.from_numpy(g["z"]).to("cuda", dtype)
    t = torch.from_numpy(g["t"]).cuda()
    cap = torch.from_numpy(g["cap"]).to("cuda", dtype)
    mask = torch.from_numpy(g["mask"]).cuda()
    return z, t, cap, mask


@pytest.mark.parametrize("name", ["nextdit_tiny", "nextdit_tiny_rect", "nextdit_tiny_gqa"])
def test_engine_matches_reference_golden(golden_dir, name):
    g, cfg = _golden(golden_dir, name)
    model = _model(cfg, int(g["seed_w"]))
    z, t, cap, mask = _inputs(g)
    out = model(z, t, cap, mask)
    assert out.shape == z.shape and out.dtype == z.dtype
    assert rel_l2(out, torch.from_numpy(g["forward"])) < TOL_FWD, rel_l2(out, torch.from_numpy(g["forward"]))
    cases = [
        ("cfg4_prop", t, dict(cfg_scale=4.0, base_seqlen=16, proportional_attn=True), TOL_CFG4),
        ("cfg4_lin2", torch.full((2,), 0.1, device="cuda"), dict(cfg_scale=4.0, scale_factor=2.0, scale_watershed=0.3, base_seqlen=16, proportional_attn=True), TOL_CFG4),
        ("cfg4_ntk2", torch.full((2,), 0.8, device="cuda"), dict(cfg_scale=4.0, scale_factor=2.0, scale_watershed=0.3, base_seqlen=16, proportional_attn=True), TOL_CFG4),
        ("cfg1_plain", t, dict(cfg_scale=1.0), TOL_FWD),
    ]
    for key, tt, kw, tol in cases:
        got = model.forward_with_cfg(z, tt, cap, mask, **kw)
        ref = torch.from_numpy(g[key])
        err = rel_l2(got, ref)
        assert err < tol, (key, err)
        # CFG quirk: channels [:3] identical across the pair (model.py:908-913)
        assert torch.equal(got[0, :3], got[1, :3]), key
        # the unguided channel 3 is a clean probe of the un-amplified network error
        assert rel_l2(got[:, 3], ref[:, 3]) < TOL_FWD, key
        # round 6 (option prologue_fused: measured slower than the launches it removes, so off by default): the conditioning prologue as three launches less, the separate kernels' own statements
        set_option("prologue_fused", 7)
        try:
            fused = model.forward_with_cfg(z, tt, cap, mask, **kw)
        finally:
            set_option("prologue_fused", 0)
        assert torch.equal(fused, got), key


@pytest.mark.parametrize("name", ["nextdit_tiny", "nextdit_tiny_gqa"])
def test_engine_matches_bf16_choreography_oracle(golden_dir, name):
    g, cfg = _golden(golden_dir, name)
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    model = _model(cfg, int(g["seed_w"]))
    z, t, cap, mask = _inputs(g)
    got = model.forward_with_cfg(z, t, cap, mask, 4.0, base_seqlen=16, proportional_attn=True)
    want = O.forward_with_cfg(sd, cfg, z.float().cpu(), t.cpu(), cap.float().cpu(), mask.cpu(), 4.0, base_seqlen=16,
                              proportional_attn=True, bf16=True)
    err = rel_l2(got, want)
    assert err < 2e-2, err


def test_engine_fp32_io_and_determinism(golden_dir):
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    z, t, cap, mask = _inputs(g)
    a = model.forward_with_cfg(z, t, cap, mask, 4.0)
    b = model.forward_with_cfg(z, t, cap, mask, 4.0)
    assert torch.equal(a, b)
    # fp32 state in/out (values are bf16-exact inputs, so only the output cast differs)
    c = model.forward_with_cfg(z.float(), t, cap, mask, 4.0)
    assert c.dtype == torch.float32 and torch.equal(c.to(torch.bfloat16), a)
    # the second half of x is ignored (model.py:901-902)
    z2 = z.clone()
    z2[1] = 123.0
    assert torch.equal(model.forward_with_cfg(z2, t, cap, mask, 4.0), a)


def test_engine_batch_of_several_images_is_row_independent(golden_dir):
    """forward_with_cfg on 3 images (batch 6 = 3 cond + 3 uncond rows, different prompts / masks / timesteps per image) equals
    the three single-image calls bit for bit: nothing in the path mixes batch rows, and the engine's
    per-batch plumbing (prompt K/V, masks, adaLN rows, CFG pairing cond row i <-> uncond row i + B/2, model.py:901-912) holds
    for B > 2; the engine is re-created with larger limits on the way."""
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    gen = torch.Generator().manual_seed(5)
    n = 3
    zc = torch.randn(n, 4, 16, 16, generator=gen).to("cuda", torch.bfloat16)
    z = torch.cat([zc, zc])
    t1 = torch.tensor([0.2, 0.5, 0.9])
    t = torch.cat([t1, t1]).cuda()
    cap = torch.randn(2 * n, 16, cfg.cap_feat_dim, generator=gen).to("cuda", torch.bfloat16)
    mask = torch.ones(2 * n, 16, dtype=torch.int32, device="cuda")
    for i in range(n):
        mask[i, 16 - 2 * i:] = 0
        mask[n + i, 8 - i:] = 0
    kw = dict(cfg_scale=4.0, proportional_attn=True, base_seqlen=16)
    full = model.forward_with_cfg(z, t, cap, mask, **kw)
    assert full.shape == z.shape and torch.equal(full[:n, :3], full[n:, :3])
    for i in range(n):
        idx = torch.tensor([i, n + i], device="cuda")
        one = model.forward_with_cfg(z[idx], t[idx], cap[idx].contiguous(), mask[idx].contiguous(), **kw)
        assert torch.equal(one, full[idx]), i


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_engine_ode_loop_equals_stepwise_torch(golden_dir, method):
    """lt_sample_ode (C++ loop + ode_combine kernels) == torchdiffeq arithmetic driven from Python with the
    SAME model evaluations -> must agree bit for bit."""
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    z, _, cap, mask = _inputs(g)
    kw = dict(cap_feats=cap, cap_mask=mask, cfg_scale=4.0, proportional_attn=True, base_seqlen=16)
    fn = Sampler(create_transport()).sample_ode(sampling_method=method, num_steps=5, time_shifting_factor=4)
    fast = fn(z, model.forward_with_cfg, **kw)
    assert fast.shape == (5,) + tuple(z.shape)
    assert model._engine.last_nfe() == 4 * {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    slow = OD.sample_ode(lambda x, tv, **k: model.forward_with_cfg(x, tv.cuda(), **k), z, 5, method=method,
                         time_shifting_factor=4, **kw)
    assert torch.equal(fast, slow), max_abs(fast, slow)


def test_mini_ode_img2img_on_engine_equals_generic_stepping(golden_dir):
    """lumina_next_t2i_mini ODE class with the img2img `strength` cut (transport.py:79-80, sample_img2img.py:146-216): the
    engine fast path (one lt_sample_ode call over the cut grid) == the same grid stepped through the model callable."""
    from lumina_t2x_amd.transport.integrators import fixed_grid_odeint
    from lumina_t2x_amd.transport.mini import ODE

    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    z, _, cap, mask = _inputs(g)
    kw = dict(cap_feats=cap, cap_mask=mask, cfg_scale=4.0, proportional_attn=True, base_seqlen=16, scale_factor=1.0,
              scale_watershed=1.0)
    o = ODE(9, "midpoint", 4, strength=0.6)
    assert len(o.t) == 9 - int(9 * (1 - 0.6))
    fast = o.sample(z, model.forward_with_cfg, **kw)
    assert fast.shape == (len(o.t),) + tuple(z.shape)
    assert model._engine.last_nfe() == 2 * (len(o.t) - 1)

    def fn(t, y):  # bf16 state: torchdiffeq hands the model t rounded to the state dtype
        return model.forward_with_cfg(y, torch.ones(y.size(0), device=y.device) * t.to(y.dtype).float(), **kw)

    slow = fixed_grid_odeint(fn, z, o.t.cuda(), method="midpoint")
    assert torch.equal(fast, slow), max_abs(fast, slow)


@pytest.mark.parametrize("method", ["euler", "midpoint"])
def test_engine_trajectory_vs_reference_golden(golden_dir, method):
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    z, _, cap, mask = _inputs(g)
    fn = Sampler(create_transport()).sample_ode(sampling_method=method, num_steps=5, time_shifting_factor=4)
    traj = fn(z, model.forward_with_cfg, cap_feats=cap, cap_mask=mask, cfg_scale=4.0, proportional_attn=True, base_seqlen=16)
    ref = torch.from_numpy(g[f"traj_{method}"])
    # gate: 1.5 x the error of the reference's own bf16 choreography against its fp32 self at the same
    # level, re-measured here on this weight draw with the bf16-emulating oracle (bf16 state, bf16 rounding points);
    # A.6 measured 1.9e-2 after 4 Euler steps on the 600M model; guidance 4 and 8 NFE (midpoint) sit higher
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    zc, capc, maskc = z.cpu(), cap.float().cpu(), mask.cpu()
    floor_traj = OD.sample_ode(lambda x, tv, **k: O.forward_with_cfg(sd, cfg, x, tv, bf16=True, **k).to(torch.bfloat16), zc, 5,
                               method=method, time_shifting_factor=4, cap_feats=capc, cap_mask=maskc, cfg_scale=4.0,
                               proportional_attn=True, base_seqlen=16)
    floor = rel_l2(floor_traj[-1], ref[-1])
    err = rel_l2(traj[-1], ref[-1])
    assert err < max(5e-2, 1.5 * floor), (err, floor)


@pytest.mark.parametrize("grid,num_steps,shift", [("uniform5", 5, None), ("shift4_6", 6, 4)])
def test_engine_midpoint_trajectory_vs_reference_midpoint_solver(golden_dir, grid, num_steps, shift):
    """lt_sample_ode `midpoint` against the trajectory the reference's OWN in-tree midpoint step produced with the unmodified tiny
    NextDiT as drift (visual_anagrams/generate.py:212-219 driven as :385-414; tests/golden/solver_kat.npz).  The CPU suite holds
    odeint_oracle to the same fixture and test_engine_ode_loop_equals_stepwise_torch holds lt_sample_ode to odeint_oracle bit for
    bit; this closes the chain end to end (bf16 engine vs fp32 reference: the trajectory gate of SURVEY 8d)."""
    g, cfg = _golden(golden_dir, "solver_kat")
    model = _model(cfg, int(g["seed_w"]))
    kw = json.loads(str(g["model_kw"]))
    cap = torch.from_numpy(g["cap"]).to("cuda", torch.bfloat16)
    mask = torch.from_numpy(g["mask"]).cuda()
    ref = torch.from_numpy(g[f"traj_{grid}_tiny"])
    z = ref[0].to("cuda", torch.bfloat16)
    fn = Sampler(create_transport()).sample_ode(sampling_method="midpoint", num_steps=num_steps, time_shifting_factor=shift)
    traj = fn(z, model.forward_with_cfg, cap_feats=cap, cap_mask=mask, **kw)
    assert traj.shape == ref.shape and model._engine.last_nfe() == 2 * (num_steps - 1)
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    floor_traj = OD.odeint(lambda t, y: O.forward_with_cfg(sd, cfg, y, torch.ones(2) * t.float(), bf16=True, cap_feats=cap.float().cpu(),
                                                           cap_mask=mask.cpu(), **kw).to(torch.bfloat16),
                           ref[0].to(torch.bfloat16), torch.from_numpy(g[f"grid_{grid}"]), method="midpoint")
    floor = rel_l2(floor_traj[-1], ref[-1])
    err = rel_l2(traj[-1], ref[-1])
    assert err < max(5e-2, 1.5 * floor), (err, floor)
    # fp32 state: the engine's arithmetic on the state is then the reference's (fp32 y, fp32 dt); only the drift is bf16
    z32 = ref[0].cuda()
    traj32 = fn(z32, model.forward_with_cfg, cap_feats=cap, cap_mask=mask, **kw)
    assert traj32.dtype == torch.float32 and rel_l2(traj32[-1], ref[-1]) < max(5e-2, 1.5 * floor)


def test_full_width_two_layers_vs_oracle():
    """Next-DiT 2B widths (d 2304, hd 72, F 6144), 1024^2 latent (N = 4096, M = 8192 rows), 2 layers: exercises
    every kernel at the BASELINE cfg-2 shapes against the CPU oracle."""
    cfg = synth.NextDiTConfig(n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=11)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(128, 128), text_len=128, uncond_len=8, seed=12)
    model = models.NextDiT(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    zb, capb = z.to("cuda", torch.bfloat16), cap.to("cuda", torch.bfloat16)
    got = model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 4.0, base_seqlen=4096, proportional_attn=True)
    want = O.forward_with_cfg(sd, cfg, zb.float().cpu(), t, capb.float().cpu(), mask, 4.0, base_seqlen=4096,
                              proportional_attn=True)
    err = rel_l2(got, want)
    assert err < TOL_CFG4, err
    assert rel_l2(got[:, 3], want[:, 3]) < TOL_FWD
    # size-independent properties
    assert torch.equal(got[0, :3], got[1, :3])
    one = model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 1.0, base_seqlen=4096, proportional_attn=True)
    plain = model(zb, t.cuda(), capb, mask.cuda())  # proportional flags persist on the module like the reference
    assert rel_l2(one[0], plain[0]) < 1e-2


def test_cfg4_2048px_gqa_ntk_one_layer_vs_oracle():
    """BASELINE configs[3] shapes: Lumina-Next-SFT 2B at 2048x2048 (latent 256x256 -> 16384 tokens per sample, M = 32768
    rows), GQA (32 query / 8 kv heads), NTK-aware RoPE (scale_factor 2, t >= watershed), proportional attention against
    base_seqlen 4096 - one layer at full width against the CPU oracle, plus the size-independent CFG property."""
    cfg = synth.NextDiTConfig(n_layers=1, n_kv_heads=8)
    sd = synth.synth_state_dict(cfg, seed=51)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(256, 256), text_len=64, uncond_len=8, seed=52, t_value=0.6)
    model = models.NextDiT(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    zb, capb = z.to("cuda", torch.bfloat16), cap.to("cuda", torch.bfloat16)
    kw = dict(scale_factor=2.0, scale_watershed=0.3, base_seqlen=4096, proportional_attn=True)
    got = model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 4.0, **kw)
    assert got.shape == zb.shape and torch.isfinite(got.float()).all()
    assert torch.equal(got[0, :3], got[1, :3])
    want = O.forward_with_cfg(sd, cfg, zb.float().cpu(), t, capb.float().cpu(), mask, 4.0, **kw)
    err = rel_l2(got, want)
    assert err < TOL_CFG4, err
    assert rel_l2(got[:, 3], want[:, 3]) < TOL_FWD
    # linear-interpolation branch (t < watershed) must differ from the NTK branch and still match the oracle
    t_lo = torch.full((2,), 0.1)
    got_lo = model.forward_with_cfg(zb, t_lo.cuda(), capb, mask.cuda(), 4.0, **kw)
    want_lo = O.forward_with_cfg(sd, cfg, zb.float().cpu(), t_lo, capb.float().cpu(), mask, 4.0, **kw)
    assert rel_l2(got_lo, want_lo) < TOL_CFG4, rel_l2(got_lo, want_lo)


def test_sample_driver_end_to_end_with_injected_encoder(golden_dir, tmp_path):
    """lumina_t2x_amd.sample.run (reference sample.py:85-265 flow) on the engine: checkpoint directory -> model ->
    per-caption CFG sampling -> output files.  The third-party stages (text encoder, VAE) are injected; the latent the
    driver writes must equal a direct Sampler call with the same seed and kwargs."""
    import argparse

    from safetensors.torch import save_file

    from lumina_t2x_amd import sample as S

    g, cfg = _golden(golden_dir, "nextdit_tiny")
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ck / "consolidated_ema.00-of-01.safetensors"))
    torch.save(argparse.Namespace(model="NextDiT_tiny_test", qk_norm=cfg.qk_norm, image_size=256, vae="sdxl"), str(ck / "model_args.pth"))
    models.__dict__["NextDiT_tiny_test"] = lambda **kw: models.NextDiT(**{**cfg.ctor_kwargs(), **kw})
    (tmp_path / "prompts.txt").write_text("a red cube\n\na blue sphere\n")
    gen = torch.Generator().manual_seed(3)
    table = {c: torch.randn(16, cfg.cap_feat_dim, generator=gen) for c in ("a red cube", "a blue sphere", "")}

    def encode(caps):
        feats = torch.stack([table[c] for c in caps]).to("cuda", torch.bfloat16)
        mask = torch.ones(len(caps), 16, dtype=torch.int64, device="cuda")
        mask[-1, 8:] = 0
        return feats, mask

    decoded = []

    def decode(lat):
        decoded.append(lat.clone())
        return torch.sigmoid(lat[:, :3].float())

    out = tmp_path / "out"
    args = S.build_parser().parse_args(["--ckpt", str(ck), "--caption_path", str(tmp_path / "prompts.txt"), "--resolution",
                                        "256:128x128", "--num_sampling_steps", "4", "--sampling-method", "midpoint",
                                        "--time_shifting_factor", "4", "--seed", "11", "--image_save_path", str(out)])
    try:
        info = S.run(args, encode_fn=encode, cap_feat_dim=cfg.cap_feat_dim, decode_fn=decode)
    finally:
        del models.__dict__["NextDiT_tiny_test"]
    assert [i["caption"] for i in info] == ["a red cube", "a blue sphere"] and len(decoded) == 2
    assert all(os.path.exists(i["image_url"]) and i["image_url"].endswith("_128x128.png") for i in info)
    assert json.load(open(out / "data.json")) == info
    # the same sample through the public API
    model = _model(cfg, int(g["seed_w"]))
    torch.manual_seed(11)
    z = torch.randn([1, 4, 16, 16], device="cuda").to(torch.bfloat16).repeat(2, 1, 1, 1)
    feats, mask = encode(["a red cube", ""])
    fn = Sampler(create_transport()).sample_ode(sampling_method="midpoint", num_steps=4, time_shifting_factor=4.0)
    want = fn(z, model.forward_with_cfg, cap_feats=feats, cap_mask=mask, cfg_scale=4.0, proportional_attn=True,
              base_seqlen=256, scale_factor=1.0, scale_watershed=1.0)[-1][:1]
    assert torch.equal(decoded[0], want / 0.13025)


def test_img2img_driver_end_to_end_with_injected_encoders(golden_dir, tmp_path):
    """lumina_t2x_amd.sample_img2img.run (reference lumina_next_t2i_mini/sample_img2img.py:146-260): image -> VAE latent -> noise mix
    at the cut grid's first time -> ODE(strength) on the engine -> files.  Third-party stages injected; the latent must equal the
    same mix + ODE.sample done by hand, and the cut grid must be the reference's (mini_ode.npz)."""
    import argparse

    from safetensors.torch import save_file

    from lumina_t2x_amd import sample_img2img as S
    from lumina_t2x_amd.transport.mini import ODE

    g, cfg = _golden(golden_dir, "nextdit_tiny")
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ck / "consolidated_ema.00-of-01.safetensors"))
    torch.save(argparse.Namespace(model="NextDiT_tiny_test", qk_norm=cfg.qk_norm, image_size=256, vae="sdxl"), str(ck / "model_args.pth"))
    models.__dict__["NextDiT_tiny_test"] = lambda **kw: models.NextDiT(**{**cfg.ctor_kwargs(), **kw})
    (tmp_path / "prompts.txt").write_text("a red cube\n")
    gen = torch.Generator().manual_seed(5)
    table = {c: torch.randn(16, cfg.cap_feat_dim, generator=gen) for c in ("a red cube", "")}
    image = torch.rand(3, 128, 128, generator=gen).mul(2).sub(1).cuda()

    def encode(caps):
        feats = torch.stack([table[c] for c in caps]).to("cuda", torch.bfloat16)
        mask = torch.ones(len(caps), 16, dtype=torch.int64, device="cuda")
        mask[-1, 8:] = 0
        return feats, mask

    def vae_encode(img):  # a stand-in "VAE": 8 x 8 average pooling to 4 channels
        return torch.nn.functional.avg_pool2d(torch.cat([img, img[:, :1]], dim=1), 8)

    decoded = []

    def decode(lat):
        decoded.append(lat.clone())
        return torch.sigmoid(lat[:, :3].float())

    out = tmp_path / "out"
    args = S.build_parser().parse_args(["--ckpt", str(ck), "--image", "unused.png", "--caption_path", str(tmp_path / "prompts.txt"),
                                        "--resolution", "256:128x128", "--num_sampling_steps", "9", "--solver", "midpoint", "--strength", "0.6",
                                        "--time_shifting_factor", "4", "--seed", "13", "--image_save_path", str(out)])
    try:
        info = S.run(args, encode_fn=encode, cap_feat_dim=cfg.cap_feat_dim, vae_encode_fn=vae_encode, decode_fn=decode, image=image)
    finally:
        del models.__dict__["NextDiT_tiny_test"]
    assert len(info) == 1 and os.path.exists(info[0]["image_url"]) and info[0]["solver"] == "midpoint"
    assert json.load(open(out / "data.json")) == info
    ode = ODE(9, "midpoint", 4.0, strength=0.6)
    assert len(ode.t) == 9 - int(9 * (1 - 0.6))
    model = _model(cfg, int(g["seed_w"]))
    torch.random.manual_seed(13)
    x1 = vae_encode(image[None]).mul(0.13025)
    z = torch.randn([1, 4, 16, 16], device="cuda").to(torch.bfloat16)
    z = (z * (1 - float(ode.t[0])) + x1.to(torch.bfloat16) * float(ode.t[0])).repeat(2, 1, 1, 1)
    feats, mask = encode(["a red cube", ""])
    want = ode.sample(z, model.forward_with_cfg, cap_feats=feats, cap_mask=mask, cfg_scale=4.0, proportional_attn=True, base_seqlen=256,
                      scale_factor=1.0, scale_watershed=1.0)[-1][:1]
    assert model._engine.last_nfe() == 2 * (len(ode.t) - 1)
    assert torch.equal(decoded[0], want / 0.13025)


def test_lumina_next_cli_infer_with_injected_encoder(golden_dir, tmp_path):
    """lumina_t2x_amd.cli.infer (reference utils/cli.py:161-333 flow): yaml settings -> sampler + model kwargs -> one CFG solve
    on the engine -> decoded file named after the caption.  The latent handed to the (injected) VAE must equal a direct Sampler
    call with the same seed; note the CLI's height-first latent and its '(Extrapolation)' time-aware scaling."""
    import argparse

    from safetensors.torch import save_file

    from lumina_t2x_amd import cli

    g, cfg = _golden(golden_dir, "nextdit_tiny")
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ck / "consolidated.00-of-01.safetensors"))
    torch.save(argparse.Namespace(model="NextDiT_tiny_cli", qk_norm=cfg.qk_norm, image_size=64, vae="sdxl"), str(ck / "model_args.pth"))
    models.__dict__["NextDiT_tiny_cli"] = lambda **kw: models.NextDiT(**{**cfg.ctor_kwargs(), **kw})
    settings = tmp_path / "settings.yaml"
    settings.write_text(f"""
- settings:
  model: {{ckpt: "{ck}", ckpt_lm: "unused", token: ""}}
  transport: {{path_type: "Linear", prediction: "velocity", loss_weight: "velocity", sample_eps: 0.1, train_eps: 0.2}}
  ode: {{atol: 1.0e-6, rtol: 1.0e-3, reverse: false, likelihood: false}}
  infer: {{resolution: "(Extrapolation) 64x128", num_sampling_steps: 3, cfg_scale: 3.0, solver: "euler", t_shift: 4,
          scaling_method: "Time-aware", scale_watershed: 0.3, proportional_attn: true, seed: 5}}
""")
    gen = torch.Generator().manual_seed(4)
    table = {c: torch.randn(16, cfg.cap_feat_dim, generator=gen) for c in ("a tall tower. at dusk", "")}

    def encode(caps):
        feats = torch.stack([table[c] for c in caps]).to("cuda", torch.bfloat16)
        mask = torch.ones(len(caps), 16, dtype=torch.int64, device="cuda")
        mask[-1, 8:] = 0
        return feats, mask

    decoded = []

    def decode(lat):
        decoded.append(lat.clone())
        return torch.sigmoid(lat[:, :3].float())

    try:
        out = cli.infer("a tall tower. at dusk", str(tmp_path / "out"), ckpt=None, ckpt_lm=None, ema=False, precision="bf16",
                        config_path=str(settings), token=None, encode_fn=encode, cap_feat_dim=cfg.cap_feat_dim, decode_fn=decode)
    finally:
        del models.__dict__["NextDiT_tiny_cli"]
    assert os.path.exists(out) and os.path.basename(out).startswith("a_tall_tower_") and out.endswith("_lumina.png")
    model = _model(cfg, int(g["seed_w"]))
    torch.random.manual_seed(5)
    z = torch.randn([1, 4, 128 // 8, 64 // 8], device="cuda").to(torch.bfloat16).repeat(2, 1, 1, 1)
    feats, mask = encode(["a tall tower. at dusk", ""])
    fn = Sampler(create_transport("Linear", "velocity", "velocity", 0.2, 0.1)).sample_ode(sampling_method="euler", num_steps=3,
                                                                                         time_shifting_factor=4)
    want = fn(z, model.forward_with_cfg, cap_feats=feats, cap_mask=mask, cfg_scale=3.0, proportional_attn=True, base_seqlen=16,
              scale_factor=math.sqrt(64 * 128 / 64 ** 2), scale_watershed=0.3)[-1][:1]
    assert len(decoded) == 1 and torch.equal(decoded[0], want / 0.13025)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_engine_packed_variable_resolution_vs_reference_list_path(golden_dir, dtype):
    """NextDiT.forward(list of [C, H_b, W_b]) (reference model.py:789-834, unpatchify :757-768) through lt_forward_packed
    against the unmodified reference's list path (CPU fp32 golden): mixed lengths 64 / 60 / 48 / 64 tokens, pad_token rows,
    per-sample rotary grid width, masked padded keys, per-sample output shapes; plain and proportional attention
    (the proportional scale uses the padded length, model.py:373-374)."""
    g, cfg = _golden(golden_dir, "nextdit_tiny_packed")
    model = _model(cfg, int(g["seed_w"]))
    sizes = [tuple(int(v) for v in hw) for hw in g["sizes"]]
    xs = [torch.from_numpy(g[f"x{b}"]).to("cuda", dtype) for b in range(len(sizes))]
    t, cap = torch.from_numpy(g["t"]).cuda(), torch.from_numpy(g["cap"]).to("cuda", torch.bfloat16)
    mask = torch.from_numpy(g["mask"]).cuda()
    ys = model(xs, t, cap, mask)
    assert isinstance(ys, list) and len(ys) == len(sizes)
    for b, y in enumerate(ys):
        ref = torch.from_numpy(g[f"y{b}"])
        assert tuple(y.shape) == (cfg.in_channels,) + sizes[b] and y.dtype == dtype
        assert rel_l2(y, ref) < TOL_FWD, (b, rel_l2(y, ref))
    # a sample of the maximum length is bit-identical to running it alone as a [1, C, H, W] tensor (same kernels, same order)
    solo = model(xs[0][None], t[:1], cap[:1], mask[:1])[0]
    assert torch.equal(solo, ys[0])
    for layer in model.layers:  # as forward_with_cfg leaves them (model.py:891-899)
        layer.attention.proportional_attn, layer.attention.base_seqlen = True, 16
    yp = model(xs, t, cap, mask)
    for b, y in enumerate(yp):
        ref = torch.from_numpy(g[f"yprop{b}"])
        assert rel_l2(y, ref) < TOL_FWD, (b, rel_l2(y, ref))
    short = min(range(len(sizes)), key=lambda b: sizes[b][0] * sizes[b][1])
    assert rel_l2(yp[short], torch.from_numpy(g[f"solo_prop{short}"])) > rel_l2(yp[short], torch.from_numpy(g[f"yprop{short}"]))
    with pytest.raises(TypeError):
        model.forward_with_cfg(xs, t, cap, mask, 4.0)


def test_hip_graph_replay_is_bit_identical_to_eager_launches(golden_dir):
    """lt_set_option("graph", 1) (default): the third evaluation with one set of arguments replays a captured HIP graph of the
    ~250 launches.  Replays must be bit-identical to eager launches, pick up NEW inputs / prompts / weights
    (contents sit behind fixed pointers), and a change of arguments must start a new graph, not reuse the old one."""
    from gpu_util import set_option
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    t = torch.from_numpy(g["t"]).cuda()
    cap, mask = torch.from_numpy(g["cap"]).to("cuda", torch.bfloat16), torch.from_numpy(g["mask"]).cuda()
    gen = torch.Generator().manual_seed(4)
    xs = [torch.randn(1, 4, 16, 16, generator=gen).repeat(2, 1, 1, 1).to("cuda", torch.bfloat16) for _ in range(5)]
    kw = dict(base_seqlen=16, proportional_attn=True)

    def runs():
        outs = [model.forward_with_cfg(x, t, cap, mask, 4.0, **kw) for x in xs]                  # one key, five inputs
        outs.append(model.forward_with_cfg(xs[0], t, cap, mask, 2.5, **kw))                     # other cfg scale: other key
        outs.append(model.forward_with_cfg(xs[1], t * 0.5, cap, mask, 4.0, **kw))               # other time: same key, new t
        cap2 = (cap * 0.5).contiguous()
        outs.append(model.forward_with_cfg(xs[2], t, cap2, mask, 4.0, **kw))                    # new prompt, same shapes: same key
        fn = Sampler(create_transport()).sample_ode(sampling_method="midpoint", num_steps=4, time_shifting_factor=4)
        outs.append(fn(xs[3], model.forward_with_cfg, cap_feats=cap, cap_mask=mask, cfg_scale=4.0, **kw))
        return outs

    try:
        set_option("graph", 1)
        before = model._engine.graph_replays() if model._engine is not None else 0
        with_graph = runs()
        replays = model._engine.graph_replays() - before
        set_option("graph", 0)
        eager = runs()
        assert model._engine.graph_replays() - before == replays  # no replay while switched off
    finally:
        set_option("graph", 2)
    assert replays >= 8, replays  # 5 + 1 + 1 + 6 evaluations on the main key, minus warm-up and capture
    for a, b in zip(with_graph, eager):
        assert torch.equal(a, b)


def test_engine_options_are_per_engine_and_do_not_leak(golden_dir):
    """SURVEY.md 8b "no hidden global state" (VERDICT r4 item 8): lt_engine_set_option overrides an option for ONE engine.  Two engines of
    the same model in one process: A runs with the HIP graph switched off for itself only, B follows the process default - B replays,
    A never does, results are bit-identical; dropping the override (LT_OPTION_INHERIT) hands A back to the default; a process-default
    change moves only the engine without an override."""
    from gpu_util import set_option
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    a, b = _model(cfg, int(g["seed_w"])), _model(cfg, int(g["seed_w"]))
    z, t, cap, mask = _inputs(g)
    kw = dict(base_seqlen=16, proportional_attn=True)
    run = lambda m: [m.forward_with_cfg(z, t, cap, mask, 4.0, **kw) for _ in range(5)][-1]
    try:
        set_option("graph", 1)  # (the default, 2, replays above 1024 rows only: this model has 128)
        run(a), run(b)  # engines exist now
        ea, eb = a._engine, b._engine
        assert ea.get_option("graph") == 1 and eb.get_option("graph") == 1
        ea.set_option("graph", 0)
        assert ea.get_option("graph") == 0 and eb.get_option("graph") == 1
        ra, rb = ea.graph_replays(), eb.graph_replays()
        oa, ob = run(a), run(b)
        assert ea.graph_replays() == ra and eb.graph_replays() > rb
        assert torch.equal(oa, ob)
        with pytest.raises(Exception, match="must be"):
            ea.set_option("gemm_splitk", 7)  # range-checked per engine as well
        set_option("graph", 0)             # process default off: B (no override) stops replaying ...
        rb = eb.graph_replays()
        run(b)
        assert eb.graph_replays() == rb
        ea.set_option("graph", 1)          # ... while A's own override says on
        ra = ea.graph_replays()
        run(a)
        assert ea.graph_replays() > ra
        set_option("graph", 1)
        ea.set_option("graph", None)           # back to inheriting
        assert ea.get_option("graph") == 1
        ea.set_option("attention_variant", 3)  # another kernel for A only: B's output is untouched, A's agrees (bit-equal by test_attention_v4_*)
        assert eb.get_option("attention_variant") == 4
        assert torch.equal(run(b), ob)
        ea.set_option("attention_variant", None)
        # the default: replay above 1024 rows only - this 128-row model runs plain launches under it
        set_option("graph", 2)
        rb = eb.graph_replays()
        assert torch.equal(run(b), ob) and eb.graph_replays() == rb
    finally:
        set_option("graph", 2)


def test_hip_graph_replay_rebuilds_the_shared_rope_table_when_keys_alternate(golden_dir):
    """the RoPE table is one buffer outside every graph.  Key A (scale_factor 1) captured and replayed, then key B
    (scale_factor 2) rebuilds the table, then key A again: its cached graph must not replay against B's table."""
    from gpu_util import set_option
    g, cfg = _golden(golden_dir, "nextdit_tiny")
    model = _model(cfg, int(g["seed_w"]))
    z, t, cap, mask = _inputs(g)
    t_hi = torch.full((2,), 0.8, device="cuda")
    kwA = dict(base_seqlen=16, proportional_attn=True, scale_factor=1.0, scale_watershed=1.0)
    kwB = dict(base_seqlen=16, proportional_attn=True, scale_factor=2.0, scale_watershed=0.3)

    def runs():
        outs = [model.forward_with_cfg(z, t_hi, cap, mask, 4.0, **kwA) for _ in range(3)]
        outs += [model.forward_with_cfg(z, t_hi, cap, mask, 4.0, **kwB) for _ in range(3)]
        outs += [model.forward_with_cfg(z, t_hi, cap, mask, 4.0, **kwA) for _ in range(2)]
        outs += [model.forward_with_cfg(z, t_hi, cap, mask, 4.0, **kwB) for _ in range(2)]
        return outs

    try:
        set_option("graph", 1)
        before = model._engine.graph_replays() if model._engine is not None else 0
        with_graph = runs()
        replays = model._engine.graph_replays() - before
        set_option("graph", 0)
        eager = runs()
    finally:
        set_option("graph", 2)
    assert replays >= 6, replays
    assert not torch.equal(eager[0], eager[3])  # the two keys really differ (NTK branch at t = 0.8)
    for i, (a, b) in enumerate(zip(with_graph, eager)):
        assert torch.equal(a, b), i
    assert torch.equal(with_graph[6], with_graph[0]) and torch.equal(with_graph[8], with_graph[3])


@pytest.mark.gpu
def test_attention_variants_3_and_4_agree_in_the_model():
    """the default attention kernel (variant 4: one wave per SIMD, text phase through the same tile pipeline) and the ping-pong
    kernel (variant 3) run the same arithmetic in the same order: a whole forward_with_cfg - self-attention at 4096 tokens + fused
    text cross-attention with a padded caption - must not change by a bit"""
    from gpu_util import set_option
    cfg = synth.NextDiTConfig(dim=1152, n_layers=2, n_heads=16, cap_feat_dim=256)
    sd = synth.synth_state_dict(cfg, seed=73)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(128, 128), text_len=100, uncond_len=8, seed=74)
    model = models.NextDiT(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    zb, capb = z.to("cuda", torch.bfloat16), cap.to("cuda", torch.bfloat16)
    outs = {}
    try:
        set_option("attn_q_fused", 0)  # (both kernels on head-major queries from qk_norm_rope: the comparison is about the kernels)
        for v in (4, 3):
            set_option("attention_variant", v)
            outs[v] = model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 4.0, base_seqlen=4096, proportional_attn=True)
    finally:
        set_option("attention_variant", 4)
        set_option("attn_q_fused", 1)
    assert torch.equal(outs[4], outs[3]), rel_l2(outs[4], outs[3])


@pytest.mark.gpu
def test_pair_layout_is_bit_identical_and_survives_a_change_of_regime():
    """round 6, option pair_layout: at the 2B widths and 4096 tokens (M = 8192: every GEMM of the block on the persistent kernel) the
    engine keeps the GEMMs' A operands and the four weights of every layer in the row-pair-interleaved layout.  Same products in the
    same order -> the output must not change by a bit, eager and replayed.  A 256-token call in between runs the small-M kernels, which
    read row-major weights: the engine converts back and forth in place, and both shapes keep giving the same bits."""
    from gpu_util import set_option
    cfg = synth.NextDiTConfig(n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=81)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(128, 128), text_len=128, uncond_len=8, seed=82)
    zs, ts, _, _ = synth.synth_inputs(cfg, latent_hw=(32, 32), text_len=128, uncond_len=8, seed=83)
    model = models.NextDiT(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    zb, zsb, capb = z.to("cuda", torch.bfloat16), zs.to("cuda", torch.bfloat16), cap.to("cuda", torch.bfloat16)
    big = lambda: model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 4.0, base_seqlen=4096, proportional_attn=True).clone()
    small = lambda: model.forward_with_cfg(zsb, ts.cuda(), capb, mask.cuda(), 4.0, base_seqlen=4096, proportional_attn=True).clone()
    outs = {}
    try:
        for v in (0, 1):
            set_option("pair_layout", v)
            seq = [big(), big(), big(), big(), small(), big(), small(), small(), big()]  # (third call of a key on: graph replays)
            outs[v] = seq
    finally:
        set_option("pair_layout", 1)
    for i, (a, b) in enumerate(zip(outs[0], outs[1])):
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a, b), (i, rel_l2(b, a))
    for i in (1, 2, 3, 5, 8):
        assert torch.equal(outs[1][i], outs[1][0]), i
    assert torch.equal(outs[1][6], outs[1][4]) and torch.equal(outs[1][7], outs[1][4])
    # the weights are in the pair layout now (the last call was a large one): an upload converts them back first, and the next large
    # evaluation converts the NEW weights again
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda", torch.bfloat16)
    assert torch.equal(small(), outs[1][4])
    assert torch.equal(big(), outs[1][0])
    sd2 = {k: (v * 1.5 if k.endswith("attention.wo.weight") else v) for k, v in sd.items()}
    model.load_state_dict(sd2, strict=True)
    model = model.to("cuda", torch.bfloat16)
    changed = big()
    assert torch.isfinite(changed.float()).all() and not torch.equal(changed, outs[1][0])
    set_option("pair_layout", 0)
    try:
        assert torch.equal(big(), changed)
    finally:
        set_option("pair_layout", 1)


@pytest.mark.parametrize("opt", ["qk_post_pair", "qkv_vt_epilogue", "qkv_fused_gemm", "gemm_w4q", "norm_specialize", "attn_q_fused"])
def test_engine_path_switches_do_not_change_results(opt):
    """the launch-structure options of the engine (q / k post-processing in one launch, V projection with the V^T epilogue,
    persistent 16x16x32 GEMM, specialised row kernels) on a model wide and long enough to take those paths (d 1152, 4096 tokens,
    M = 8192 rows): each switch off vs on.  Row / layout kernels must be bit-identical; the GEMM switch changes the fp32 summation order
    inside a 32-deep slab, so there the outputs agree to bf16 rounding."""
    from gpu_util import set_option
    cfg = synth.NextDiTConfig(dim=1152, n_layers=2, n_heads=16, cap_feat_dim=256)
    sd = synth.synth_state_dict(cfg, seed=71)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(128, 128), text_len=64, uncond_len=8, seed=72)
    model = models.NextDiT(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    zb, capb = z.to("cuda", torch.bfloat16), cap.to("cuda", torch.bfloat16)
    outs = {}
    on, default = 1, 1
    # the prologue form of q_norm + RoPE exists only behind the fused QKV launch: a switch that takes that launch away would also move the
    # queries to qk_norm_rope (other summation order of the LayerNorm statistics) - those switches are compared with the prologue off
    pin_q = opt in ("qkv_vt_epilogue", "qkv_fused_gemm", "gemm_w4q")
    try:
        if pin_q:
            set_option("attn_q_fused", 0)
        for v in (on, 0):
            set_option(opt, v)
            outs[1 if v else 0] = model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 4.0, base_seqlen=4096, proportional_attn=True)
    finally:
        set_option(opt, default)
        set_option("attn_q_fused", 1)
    if opt in ("gemm_w4q", "qkv_fused_gemm"):  # (fused QKV: the V columns move from the 32x32x16 classic kernel to the 16x16x32 one)
        assert rel_l2(outs[1], outs[0]) < 1e-2, rel_l2(outs[1], outs[0])
    elif opt == "attn_q_fused":
        # round 4: q_norm + RoPE of the queries in the attention prologue instead of qk_norm_rope - the same arithmetic per element, but
        # the row's mean / variance come from (sum, sum of squares) partials in another summation order: a few queries land one bf16
        # ulp away, which the softmax and 2 layers carry to the output at the 1e-3 level
        assert rel_l2(outs[1], outs[0]) < 6e-3, rel_l2(outs[1], outs[0])
        # and the other queries' path must really have been taken: with the option off the result equals the kernel-variant test's
        set_option("qk_post_pair", 0)
        try:
            again = model.forward_with_cfg(zb, t.cuda(), capb, mask.cuda(), 4.0, base_seqlen=4096, proportional_attn=True)
        finally:
            set_option("qk_post_pair", 1)
        assert torch.equal(again, outs[1])  # with the prologue path on, how q / k post-processing is launched cannot matter for q
    else:
        assert torch.equal(outs[1], outs[0]), rel_l2(outs[1], outs[0])
