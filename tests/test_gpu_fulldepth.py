"""-m gpu: COMPLETE models of the BASELINE configs (every layer, every width) on the engine against full-depth fixtures made
from the unmodified reference modules (oracle/make_fulldepth_golden.py -> tests/golden/full_*.npz):

  full_2b          configs[1]  Lumina-Next-T2I 2B, 24 layers, 1024^2 (4096 tokens), T = 128, CFG 4, proportional attention
  full_2b_gqa_ntk  configs[3]  Lumina-Next-SFT 2B GQA (32 / 8 heads), 24 layers, time-aware RoPE scaling (scale_factor 2): NTK branch
                               (t = 0.6) and linear-interpolation branch (t = 0.1), 4096 tokens as a 32 x 128 patch grid (the config's
                               column positions 0..127; its 16 384 tokens do not fit the fp32 reference's explicit mask in 62 GB)
  full_flag5b      configs[2]  Lumina-T2I 5B Flag-DiT, 32 layers, d 3072, hd 96, the config's own 64 x 65 = 4160 tokens incl. eol
  full_moe600m     configs[4]  Next-DiT-MoE 600M "Both", 16 layers, 4 + 4 experts, 1024 tokens; full_moe600m_256: the config's own 256
                               tokens.  Both also with the discrete routing held equal to the reference's (routing-pinned gate)
  full_moe600m_4096  configs[4] AS WORDED ("1024x1024"): the same weights at 4096 tokens = 8192 rows, 16 384 routed rows per MoE FFN
                               (the grouped mode of the persistent GEMM kernel), free-running and routing-pinned gates
  full_imagenet600m  configs[0]  Next-DiT-ImageNet 600M, 16 layers, 256 tokens, vs the unmodified Next-DiT-ImageNet/models/models.py
  full_2b_gqa_16k  configs[3] AT ITS OWN SIZE: 16 384 tokens (128 x 128 patches), 24 layers, both RoPE branches.  No reference output at
                               this size (see above): `oracle_*` of the restatement that full_2b_gqa_ntk pins bit for bit on the same weights

Gate: engine(bf16) vs reference(fp32) <= 1.5 x [reference's own bf16 choreography vs its fp32
self] on the same draw, for all channels and for the unguided channel 3; the yardstick (`floor_*`) is stored next to the reference
output, so the GPU box spends no CPU-oracle time.  Weights / inputs are regenerated from the fixture's seeds and checked against
its checksums first: if the draw does not reproduce (different numpy), the test computes reference values with the live oracle.
"""
import json
import os

import numpy as np
import pytest
import torch

import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import models
from oracle import synth

from gpu_util import rel_l2

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return g, synth.NextDiTConfig(**json.loads(str(g["config"])))


def _draw(g, cfg):
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]), streams=True)
    keys = json.loads(str(g["wkeys"]))
    wsum = np.array([float(sd[k].double().abs().sum()) for k in keys[:3]])
    same = np.allclose(wsum, g["wsum"], rtol=1e-12) and np.array_equal(sd[keys[3]].flatten()[:8].double().numpy(), g["wprobe"])
    return sd, bool(same)


def _inputs(g, cfg, tv):
    hw = tuple(int(v) for v in g["latent_hw"])
    if cfg.has_text:
        return synth.synth_inputs(cfg, latent_hw=hw, text_len=int(g["text_len"]), uncond_len=int(g["uncond_len"]), seed=int(g["seed_x"]),
                                  t_value=tv)
    return synth.synth_inputs(cfg, latent_hw=hw, seed=int(g["seed_x"]), t_value=tv)


def _live_reference(cfg, sd, ins, kw):
    """fallback when the weight draw does not reproduce the fixture: the CPU oracle, fp32 and bf16-choreography"""
    from oracle import nextdit_oracle as O
    from oracle import variants_oracle as V
    fn = {"next_t2i": O.forward_with_cfg, "flag_t2i": V.flag_forward_with_cfg}.get(cfg.family, V.imagenet_forward_with_cfg)
    ins = list(ins)
    ins[0] = ins[0].to(torch.bfloat16).float()
    if cfg.has_text:
        ins[2] = ins[2].to(torch.bfloat16).float()
    return fn(sd, cfg, *ins, **kw), fn(sd, cfg, *ins, bf16=True, **kw).float()


def _check(name, golden_dir, ctor, model=None, keep=False):
    g, cfg = _load(golden_dir, name)
    if model is None:
        sd, same = _draw(g, cfg)
        model = ctor(cfg)
        model.load_state_dict(sd, strict=True)
        model = model.eval().to("cuda", torch.bfloat16)
        model._fixture_draw = (sd if not same else None, same)
    sd, same = model._fixture_draw
    report = []
    for tag, tv, kw in json.loads(str(g["calls"])):
        ins = _inputs(g, cfg, tv)
        if same:  # (`ref_*` absent: a size the reference module cannot run - the fixture holds the pinned restatement's output)
            ref, floor = torch.from_numpy(g[f"ref_{tag}" if f"ref_{tag}" in g.files else f"oracle_{tag}"]), torch.from_numpy(g[f"floor_{tag}"])
        else:
            ref, floor = _live_reference(cfg, sd, ins, kw)
        kw = dict(kw)
        scale = kw.pop("cfg_scale")
        zb = ins[0].to("cuda", torch.bfloat16)
        if cfg.has_text:
            got = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].to("cuda", torch.bfloat16), ins[3].cuda(), scale, **kw)
        else:
            got = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), scale, **kw)
        got = got.float().cpu()
        assert got.shape == ref.shape and torch.isfinite(got).all()
        assert torch.equal(got[0, :3], got[1, :3])
        f_all, f_c3 = rel_l2(floor, ref), rel_l2(floor[:, 3], ref[:, 3])
        pin, e_rb = "", None
        if same and f"refbf16_{tag}" in g.files:
            # round 6: the yardstick is the SMALLEST of three bf16 realisations of the reference on this draw - the builder's emulation
            # (floor_*) and the unmodified reference module in bf16, plain and under autocast (refbf16_* / refbf16ac_*; tests/test_oracle_golden.py
            # holds the three within 10-15 % of each other)
            rb, ra = torch.from_numpy(g[f"refbf16_{tag}"]), torch.from_numpy(g[f"refbf16ac_{tag}"])
            r_all, r_c3 = min(rel_l2(rb, ref), rel_l2(ra, ref)), min(rel_l2(rb[:, 3], ref[:, 3]), rel_l2(ra[:, 3], ref[:, 3]))
            pin = (f"; reference module in bf16 vs fp32 {rel_l2(rb, ref):.3e} / autocast {rel_l2(ra, ref):.3e} (ch3 {rel_l2(rb[:, 3], ref[:, 3]):.3e} / "
                   f"{rel_l2(ra[:, 3], ref[:, 3]):.3e}); engine vs the reference module in bf16 {rel_l2(got, ra):.3e}")
            f_all, f_c3 = min(f_all, r_all), min(f_c3, r_c3)
            e_rb = rel_l2(got, ra)
        e_all, e_c3, e_floor = rel_l2(got, ref), rel_l2(got[:, 3], ref[:, 3]), rel_l2(got, floor)
        report.append((tag, e_all, e_c3, f_all, f_c3, e_floor, e_rb))
        print(f"{name}/{tag}: engine vs reference fp32 {e_all:.3e} (ch3 {e_c3:.3e}); reference bf16 choreography vs fp32 {f_all:.3e} "
              f"(ch3 {f_c3:.3e}); engine vs bf16 choreography {e_floor:.3e}; fixture draw reproduced: {same}" + pin)
        assert e_all < 1.5 * f_all and e_c3 < 1.5 * f_c3, (name, tag, e_all, f_all, e_c3, f_c3)
        if same and f"route_{tag}" in g.files:
            _check_moe_routing_pinned(name, tag, g, model, zb, ins, scale, kw, ref)
    if keep:
        return report, model
    del model
    torch.cuda.empty_cache()
    return report


def _check_moe_routing_pinned(name, tag, g, model, zb, ins, scale, kw, ref):
    """VERDICT r2 item 3a.  A free-running MoE comparison passes at 30 % error because one rounded router logit replaces a token's
    whole expert output - it cannot tell a routing BUG from routing noise.  Two sharper gates:
      * agreement: the engine's own top-2 selections (lt_moe_routing_record) against the fp32 reference's, per (layer, branch,
        row) - must not be worse than the reference's own bf16 choreography's agreement by more than 3 points;
      * routing-pinned: the reference's selections forced onto the engine (lt_moe_routing_force; softmax weights, expert GEMMs,
        combine stay the engine's own) -> the dense models' gate, 1.5 x [bf16 choreography with the same routing forced vs fp32]."""
    eng = model._engine
    route = g[f"route_{tag}"].astype(np.int32)
    rows = route.shape[2]
    eng.moe_routing_record(True)
    try:
        free = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), scale, **kw)
        mine = eng.moe_routing_read(rows)
    finally:
        eng.moe_routing_record(False)
    ran = route >= 0
    assert ((mine >= 0) == ran).all()
    agree = float(((mine == route) | ~ran).all(axis=-1).mean())
    floor_agree = float(g[f"floor_agree_{tag}"])
    eng.moe_routing_force(route)
    try:
        forced = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), scale, **kw).float().cpu()
        eng.moe_routing_record(True)
        model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), scale, **kw)
        assert np.array_equal(eng.moe_routing_read(rows), route)  # the hook really drove every layer
    finally:
        eng.moe_routing_record(False)
        eng.moe_routing_force(None)
    floor_forced = torch.from_numpy(g[f"floor_forced_{tag}"])
    f_all, f_c3 = rel_l2(floor_forced, ref), rel_l2(floor_forced[:, 3], ref[:, 3])
    e_all, e_c3 = rel_l2(forced, ref), rel_l2(forced[:, 3], ref[:, 3])
    print(f"{name}/{tag} routing: engine agrees with the fp32 reference on {agree * 100:.2f} % of selections (reference bf16 choreography: "
          f"{floor_agree * 100:.2f} %); routing pinned: engine vs reference fp32 {e_all:.3e} (ch3 {e_c3:.3e}), bf16 choreography pinned "
          f"{f_all:.3e} (ch3 {f_c3:.3e})")
    assert agree > floor_agree - 0.03, (agree, floor_agree)
    # with the discrete choice held equal the reference's own bf16-vs-fp32 distance drops from 0.20 to 0.11 (unguided channel 0.082
    # -> 0.053); what is left above the dense models' 0.03 is continuous: the two gate weights are a softmax over bf16-rounded
    # router logits, 32 times per forward.  The gate below is therefore 0.17 instead of the free-running 0.30.
    free_floor = rel_l2(torch.from_numpy(g[f"floor_{tag}"]), ref)
    assert f_all < 0.7 * free_floor, (f_all, free_floor)
    assert e_all < 1.5 * f_all and e_c3 < 1.5 * f_c3, (name, tag, e_all, f_all, e_c3, f_c3)
    back = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), scale, **kw)
    assert torch.equal(back, free)  # hooks off: the free-running result is back, bit for bit


def test_full_2b_24_layers_vs_reference(golden_dir):
    """BASELINE configs[1], the bench workload: NextDiT_2B_patch2, all 24 layers, 4096 tokens (model.py:836-913)"""
    _check("full_2b", golden_dir, lambda cfg: models.NextDiT_2B_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))


def test_full_2b_gqa_24_layers_ntk_and_linear_rope_vs_reference(golden_dir):
    """BASELINE configs[3]: NextDiT_2B_GQA_patch2, all 24 layers, scale_factor 2 with both branches of the time-aware RoPE
    (model.py:944-952: t < watershed -> linear interpolation, else NTK)"""
    rep = _check("full_2b_gqa_ntk", golden_dir, lambda cfg: models.NextDiT_2B_GQA_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))
    assert {r[0] for r in rep} == {"ntk", "lin"}


def test_full_flag_dit_5b_32_layers_vs_reference(golden_dir):
    """BASELINE configs[2]: DiT_Llama_5B_patch2 (lumina_t2i/models/model.py:866-922), all 32 layers, d 3072, hd 96, eol tokens"""
    _check("full_flag5b", golden_dir, lambda cfg: models.flag_dit.DiT_Llama_5B_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))


def test_full_moe_600m_16_layers_vs_reference(golden_dir):
    """BASELINE configs[4]: DiT_Llama_600M_patch2_Both (Next-DiT-MoE/models/models2.py:850-958), all 16 layers, 4 + 4 experts.
    Routing is discrete, so the reference's own bf16 path differs from its fp32 self by ~0.2 here (a rounded router logit
    replaces a token's expert); the gate is 1.5 x that, like everywhere else."""
    ctor = lambda cfg: models.moe.DiT_Llama_600M_patch2_Both(qk_norm=True, num_classes=cfg.num_classes)
    _, model = _check("full_moe600m", golden_dir, ctor, keep=True)
    # the same weights (same seed) at the config's own 256 tokens: the small-M GEMM tiles and 2-tile attention loops
    g1, g2 = _load(golden_dir, "full_moe600m")[0], _load(golden_dir, "full_moe600m_256")[0]
    assert int(g1["seed_w"]) == int(g2["seed_w"]) and np.array_equal(g1["wsum"], g2["wsum"])
    _, model = _check("full_moe600m_256", golden_dir, ctor, model=model, keep=True)
    # round 5: the MoE row kernel (top-2 combine + gated residual + next pre-norm) runs mode-specialised instantiations at d = 1536 - the
    # same statements in the same order as the generic kernel, so the whole 16-layer forward must not move by a bit
    from gpu_util import set_option
    cfg2 = synth.NextDiTConfig(**json.loads(str(g2["config"])))
    ins = _inputs(g2, cfg2, 0.5)
    zb = ins[0].to("cuda", torch.bfloat16)
    spec = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), 4.0)
    try:
        set_option("norm_specialize", 0)
        generic = model.forward_with_cfg(zb, ins[1].cuda(), ins[2].cuda(), 4.0)
    finally:
        set_option("norm_specialize", 1)
    assert torch.equal(spec, generic)
    # BASELINE configs[4] as worded, 1024 x 1024: 4096 tokens, 16 384 routed rows per MoE FFN on the grouped persistent GEMM
    g3 = _load(golden_dir, "full_moe600m_4096")[0]
    assert int(g1["seed_w"]) == int(g3["seed_w"]) and np.array_equal(g1["wsum"], g3["wsum"]) and tuple(g3["latent_hw"]) == (128, 128)
    _check("full_moe600m_4096", golden_dir, ctor, model=model)


def test_full_imagenet_600m_16_layers_vs_reference(golden_dir):
    """BASELINE configs[0]: DiT_Llama_600M_patch2 (Next-DiT-ImageNet/models/models.py:759-974), all 16 layers, 256 tokens, against a
    stored output of the unmodified reference module (round 3 had only the live oracle at full depth)"""
    _check("full_imagenet600m", golden_dir, lambda cfg: models.imagenet.DiT_Llama_600M_patch2(qk_norm=True, num_classes=cfg.num_classes))


def test_full_2b_gqa_16k_tokens_24_layers_vs_pinned_restatement(golden_dir):
    """BASELINE configs[3] at its own 2048 x 2048: 16 384 tokens, 24 layers, GQA 32 / 8, 256 key tiles per query block, RoPE positions
    to 127 on both axes, NTK and linear-interpolation branches (lumina_next_t2i_mini/models/nextdit.py:358-361,
    lumina_next_t2i/models/model.py:944-952)"""
    rep = _check("full_2b_gqa_16k", golden_dir, lambda cfg: models.NextDiT_2B_GQA_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))
    assert {r[0] for r in rep} == {"ntk", "lin"}
    # round 6: a REFERENCE-held output at this size - the unmodified mini module's flash branch in bf16
    # (lumina_next_t2i_mini/models/nextdit.py:328-357; `refbf16ac_*`).  Engine and reference module are two bf16 realisations of the same
    # function: their distance at 16 384 tokens must not exceed 1.5 x the distance of two bf16 realisations of the SAME weights at 4096 tokens
    # (builder's emulation vs the reference module in bf16, both stored in full_2b_gqa_ntk)
    g16, g4 = _load(golden_dir, "full_2b_gqa_16k")[0], _load(golden_dir, "full_2b_gqa_ntk")[0]
    if "refbf16ac_ntk" in g16.files and "refbf16ac_ntk" in g4.files:
        for tag, *_, e_rb in rep:
            if e_rb is None:
                continue
            d4 = rel_l2(torch.from_numpy(g4[f"floor_{tag}"]), torch.from_numpy(g4[f"refbf16ac_{tag}"]))
            print(f"full_2b_gqa_16k/{tag}: engine vs the reference module in bf16 at 16 384 tokens {e_rb:.3e}; two bf16 realisations at 4096 tokens {d4:.3e}")
            assert e_rb <= 1.5 * d4, (tag, e_rb, d4)


# ---- whole TRAJECTORIES at full depth (fixtures: oracle/make_traj_golden.py) ----------------------------------
def _traj_check(name, golden_dir, ctor, gate=1.5):
    """The reference's product is the latent after a whole flow-matching trajectory (sample.py:216-234).  Fixture: the UNMODIFIED reference
    sampler + model in fp32 (`ref_*`) and the reference's bf16 choreography with a bf16 state over the same grid (`floor_*`,
    `drift_floor`).  Here: the engine's trajectory through the public Sampler (-> ONE lt_sample_ode call, bf16 state), t cast to the
    state dtype as torchdiffeq does (default) and not cast; gate: final-state rel-L2 vs the reference <= 1.5 x the floor's, on all rows and
    on row 0 (the sample the reference decodes); the per-grid-point drift is printed beside the floor's ("report
    trajectory drift separately from per-NFE error")."""
    from lumina_t2x_amd.transport import Sampler, create_transport
    path = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (oracle/make_traj_golden.py: hours of CPU in the authoring container)")
    g = np.load(path, allow_pickle=False)
    cfg = synth.NextDiTConfig(**json.loads(str(g["config"])))
    sd, same = _draw(g, cfg)
    if not same:
        pytest.skip("the seeded weight draw does not reproduce on this numpy: a live CPU trajectory at full depth would take hours")
    model = ctor(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    del sd
    ins = _inputs(g, cfg, 0.5)
    kw = json.loads(str(g["model_kw"]))
    if cfg.has_text:
        kw.update(cap_feats=ins[2].to("cuda", torch.bfloat16), cap_mask=ins[3].cuda())
    else:
        kw.update(y=ins[2].cuda())
    z0 = torch.from_numpy(g["z0"]).repeat(2, 1, 1, 1).to("cuda", torch.bfloat16)
    assert torch.equal(z0[0].float().cpu(), ins[0][0].to(torch.bfloat16).float())  # the fixture's start IS the per-NFE fixture's draw
    method, num_steps, shift = str(g["method"]), int(g["num_steps"]), float(g["shift"])
    ref_final, floor_final = torch.from_numpy(g["ref_final"]), torch.from_numpy(g["floor_final"])
    pts = [int(p) for p in g["points"]]
    ref_pts, floor_pts = torch.from_numpy(g["ref_points"]), torch.from_numpy(g["floor_points"])
    f_all, f_r0 = rel_l2(floor_final, ref_final), rel_l2(floor_final[0], ref_final[0])
    results = {}
    driver = str(g["driver"]) if "driver" in g.files else "sampler"
    for t_round in (True, False):
        if driver == "mini":  # the mini fork's flat ODE class (transport/mini.py = lumina_next_t2i_mini/transport.py:57-111)
            from lumina_t2x_amd.transport.mini import ODE
            solver = ODE(num_steps, method, shift if shift > 0 else None)
            fn = solver.sample
        else:
            fn = Sampler(create_transport()).sample_ode(sampling_method=method, num_steps=num_steps, time_shifting_factor=shift if shift > 0 else None)
            solver = fn.__self__
        assert np.array_equal(solver.t.numpy().astype(np.float32), g["grid"]), "time grid differs from the reference's"
        solver.t_round_to_state_dtype = t_round
        traj = fn(z0, model.forward_with_cfg, **kw).float().cpu()
        assert traj.shape == (num_steps,) + tuple(z0.shape) and torch.isfinite(traj).all()
        assert model._engine.last_nfe() == (num_steps - 1) * {"euler": 1, "midpoint": 2}[method]
        e_all, e_r0 = rel_l2(traj[-1], ref_final), rel_l2(traj[-1, 0], ref_final[0])
        curve = [(k, rel_l2(traj[k, 0], ref_pts[i]), rel_l2(floor_pts[i], ref_pts[i]), rel_l2(traj[k, 0], floor_pts[i])) for i, k in enumerate(pts)]
        print(f"{name} [t cast to the state dtype: {t_round}]: final state, engine vs reference fp32 {e_all:.3e} (row 0 {e_r0:.3e}); reference bf16 "
              f"choreography vs fp32 {f_all:.3e} (row 0 {f_r0:.3e}); engine vs bf16 choreography {rel_l2(traj[-1], floor_final):.3e}")
        print(f"{name} drift, row 0, grid point: engine | floor | engine-vs-floor   " +
              "  ".join(f"{k}: {e:.2e} | {f:.2e} | {d:.2e}" for k, e, f, d in curve))
        results[t_round] = (e_all, e_r0)
        # at every stored grid point the engine may not drift away faster than the reference's own bf16 path does
        for k, e, f, _ in curve:
            assert e < gate * f + 1e-3, (name, t_round, k, e, f)
    # the gate proper: the configuration the reference runs (torchdiffeq casts t to the state dtype)
    e_all, e_r0 = results[True]
    assert e_all < gate * f_all and e_r0 < gate * f_r0, (name, e_all, f_all, e_r0, f_r0)
    # without the cast the engine sees the exact grid (what the fp32 reference sees): it must not be WORSE than 1.5 x the floor either
    assert results[False][0] < gate * f_all, (name, results[False], f_all)
    del model
    torch.cuda.empty_cache()


def test_full_2b_euler_30_point_trajectory_vs_reference(golden_dir):
    """BASELINE configs[1] AS THE CLI RUNS IT: NextDiT_2B_patch2, 24 layers, 4096 tokens, 30-point flow-matching Euler grid with time shift
    4 = 29 forward_with_cfg evaluations (sample.py:216-234, integrators.py:97-116), against the unmodified reference's own trajectory"""
    _traj_check("full_2b_traj_euler30", golden_dir, lambda cfg: models.NextDiT_2B_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))


def test_full_2b_midpoint_10_point_trajectory_vs_reference(golden_dir):
    """the demo's default solver (demo.py:395-431): midpoint, two evaluations per interval; 10 grid points = 18 NFE at full depth"""
    _traj_check("full_2b_traj_midpoint10", golden_dir, lambda cfg: models.NextDiT_2B_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))


def test_full_2b_midpoint_30_point_trajectory_vs_reference(golden_dir):
    """the demo's recipe in full (demo.py:395-431): midpoint over the 30-point shifted grid = 58 NFE at full depth"""
    _traj_check("full_2b_traj_midpoint30", golden_dir, lambda cfg: models.NextDiT_2B_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))


def test_full_imagenet_600m_4_step_euler_trajectory_vs_reference(golden_dir):
    """BASELINE configs[0] in its own form: class-conditional 256^2, 4-step Euler ODE (5 grid points, no time shift), 16 layers, against
    the unmodified Next-DiT-ImageNet sampler + model"""
    _traj_check("full_imagenet600m_traj_euler5", golden_dir,
                lambda cfg: models.imagenet.DiT_Llama_600M_patch2(qk_norm=True, num_classes=cfg.num_classes))


def test_full_moe_600m_30_point_euler_trajectory_vs_reference(golden_dir):
    """BASELINE configs[4]'s model (Next-DiT-MoE "Both": 16 layers, 4 time + 4 space experts, top-2) at 256^2 over a 30-point Euler grid
    (29 NFE), against the unmodified Next-DiT-MoE sampler + models2.DiT_Llama.  Routing is discrete - a flipped near-tie replaces a
    token's expert outright in either bf16 path - so the trajectory drifts further than the dense models' (the floor ends at 0.35); the
    gate is the same 1.5 x floor."""
    _traj_check("full_moe600m_traj_euler30", golden_dir, lambda cfg: models.moe.DiT_Llama_600M_patch2_Both(qk_norm=True, num_classes=cfg.num_classes))


def test_full_moe_600m_trajectory_with_the_reference_routing_pinned(golden_dir):
    """the free-running MoE trajectory gate above is 1.5 x a floor that ENDS 0.35 from the reference - a real routing or plan
    regression could hide inside it.  Here the discrete choice of every NFE is held equal to the fp32 reference run's (`route_steps`,
    lt_moe_routing_force before each evaluation; softmax weights, plan, expert GEMMs and combine stay the engine's own), the Euler grid is
    stepped on the host through the public Sampler (a plain callable takes the host loop, same arithmetic as lt_sample_ode: bf16 state, t cast
    to the state dtype), and the gate is the dense models': 1.5 x [the bf16 choreography with the same routing pinned vs fp32] at every stored
    grid point and at the end - a floor three times tighter than the free-running one."""
    from lumina_t2x_amd.transport import Sampler, create_transport
    name = "full_moe600m_traj_euler30"
    path = os.path.join(golden_dir, name + ".npz")
    g = np.load(path, allow_pickle=False)
    if "route_steps" not in g.files:
        pytest.skip("fixture without route_steps (oracle/make_traj_golden.py --forced-routing)")
    cfg = synth.NextDiTConfig(**json.loads(str(g["config"])))
    sd, same = _draw(g, cfg)
    if not same:
        pytest.skip("the seeded weight draw does not reproduce on this numpy")
    model = models.moe.DiT_Llama_600M_patch2_Both(qk_norm=True, num_classes=cfg.num_classes)
    model.load_state_dict(sd, strict=True)
    model = model.eval().to("cuda", torch.bfloat16)
    del sd
    ins = _inputs(g, cfg, 0.5)
    kw = json.loads(str(g["model_kw"]))
    kw.update(y=ins[2].cuda())
    z0 = torch.from_numpy(g["z0"]).repeat(2, 1, 1, 1).to("cuda", torch.bfloat16)
    num_steps = int(g["num_steps"])
    route = g["route_steps"].astype(np.int32)
    assert route.shape[0] == num_steps - 1
    step = {"k": 0}
    model.forward_with_cfg(z0, torch.zeros(2, device="cuda"), **kw)  # (creates the engine: the hooks live on it)
    eng = model._engine

    def pinned(x, t, **k):
        eng.moe_routing_force(route[step["k"]])
        step["k"] += 1
        return model.forward_with_cfg(x, t, **k)

    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=num_steps, time_shifting_factor=None)
    assert np.array_equal(fn.__self__.t.numpy().astype(np.float32), g["grid"])
    try:
        traj = fn(z0, pinned, **kw).float().cpu()
    finally:
        eng.moe_routing_force(None)
    assert step["k"] == num_steps - 1 and torch.isfinite(traj).all()
    pts = [int(p) for p in g["points"]]
    ref_pts, ff_pts = torch.from_numpy(g["ref_points"]), torch.from_numpy(g["floor_forced_points"])
    ref_final, ff_final, free_final = torch.from_numpy(g["ref_final"]), torch.from_numpy(g["floor_forced_final"]), torch.from_numpy(g["floor_final"])
    curve = [(k, rel_l2(traj[k, 0], ref_pts[i]), rel_l2(ff_pts[i], ref_pts[i])) for i, k in enumerate(pts)]
    e_all, f_all, free_all = rel_l2(traj[-1], ref_final), rel_l2(ff_final, ref_final), rel_l2(free_final, ref_final)
    print(f"{name}, routing pinned to the fp32 reference's at every NFE: final state, engine vs reference fp32 {e_all:.3e}; bf16 choreography "
          f"pinned {f_all:.3e} (free-running floor {free_all:.3e}); drift row 0, grid point: engine | pinned floor   " +
          "  ".join(f"{k}: {e:.2e} | {f:.2e}" for k, e, f in curve))
    assert f_all < 0.7 * free_all  # the pinned floor is the sharper yardstick (0.21 against 0.355 at the end)
    for k, e, f in curve:
        assert e < 1.5 * f + 1e-3, (k, e, f)
    assert e_all < 1.5 * f_all, (e_all, f_all)
    del model
    torch.cuda.empty_cache()


def test_full_flag_dit_5b_10_point_euler_trajectory_vs_reference(golden_dir):
    """BASELINE configs[2]: Lumina-T2I 5B (Flag-DiT, 32 layers, d 3072, 4160 tokens incl. eol), CFG 4, proportional attention, time shift
    4 (lumina_t2i/demo.py:134-142), Euler over 10 grid points = 9 NFE, against the unmodified lumina_t2i sampler + model"""
    _traj_check("full_flag5b_traj_euler10", golden_dir, lambda cfg: models.flag_dit.DiT_Llama_5B_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))


def test_full_2b_gqa_time_aware_rope_10_point_euler_trajectory_vs_reference(golden_dir):
    """BASELINE configs[3]'s model (Lumina-Next-SFT 2B, GQA 32 / 8 heads, 24 layers) with time-aware RoPE scaling (scale_factor 2, watershed
    0.3): the trajectory starts on the linear-interpolation branch and crosses to the NTK branch between grid points 5 and 6 - decided on
    the device from t inside ONE lt_sample_ode call; 4096 tokens on a 64 x 256 latent, against the unmodified mini fork's sampler + model"""
    _traj_check("full_2b_gqa_ntk_traj_euler10", golden_dir, lambda cfg: models.NextDiT_2B_GQA_patch2(qk_norm=True, cap_feat_dim=cfg.cap_feat_dim))
