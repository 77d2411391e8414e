"""Full-depth TRAJECTORY fixtures (tests/golden/full_*_traj*.npz) - TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_traj_golden [case ...]      (authoring container: needs /root/reference; ~30 GB RAM; hours of CPU)

The per-NFE fixtures (oracle/make_fulldepth_golden.py) compare ONE evaluation of a complete model.  The reference's product is the
latent after a whole flow-matching trajectory: ``sample_fn(z, model.forward_with_cfg, **kw)[-1]`` (lumina_next_t2i/sample.py:216-234,
demo.py:395-431) with ``sample_fn = Sampler(transport).sample_ode(sampling_method, num_steps, time_shifting_factor)``
(transport/transport.py:355-391 -> integrators.py:80-116).  Each fixture here holds, for the same seeded draw as the per-NFE fixture
of that config:

  ref    the UNMODIFIED reference: its ``create_transport`` / ``Sampler.sample_ode`` driving its ``model.forward_with_cfg``, fp32 state,
         fp32 arithmetic, CPU.  ``torchdiffeq`` is absent everywhere (oracle/odeint_oracle.py header): the fixed-grid stepping under
         the reference's ``ode.sample`` is the restated ``odeint`` (euler = y + dt f(t, y); midpoint pinned by the reference's own
         in-tree midpoint_solver, tests/golden/solver_kat.npz)
  floor  the same grid stepped with the bf16-choreography restatement of the model and a bf16 state, t cast to the state dtype per
         call (what the reference does on a GPU under sample.py's ``dtype = bf16``): its distance to `ref` is the reference's own
         bf16-vs-fp32 TRAJECTORY drift on this draw, the yardstick the engine is gated against (1.5 x floor)

Stored: states at `points` (grid indices; row 0 = the sample the reference decodes, sample.py:233), the final state of both rows,
`drift_floor` = rel-L2(floor_k, ref_k) at EVERY grid point, and the state norms.  Weights / inputs are regenerated from seeds
(oracle.synth), checksummed like the per-NFE fixtures.  Every finished grid point is check-pointed under /tmp so that a killed run
resumes instead of starting over.
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

from oracle import make_fulldepth_golden as F  # noqa: E402
from oracle import odeint_oracle as OD  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
CKPT = os.environ.get("LT_TRAJ_CKPT", "/tmp/lt_traj_ckpt")

# name -> the per-NFE case whose draw it shares, solver, num_steps (grid points; NFE = (num_steps - 1) x evaluations per step),
# time_shifting_factor, stored grid indices
CASES = {
    # BASELINE configs[1] as the CLI runs it: 30 grid points = 29 Euler NFE, time shift 4 (sample.py:77-110 defaults + BASELINE)
    "full_2b_traj_euler30": dict(base="full_2b", method="euler", num_steps=30, shift=4.0, points=[1, 5, 10, 15, 20, 25, 29]),
    # the demo's default solver (demo.py:395-431): midpoint, 2 NFE per interval; 10 grid points = 18 NFE (the CPU budget of one round)
    "full_2b_traj_midpoint10": dict(base="full_2b", method="midpoint", num_steps=10, shift=4.0, points=[1, 3, 5, 7, 9]),
    # ... and the demo's recipe in full: 30 grid points = 58 NFE (three hours of CPU for the two legs)
    "full_2b_traj_midpoint30": dict(base="full_2b", method="midpoint", num_steps=30, shift=4.0, points=[1, 5, 10, 15, 20, 25, 29]),
    # BASELINE configs[0] in its own form: class-conditional 256^2, "4-step Euler ODE" = 5 grid points, no time shift
    # (Next-DiT-ImageNet/sample.py defaults)
    "full_imagenet600m_traj_euler5": dict(base="full_imagenet600m", method="euler", num_steps=5, shift=None, points=[1, 2, 3, 4]),
    # BASELINE configs[4]'s model (Next-DiT-MoE "Both", 16 layers, 4 time + 4 space experts) at the config's own 256^2: a 30-point Euler
    # grid, no time shift (Next-DiT-MoE/sample.py:131-137 passes no shift; its default solver is the adaptive dopri5 of the absent
    # torchdiffeq, so the fixed-grid solver of that package's own list stands in).  Routing is discrete: a flipped near-tie replaces a
    # token's expert outright in either bf16 path, which the drift of BOTH legs carries
    "full_moe600m_traj_euler30": dict(base="full_moe600m_256", method="euler", num_steps=30, shift=None, points=[1, 5, 10, 15, 20, 25, 29]),
    # BASELINE configs[2]: Lumina-T2I 5B (Flag-DiT, 32 layers, d 3072, 4160 tokens incl. eol), CFG 4, proportional attention, time shift 4
    # (lumina_t2i/demo.py:134-142): Euler over 10 grid points = 9 NFE (the CPU budget: ~3 min per fp32 evaluation in the authoring container)
    "full_flag5b_traj_euler10": dict(base="full_flag5b", method="euler", num_steps=10, shift=4.0, points=[1, 3, 5, 7, 9]),
    # BASELINE configs[3]'s model (Lumina-Next-SFT 2B, GQA 32 / 8 heads) with time-aware RoPE scaling (scale_factor 2, watershed 0.3: the
    # linear-interpolation branch below t = 0.3, the NTK branch above - a trajectory crosses it), 4096 tokens on a 64 x 256 latent, Euler
    # over 10 shifted grid points
    # over 10 shifted grid points - driven by the mini fork's own flat `ODE` class (lumina_next_t2i_mini/transport.py:57-111, sample.py:204),
    # whose grid runs to t = 1 exactly
    "full_2b_gqa_ntk_traj_euler10": dict(base="full_2b_gqa_ntk", method="euler", num_steps=10, shift=4.0, points=[1, 3, 5, 7, 9], driver="mini"),
}


def _ckpt_path(name, leg, k):
    return os.path.join(CKPT, f"{name}.{leg}.{k:03d}.npy")


class _Stepper:
    """wraps the model callable handed to the reference's Sampler: counts evaluations, prints progress; the ODE stepping itself stays
    inside the reference's ode.sample -> odeint"""

    def __init__(self, fn, name, leg):
        self.fn, self.name, self.leg, self.n, self.t0 = fn, name, leg, 0, time.time()

    def __call__(self, x, t, **kw):
        out = self.fn(x, t, **kw)
        self.n += 1
        print(f"  [{self.name}/{self.leg}] NFE {self.n}: t = {float(t[0]):.5f}, {time.time() - self.t0:.0f} s elapsed", flush=True)
        return out


def _stepwise(name, leg, grid, method, y0, call, t_cast):
    """fixed-grid trajectory one interval at a time through odeint_oracle.odeint (2-point sub-grids: the arithmetic of each interval
    is exactly what one odeint call over the whole grid performs), with a checkpoint per grid point"""
    os.makedirs(CKPT, exist_ok=True)
    states = [y0]
    for k in range(1, len(grid)):
        p = _ckpt_path(name, leg, k)
        if os.path.exists(p):
            states.append(torch.from_numpy(np.load(p)).to(y0.dtype))
            continue
        y = OD.odeint(call, states[-1], grid[k - 1:k + 1], method=method, t_cast=t_cast)[-1]
        np.save(p + ".tmp.npy", y.float().numpy())
        os.replace(p + ".tmp.npy", p)
        states.append(y)
    return torch.stack(states)


def run_case(name):
    case = CASES[name]
    base = F.CASES[case["base"]]
    cfg = base["cfg"]
    tag, tv, ckw = base["calls"][0]
    ckw = dict(ckw)
    method, num_steps, shift = case["method"], case["num_steps"], case["shift"]
    print(f"[{name}] drawing weights ...", flush=True)
    sd = synth.synth_state_dict(cfg, seed=base["seed_w"], streams=True)
    wsum, wprobe, wkeys = F.weight_checksum(sd)
    if cfg.has_text:
        ins = synth.synth_inputs(cfg, latent_hw=base["latent_hw"], text_len=base["text_len"], uncond_len=base["uncond_len"], seed=base["seed_x"],
                                 t_value=tv)
        z0 = ins[0].to(torch.bfloat16).float()
        cap, mask = ins[2].to(torch.bfloat16).float(), ins[3]
        model_kw = dict(cap_feats=cap, cap_mask=mask, **ckw)
    else:
        ins = synth.synth_inputs(cfg, latent_hw=base["latent_hw"], seed=base["seed_x"], t_value=tv)
        z0 = ins[0].to(torch.bfloat16).float()
        model_kw = dict(y=ins[2], **ckw)
    # sample.py:206-207: ONE noise draw repeated for the cond and uncond halves
    z0 = z0[:1].repeat(2, 1, 1, 1)

    # ---- reference leg: unmodified transport + unmodified model; grid from the reference's own ode.__init__ ------------------
    os.environ["TORCHDYNAMO_DISABLE"] = "1"
    mod = F._fresh_import(base["pkg"], base["module"])
    driver = case.get("driver", "sampler")
    if driver == "mini":  # the mini fork: one flat transport.py, loaded by path (needs only the torchdiffeq stub of oracle/stubs)
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_mini_transport", os.path.join(F.R.REFERENCE_ROOT, "lumina_next_t2i_mini", "transport.py"))
        tmod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tmod)
        grid = tmod.ODE(num_steps, method, shift).t.clone()
    else:
        tmod = importlib.import_module("transport")
        tr = tmod.create_transport("Linear", "velocity", None, None, None)
        integ = importlib.import_module("transport.integrators")
        okw = dict(drift=None, t0=0, t1=1, sampler_type=method, num_steps=num_steps, atol=1e-6, rtol=1e-3)
        if shift is not None:
            okw["time_shifting_factor"] = shift
        t0s, t1s = tr.check_interval(tr.train_eps, tr.sample_eps, sde=False, eval=True, reverse=False, last_step_size=0.0)
        okw["t0"], okw["t1"] = t0s, t1s
        grid = integ.ode(**okw).t.clone()
    assert len(grid) == num_steps
    done = all(os.path.exists(_ckpt_path(name, "ref", k)) for k in range(1, num_steps))
    if done:
        ref = torch.stack([z0] + [torch.from_numpy(np.load(_ckpt_path(name, "ref", k))) for k in range(1, num_steps)])
        print(f"[{name}] reference trajectory restored from checkpoints", flush=True)
    else:
        cls = getattr(mod, base["cls"])
        kw = cfg.ctor_kwargs()
        if "use_flash_attn" in cls.__init__.__code__.co_varnames:
            kw["use_flash_attn"] = False
        model = cls(**kw).eval()
        res = model.load_state_dict(sd, strict=True, assign=True)
        assert not res.missing_keys and not res.unexpected_keys
        if driver == "mini":
            sample_fn = tmod.ODE(num_steps, method, shift).sample  # lumina_next_t2i_mini/sample.py:204
        else:
            skw = dict(sampling_method=method, num_steps=num_steps)
            if shift is not None:
                skw["time_shifting_factor"] = shift
            sample_fn = tmod.Sampler(tr).sample_ode(**skw)
        t_start = time.time()
        with torch.no_grad():
            ref = sample_fn(z0, _Stepper(model.forward_with_cfg, name, "ref"), **model_kw)  # the reference's own call, sample.py:233
        print(f"[{name}] reference trajectory: {time.time() - t_start:.0f} s", flush=True)
        assert ref.shape == (num_steps,) + tuple(z0.shape) and torch.isfinite(ref).all()
        os.makedirs(CKPT, exist_ok=True)
        for k in range(1, num_steps):
            np.save(_ckpt_path(name, "ref", k), ref[k].float().numpy())
        del model
        gc.collect()

    # ---- floor leg: bf16 choreography, bf16 state, t cast to the state dtype -------------------------------------------------------
    def oracle_fn(bf16):
        def call(t, y):
            tvec = torch.ones(y.size(0)) * t.float()
            if cfg.has_text:
                o = F.oracle_call(cfg, sd, (y.float(), tvec, cap, mask), ckw, bf16)
            else:
                o = F.oracle_call(cfg, sd, (y.float(), tvec, ins[2]), ckw, bf16)
            return o.to(y.dtype)
        return call

    t_start = time.time()
    with torch.no_grad():
        floor = _stepwise(name, "floor", grid, method, z0.to(torch.bfloat16), _progress(oracle_fn(True), name, "floor"), True)
    print(f"[{name}] bf16-choreography trajectory: {time.time() - t_start:.0f} s", flush=True)

    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    drift_floor = np.array([rel(floor[k], ref[k]) for k in range(num_steps)], dtype=np.float64)
    drift_floor_row0 = np.array([rel(floor[k, 0], ref[k, 0]) for k in range(num_steps)], dtype=np.float64)
    pts = case["points"]
    out = {"config": np.array(json.dumps(cfg.to_dict())), "base": np.array(case["base"]), "seed_w": base["seed_w"], "seed_x": base["seed_x"],
           "latent_hw": np.array(base["latent_hw"]), "text_len": base["text_len"], "uncond_len": base["uncond_len"],
           "package": np.array(base["pkg"]), "wsum": wsum, "wprobe": wprobe, "wkeys": np.array(json.dumps(wkeys)),
           "method": np.array(method), "driver": np.array(driver), "num_steps": num_steps, "shift": np.float64(shift if shift is not None else 0.0),
           "model_kw": np.array(json.dumps(ckw)), "grid": grid.numpy().astype(np.float32), "points": np.array(pts),
           "z0": z0[:1].numpy(),
           "ref_points": ref[pts][:, 0].float().numpy(), "floor_points": floor[pts][:, 0].float().numpy(),
           "ref_final": ref[-1].float().numpy(), "floor_final": floor[-1].float().numpy(),
           "drift_floor": drift_floor, "drift_floor_row0": drift_floor_row0,
           "ref_norm": np.array([float(ref[k].norm()) for k in range(num_steps)]),
           "pinned_by": np.array("unmodified reference transport.Sampler.sample_ode + model.forward_with_cfg (fp32 CPU) over the restated "
                                 "fixed-grid odeint (torchdiffeq absent); floor = bf16-choreography restatement, bf16 state")}
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"[{name}] written; floor drift per grid point (all rows): " + " ".join(f"{v:.3e}" for v in drift_floor), flush=True)
    del sd
    gc.collect()


def add_forced_routing(name):
    """`--forced-routing case` (MoE trajectories): the free-running MoE trajectory gate is 1.5 x a floor that ends 0.35 from the
    reference - it bounds very little, because one rounded router logit replaces a token's expert outright and the two bf16 paths are two
    realisations.  Added to the EXISTING fixture, nothing else recomputed:
      route_steps        the experts the fp32 run selects at every NFE of the trajectory ([NFE][L][time, space][rows][2]; the restatement's fp32
                         trajectory is re-run for it and must land on the stored reference states)
      floor_forced_*     the bf16 choreography over the same grid with the selection of NFE k held equal to route_steps[k] (softmax weights,
                         experts, combine stay the run's own): what remains is continuous arithmetic, gated at the dense models' size"""
    from oracle import variants_oracle as V
    case = CASES[name]
    base = F.CASES[case["base"]]
    cfg = base["cfg"]
    assert cfg.family.startswith("moe") and case["method"] == "euler"
    path = os.path.join(OUT, f"{name}.npz")
    old = dict(np.load(path, allow_pickle=False))
    _, tv, ckw = base["calls"][0]
    ckw = dict(ckw)
    sd = synth.synth_state_dict(cfg, seed=base["seed_w"], streams=True)
    wsum, wprobe, _ = F.weight_checksum(sd)
    assert np.array_equal(wsum, old["wsum"]) and np.array_equal(wprobe, old["wprobe"])
    ins = synth.synth_inputs(cfg, latent_hw=base["latent_hw"], seed=base["seed_x"], t_value=tv)
    z0 = ins[0].to(torch.bfloat16).float()[:1].repeat(2, 1, 1, 1)
    assert np.array_equal(z0[:1].numpy(), old["z0"])
    grid = torch.from_numpy(old["grid"])
    tables = []

    def fp32_call(t, y):
        rec = V.MoeRouting(cfg.n_layers)
        o = F.oracle_call(cfg, sd, (y.float(), torch.ones(y.size(0)) * t.float(), ins[2]), ckw, False, rec)
        tables.append(rec.table().astype(np.int8))
        return o.to(y.dtype)

    with torch.no_grad():
        ref = OD.odeint(_progress(fp32_call, name, "fp32 + routing"), z0, grid, method="euler", t_cast=False)
    pts = [int(p) for p in old["points"]]
    err = float((ref[pts][:, 0] - torch.from_numpy(old["ref_points"])).abs().max())
    print(f"[{name}] re-run fp32 trajectory vs the stored reference states: max abs {err:.2e}", flush=True)
    assert err < 1e-4, err  # the restatement equals the reference module bit for bit per NFE; the sampler's own float ops may differ in the last ulp
    route = np.stack(tables)
    step = {"k": 0}

    def forced_call(t, y):
        hook = V.MoeRouting(cfg.n_layers, force=route[step["k"]].astype(np.int64))
        step["k"] += 1
        o = F.oracle_call(cfg, sd, (y.float(), torch.ones(y.size(0)) * t.float(), ins[2]), ckw, True, hook)
        return o.to(y.dtype)

    with torch.no_grad():
        forced = OD.odeint(_progress(forced_call, name, "bf16 forced"), z0.to(torch.bfloat16), grid, method="euler", t_cast=True)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    old["route_steps"] = route
    old["floor_forced_points"] = forced[pts][:, 0].float().numpy()
    old["floor_forced_final"] = forced[-1].float().numpy()
    old["drift_floor_forced"] = np.array([rel(forced[k], ref[k]) for k in range(len(grid))], dtype=np.float64)
    old["drift_floor_forced_row0"] = np.array([rel(forced[k, 0], ref[k, 0]) for k in range(len(grid))], dtype=np.float64)
    np.savez_compressed(path, **old)
    print(f"[{name}] forced-routing floor drift per grid point (row 0): " + " ".join(f"{v:.3e}" for v in old["drift_floor_forced_row0"]), flush=True)


def _progress(call, name, leg):
    state = {"n": 0, "t0": time.time()}

    def wrapped(t, y):
        out = call(t, y)
        state["n"] += 1
        print(f"  [{name}/{leg}] NFE {state['n']}: t = {float(t):.5f}, {time.time() - state['t0']:.0f} s elapsed", flush=True)
        return out
    return wrapped


def main():
    torch.set_grad_enabled(False)
    if os.environ.get("LT_TRAJ_THREADS"):
        torch.set_num_threads(int(os.environ["LT_TRAJ_THREADS"]))
    args = sys.argv[1:]
    if args and args[0] == "--forced-routing":
        for n in args[1:]:
            add_forced_routing(n)
        return
    for n in args or list(CASES):
        run_case(n)


if __name__ == "__main__":
    main()
