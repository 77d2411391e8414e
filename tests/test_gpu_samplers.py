"""-m gpu: the remaining transport/ samplers driven through the ENGINE's model callable: adaptive dopri5
(the default solver of Next-DiT-ImageNet/sample.py:48 and of Sampler.sample_ode, transport.py:349) and the SDE samplers
(transport.py:285-344, integrators.py:27-76: Euler-Maruyama / Heun with a last-step rule), each against the same sampler code
driven by the CPU oracle (fp32) on identical inputs and - for the SDE - identical noise (the samplers draw it from torch's CPU
generator, integrators.py:37 / :48, so one seed reproduces it on both sides).

Tolerances: the engine computes in bf16 where the reference's autocast path does; per model evaluation that is <= 2.5e-2 / 6e-2
(plain / cfg 4) relative to fp32.  dopri5's step controller sees that noise in its error estimate, so the test
runs it at rtol = atol = 2e-2 (at the reference's defaults 1e-3 / 1e-6 ANY bf16 model - the reference's own included - makes
the controller chase rounding noise) and compares the states at the requested grid points; accepted step counts may differ."""
import json
import os

import numpy as np
import pytest
import torch

import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import models
from lumina_t2x_amd.transport import Sampler, create_transport
from oracle import nextdit_oracle as O
from oracle import synth
from oracle import variants_oracle as V

from gpu_util import rel_l2

pytestmark = pytest.mark.gpu


def _imagenet(golden_dir):
    g = np.load(os.path.join(golden_dir, "imagenet_tiny.npz"), allow_pickle=False)
    cfg = synth.NextDiTConfig(**json.loads(str(g["config"])))
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    m = models.imagenet.DiT_Llama(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=True)
    return g, cfg, sd, m.eval().to("cuda", torch.bfloat16)


def _next(golden_dir):
    g = np.load(os.path.join(golden_dir, "nextdit_tiny.npz"), allow_pickle=False)
    cfg = synth.NextDiTConfig(**json.loads(str(g["config"])))
    sd = synth.synth_state_dict(cfg, seed=int(g["seed_w"]))
    m = models.NextDiT(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=True)
    return g, cfg, sd, m.eval().to("cuda", torch.bfloat16)


@pytest.mark.parametrize("state_dtype", [torch.float32, torch.bfloat16])
def test_dopri5_on_the_engine_vs_oracle_driven_solver(golden_dir, state_dtype):
    g, cfg, sd, model = _imagenet(golden_dir)
    z = torch.from_numpy(g["z"])
    y = torch.from_numpy(g["y"])
    tr = create_transport("Linear", "velocity", None, None, None)
    kw = dict(sampling_method="dopri5", num_steps=4, atol=2e-2, rtol=2e-2)
    calls = []

    def engine_fn(x, t, **k):
        calls.append(float(t[0]))
        return model.forward_with_cfg(x, t, **k)

    got = Sampler(tr).sample_ode(**kw)(z.to("cuda", state_dtype), engine_fn, y=y.cuda(), cfg_scale=4.0)
    # the oracle-driven solve keeps its state in the same dtype (a bf16 state re-rounds after every accepted step, which moves
    # the controller's step sequence - compare like with like)
    ref = Sampler(tr).sample_ode(**kw)(z.to(state_dtype), lambda x, t, **k: V.imagenet_forward_with_cfg(sd, cfg, x.float(), t, **k).to(state_dtype),
                                       y=y, cfg_scale=4.0)
    assert got.shape == ref.shape == (4,) + tuple(z.shape) and got.dtype == state_dtype
    # 2 evaluations for the initial step + 6 per attempted step; like torchdiffeq, steps may overshoot t1 (dense output interpolates back)
    assert len(calls) >= 8 and min(calls) >= 0.0
    assert torch.equal(got[0].float().cpu(), z.to(state_dtype).float())
    tol = 6e-2 if state_dtype == torch.float32 else 1e-1  # bf16 state: one ulp (4e-3) per accepted step on top of the model noise
    for i in (1, 2, 3):
        assert rel_l2(got[i], ref[i]) < tol, (i, rel_l2(got[i], ref[i]))


def test_likelihood_ode_on_the_engine_fd_divergence_vs_oracle_autograd(golden_dir, monkeypatch):
    """Sampler.sample_ode_likelihood (transport.py:393-450) with the ENGINE as the model: the forward-only engine takes the
    divergence by central differences (`divergence="auto"` -> "fd" for an engine-backed bound method), the fp32 oracle on the CPU
    takes it the reference's way (autograd).  Both sides see the same Hutchinson probes (drawn from torch's CPU generator).
    `forward` (no guidance: sample.py:135 asserts cfg_scale == 1) keeps the samples independent - through `forward_with_cfg`
    the second half of the batch is computed from the first (:model.py forward_with_cfg), and the reference's per-sample split of
    eps^T J^T eps differs from eps^T J eps by a zero-mean cross term; only their batch sums agree (checked on the CPU oracle).

    The gate: z within the per-evaluation bf16 tolerance; logp within a few percent of |delta logp| - the finite difference
    divides the engine's bf16 rounding noise by 2 h (documented in sample_ode_likelihood)."""
    g, cfg, sd, model = _imagenet(golden_dir)
    z, y = torch.from_numpy(g["z"]), torch.from_numpy(g["y"])
    real = torch.randint

    def cpu_randint(*a, device=None, **k):
        out = real(*a, **k)
        return out.to(device) if device is not None else out

    monkeypatch.setattr(torch, "randint", cpu_randint)
    tr = create_transport("Linear", "velocity", None, None, None)
    kw = dict(sampling_method="euler", num_steps=5)
    torch.manual_seed(5)
    logp, zz = Sampler(tr).sample_ode_likelihood(**kw)(z.cuda(), model.forward, y=y.cuda())
    torch.manual_seed(5)
    rlogp, rz = Sampler(tr).sample_ode_likelihood(divergence="autograd", **kw)(z.clone(), lambda x, t, **k: V.imagenet_forward(sd, cfg, x, t, **k), y=y)
    torch.manual_seed(5)
    flogp, fz = Sampler(tr).sample_ode_likelihood(divergence="fd", **kw)(z.clone(), lambda x, t, **k: V.imagenet_forward(sd, cfg, x, t, **k), y=y)
    def prior(v):
        return -z[0].numel() / 2.0 * np.log(2 * np.pi) - v.float().cpu().flatten(1).pow(2).sum(1) / 2.0

    # logp = log N(z) - delta: the divergence integral (what this sampler adds) and the prior of the carried state, separately
    delta, rdelta, fdelta = prior(zz) - logp.cpu(), prior(rz) - rlogp, prior(fz) - flogp
    print("delta engine-fd", delta.tolist(), "oracle autograd", rdelta.tolist(), "oracle fd", fdelta.tolist(),
          "| prior engine", prior(zz).tolist(), "oracle", prior(rz).tolist())
    assert logp.shape == (2,) and zz.shape == z.shape and torch.isfinite(logp).all()
    assert rel_l2(zz, rz) < 3e-2 and rel_l2(fz, rz) < 1e-5
    assert ((fdelta - rdelta).abs() < 1e-2 * rdelta.abs() + 0.1).all()        # fp32 model: fd == autograd to O(h^2) (0.7 % here)
    assert ((delta - rdelta).abs() < 0.1 * rdelta.abs() + 1.0).all(), (delta, rdelta)
    # the prior term moves with |z|^2 / 2 (~900 here): 3e-2 on z is up to ~5e-2 on it, with any bf16 model
    assert ((prior(zz) - prior(rz)).abs() < 5e-2 * rz.flatten(1).pow(2).sum(1) / 2.0).all()


@pytest.mark.parametrize("method,last_step", [("Euler", "Mean"), ("Heun", "Tweedie"), ("Euler", "Euler")])
def test_sample_sde_on_the_engine_vs_oracle_driven_sampler(golden_dir, method, last_step):
    g, cfg, sd, model = _next(golden_dir)
    z = torch.from_numpy(g["z"])
    cap, mask = torch.from_numpy(g["cap"]), torch.from_numpy(g["mask"])
    tr = create_transport("Linear", "velocity", None, None, None)
    kw = dict(sampling_method=method, diffusion_form="sigma", diffusion_norm=0.5, last_step=last_step, last_step_size=0.04, num_steps=6)
    mk = dict(cfg_scale=4.0, proportional_attn=True, base_seqlen=16)
    torch.manual_seed(77)
    got = Sampler(tr).sample_sde(**kw)(z.to("cuda", torch.bfloat16), model.forward_with_cfg, cap_feats=cap.to("cuda", torch.bfloat16),
                                       cap_mask=mask.cuda(), **mk)
    torch.manual_seed(77)
    capr = cap.to(torch.bfloat16).float()
    ref = Sampler(tr).sample_sde(**kw)(z.to(torch.bfloat16).float(), lambda x, t, **k: O.forward_with_cfg(sd, cfg, x, t, **k),
                                       cap_feats=capr, cap_mask=mask, **mk)
    assert len(got) == len(ref) == 6
    # the bf16 state re-rounds after every step (like the reference's autocast path); the gate is the trajectory tolerance
    err = rel_l2(got[-1], ref[-1])
    assert torch.isfinite(got[-1].float()).all() and err < 6e-2, err
    assert rel_l2(got[0], ref[0]) < 3e-2


def test_weight_reload_invalidates_the_hoisted_prompt_work(golden_dir):
    """forward, load different weights into the SAME model, forward again with the SAME caption tensors - the
    text K / V of every layer and the caption embedding were hoisted from the old weights and must be rebuilt, i.e. the
    second output must equal a fresh model's, not a mix."""
    g, cfg, sd, model = _next(golden_dir)
    z = torch.from_numpy(g["z"]).to("cuda", torch.bfloat16)
    t = torch.from_numpy(g["t"]).cuda()
    cap, mask = torch.from_numpy(g["cap"]).to("cuda", torch.bfloat16), torch.from_numpy(g["mask"]).cuda()
    first = model.forward_with_cfg(z, t, cap, mask, 4.0)
    sd2 = synth.synth_state_dict(cfg, seed=int(g["seed_w"]) + 1000)
    model.load_state_dict({k: v.to("cuda", torch.bfloat16) for k, v in sd2.items()}, strict=True)
    second = model.forward_with_cfg(z, t, cap, mask, 4.0)  # same cap / mask OBJECTS: the prompt cache would hit
    fresh = models.NextDiT(**cfg.ctor_kwargs())
    fresh.load_state_dict(sd2, strict=True)
    fresh = fresh.eval().to("cuda", torch.bfloat16)
    want = fresh.forward_with_cfg(z, t, cap, mask, 4.0)
    assert torch.equal(second, want)
    assert not torch.equal(second, first)
    with torch.inference_mode():  # inference tensors carry no version counter: the caches must not touch it
        zi, capi = z.clone(), cap.clone()
        third = model.forward_with_cfg(zi, t.clone(), capi, mask.clone(), 4.0)
    assert torch.equal(third, want)
