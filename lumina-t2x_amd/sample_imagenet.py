"""Class-conditional sampling driver on the MI355X engine - command-line compatible with the reference's
``Next-DiT-ImageNet/sample.py`` (``python sample.py ODE|SDE --ckpt DIR --class_labels 207 360 ...``; argument names and defaults
:215-262, flow :80-203):

    z ~ N(0, I) [n, 4, size/4, size/4] (sic, :107: the reference samples at twice the training latent), doubled for CFG with the
    null class 1000 (:176-181) --[Sampler.sample_ode | sample_sde on the engine]--> latents --[sd-vae-ft decoder]--> one png grid

What is ours: the DiT runs on the HIP engine behind ``models.imagenet``; one process, one GPU (the reference asserts
``num_gpus == 1`` too, :267).  The VAE stays third-party (``diffusers.AutoencoderKL`` from ``--local_diffusers_model_root``;
there is no network here) - without it the final latents are written as ``.pt`` next to where the png would go.
``--likelihood`` (sample.py:134-141; needs ``--cfg_scale 1``) integrates the likelihood ODE with the divergence taken by central
differences through the forward-only engine (``Sampler.sample_ode_likelihood(divergence="fd")``) where the reference uses autograd;
like the reference, the run then takes ``[-1]`` of the returned (logp, z) pair = the latent carried to the noise end.

    python -m lumina_t2x_amd.sample_imagenet ODE --ckpt /ckpts/next-dit-imagenet --sampling-method euler --num_sampling_steps 50
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Callable, Optional

import torch

from .models import imagenet as models
from .sample import save_png
from .transport import Sampler, create_transport

NULL_CLASS = 1000  # reference sample.py:179


def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2) -> torch.Tensor:
    """[n, 3, H, W] in [0, 1] -> one [3, *, *] image, ``nrow`` per row, black ``padding`` around every tile (the layout of
    torchvision's ``save_image(..., nrow=8)``, which the reference calls at :195-201)."""
    n, c, h, w = images.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = torch.zeros(c, rows * (h + padding) + padding, cols * (w + padding) + padding, dtype=images.dtype, device=images.device)
    for i in range(n):
        r, q = divmod(i, cols)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = images[i]
    return grid


def make_vae_decoder(root: Optional[str], vae_name: str, device) -> Optional[Callable[[torch.Tensor], torch.Tensor]]:
    """latents / 0.18215 -> images in [-1, 1]; None when diffusers or the local weights are unavailable."""
    if not root:
        return None
    try:
        from diffusers.models import AutoencoderKL
    except ImportError:
        return None
    vae = AutoencoderKL.from_pretrained(os.path.join(root, f"stabilityai/sd-vae-ft-{vae_name}")).to(device).eval()

    @torch.no_grad()
    def decode(lat):
        return vae.decode(lat.float()).sample

    return decode


def build_sample_fn(args, mode: str):
    sampler = Sampler(create_transport(args.path_type, args.prediction, args.loss_weight, args.train_eps, args.sample_eps))
    if mode == "ODE":
        if getattr(args, "likelihood", False):  # sample.py:134-141 (the forward-only engine takes the finite-difference divergence)
            assert args.cfg_scale == 1, "Likelihood is incompatible with guidance"
            return sampler.sample_ode_likelihood(sampling_method=args.sampling_method, num_steps=args.num_sampling_steps,
                                                 atol=args.atol, rtol=args.rtol)
        return sampler.sample_ode(sampling_method=args.sampling_method, num_steps=args.num_sampling_steps, atol=args.atol,
                                  rtol=args.rtol, reverse=args.reverse)
    return sampler.sample_sde(sampling_method=args.sampling_method, diffusion_form=args.diffusion_form,
                              diffusion_norm=args.diffusion_norm, last_step=args.last_step, last_step_size=args.last_step_size,
                              num_steps=args.num_sampling_steps)


def run(args, mode: str, *, decode_fn=None, model=None) -> str:
    """Sample one image per class label and write the grid; returns the path written."""
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    train_args = torch.load(os.path.join(args.ckpt, "model_args.pth"), map_location="cpu", weights_only=False)
    print("Model arguments used for inference:", json.dumps({k: v for k, v in vars(train_args).items()
                                                             if isinstance(v, (int, float, str, bool, type(None)))}, indent=2))
    latent_size = train_args.image_size // 4
    dtype = {"fp32": torch.float, "tf32": torch.float, "bf16": torch.bfloat16}.get(args.precision)
    if dtype is None:
        raise ValueError(f"--precision {args.precision}: the engine computes in bf16 with bf16 or fp32 state (fp32 | tf32 | bf16)")
    if model is None:
        model = models.__dict__[train_args.model](input_size=latent_size, num_classes=train_args.num_classes, qk_norm=train_args.qk_norm)
        model.to(dtype).to(device)
        ckpt = torch.load(os.path.join(args.ckpt, f"consolidated{'_ema' if args.ema else ''}.00-of-01.pth"), map_location="cpu",
                          weights_only=True)
        model.load_state_dict(ckpt, strict=True)
        model.eval()
    sample_fn = build_sample_fn(args, mode)
    if decode_fn is None:
        decode_fn = make_vae_decoder(args.local_diffusers_model_root, getattr(train_args, "vae", "ema"), device)
    n = len(args.class_labels)
    z = torch.randn(n, 4, latent_size, latent_size, dtype=dtype, device=device)
    y = torch.tensor(args.class_labels, device=device)
    z = torch.cat([z, z], 0)
    y = torch.cat([y, torch.tensor([NULL_CLASS] * n, device=device)], 0)
    samples = sample_fn(z, model.forward_with_cfg, y=y, cfg_scale=args.cfg_scale)[-1]
    samples, _ = samples.chunk(2, dim=0)  # drop the null-class half
    path = args.image_save_path or os.path.join(args.ckpt, f"sample{'_ema' if args.ema else ''}.png")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if decode_fn is None:
        path = os.path.splitext(path)[0] + ".pt"
        print(f"[sample_imagenet] no VAE decoder (diffusers or --local_diffusers_model_root missing): writing latents to {path}", flush=True)
        torch.save(samples.cpu(), path)
        return path
    images = decode_fn(samples / 0.18215)
    save_png(make_grid(((images.float() + 1.0) / 2.0).clamp(0.0, 1.0), nrow=8), path)  # normalize=True, value_range=(-1, 1)
    return path


def build_parser(mode: str) -> argparse.ArgumentParser:
    def none_or_str(v):
        return None if v == "None" else v

    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--cfg_scale", type=float, default=4.0)
    p.add_argument("--num_sampling_steps", type=int, default=250)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--class_labels", type=int, nargs="+", default=[207, 360, 387, 974, 88, 979, 417, 279],
                   help="Class labels to generate the images for.")
    p.add_argument("--precision", type=str, choices=["fp32", "tf32", "fp16", "bf16"], default="tf32")
    p.add_argument("--local_diffusers_model_root", type=str, help="root directory of the local diffusers models (no network here)")
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--ema", action="store_true", help="Use EMA models.")
    p.add_argument("--no_ema", action="store_false", dest="ema", help="Do not use EMA models.")
    p.set_defaults(ema=True)
    p.add_argument("--image_save_path", type=str, help="overrides the default sample{_ema}.png in the checkpoint directory")
    g = p.add_argument_group("Transport arguments")
    g.add_argument("--path-type", type=str, default="Linear", choices=["Linear", "GVP", "VP"])
    g.add_argument("--prediction", type=str, default="velocity", choices=["velocity", "score", "noise"])
    g.add_argument("--loss-weight", type=none_or_str, default=None, choices=[None, "velocity", "likelihood"])
    g.add_argument("--sample-eps", type=float)
    g.add_argument("--train-eps", type=float)
    if mode == "ODE":
        g = p.add_argument_group("ODE arguments")
        g.add_argument("--sampling-method", type=str, default="dopri5")
        g.add_argument("--atol", type=float, default=1e-6, help="Absolute tolerance")
        g.add_argument("--rtol", type=float, default=1e-3, help="Relative tolerance")
        g.add_argument("--reverse", action="store_true")
        g.add_argument("--likelihood", action="store_true")
    else:
        g = p.add_argument_group("SDE arguments")
        g.add_argument("--sampling-method", type=str, default="Euler", choices=["Euler", "Heun"])
        g.add_argument("--diffusion-form", type=str, default="sigma",
                       choices=["constant", "SBDM", "sigma", "linear", "decreasing", "increasing-decreasing"])
        g.add_argument("--diffusion-norm", type=float, default=1.0)
        g.add_argument("--last-step", type=none_or_str, default="Mean", choices=[None, "Mean", "Tweedie", "Euler"])
        g.add_argument("--last-step-size", type=float, default=0.04)
    return p


def parse(argv):
    """The reference reads the mode from argv[1] and falls back to ODE when it is anything else (:218-220), then parses the rest
    with parse_known_args (:262)."""
    argv = list(argv)
    mode = argv[0] if argv and argv[0] in ("ODE", "SDE") else "ODE"
    args = build_parser(mode).parse_known_args(argv[1:] if argv and argv[0] in ("ODE", "SDE") else argv)[0]
    if args.num_gpus != 1:
        raise AssertionError("Multi-GPU sampling is currently not supported.")
    return mode, args


def main(argv=None):
    mode, args = parse(sys.argv[1:] if argv is None else argv)
    return run(args, mode)


if __name__ == "__main__":
    main()
