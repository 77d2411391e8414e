"""Image-to-image sampling driver on the MI355X engine - command-line compatible with the reference's
``lumina_next_t2i_mini/sample_img2img.py`` (arguments :270-330, flow :146-260):

    input image --[resize, [-1, 1], VAE encoder (:176-181)]--> x1 = latent * vae_scale
    ode = ODE(num_sampling_steps, solver, time_shifting_factor, strength=...)          (the time grid starts at t[int(n (1 - strength))])
    z = noise * (1 - ode.t[0]) + x1 * ode.t[0]  (:189-192), repeated for cond + uncond
    ode.sample(z, model.forward_with_cfg, ...)[-1] --[VAE decoder]--> png + data.json

The step arithmetic is the flat ``ODE`` of the mini package (``transport/mini.py``, grids pinned to the reference class in
tests/golden/mini_ode.npz): a bound ``forward_with_cfg`` of an engine-backed model runs the cut trajectory as ONE ``lt_sample_ode`` call.
Third-party stages stay third-party, as in the reference: text encoder (``transformers.AutoModel``), VAE (``diffusers.AutoencoderKL``,
both directions here), image decoding (PIL).  They can be injected (tests; callers that already hold them).  One process per GPU under
``python -m torch.distributed.run``; captions shard round-robin, no collective on the data path.

    python -m lumina_t2x_amd.sample_img2img --ckpt /ckpts/Lumina-Next-SFT --image in.png --strength 0.6 --solver midpoint \\
        --num_sampling_steps 30 --time_shifting_factor 4 --caption_path prompts.txt --resolution 1024:1024x1024 \\
        --text_encoder /ckpts/gemma-2b --vae /ckpts/sdxl-vae
"""
from __future__ import annotations

import argparse
import json
import math
import os
from typing import Callable, List, Optional

import torch

from . import models
from .sample import VAE_SCALE, load_checkpoint, load_train_args, make_text_encoder, parse_resolution, save_png
from .transport.mini import ODE


def load_image(path: str, w: int, h: int, device) -> torch.Tensor:
    """PIL image -> RGB, resized to (w, h), [3, h, w] in [-1, 1] (reference :165-174: ToTensor + Normalize(0.5, 0.5))"""
    import numpy as np
    from PIL import Image
    img = Image.open(path).resize((w, h)).convert("RGB")
    arr = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div_(255.0)
    return arr.sub_(0.5).div_(0.5).to(device)


def make_vae(path: Optional[str], device):
    """(encode(image [1,3,h,w] in [-1,1]) -> latent sample, decode(latent) -> image in [0,1]) or (None, None)"""
    if not path:
        return None, None
    try:
        from diffusers.models import AutoencoderKL
    except ImportError:
        return None, None
    vae = AutoencoderKL.from_pretrained(path, torch_dtype=torch.float32).to(device).eval()

    @torch.no_grad()
    def encode(img):
        return vae.encode(img.float()).latent_dist.sample()

    @torch.no_grad()
    def decode(lat):
        return ((vae.decode(lat.float()).sample + 1.0) / 2.0).clamp_(0.0, 1.0)

    return encode, decode


def run(args, *, encode_fn=None, cap_feat_dim=None, vae_encode_fn: Optional[Callable] = None, decode_fn=None, model=None,
        image: Optional[torch.Tensor] = None) -> List[dict]:
    torch.set_grad_enabled(False)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[args.precision]
    train_args = load_train_args(args.ckpt)
    if encode_fn is None:
        encode_fn, cap_feat_dim = make_text_encoder(args.text_encoder, dtype, device)
    if model is None:
        model = models.__dict__[train_args.model](qk_norm=train_args.qk_norm, cap_feat_dim=cap_feat_dim)
        model.eval().to(device, dtype=dtype)
        if not args.debug:
            model.load_state_dict(load_checkpoint(args.ckpt, args.ema), strict=True)
    if vae_encode_fn is None or decode_fn is None:
        enc, dec = make_vae(args.vae, device)
        vae_encode_fn = vae_encode_fn or enc
        decode_fn = decode_fn or dec
    if vae_encode_fn is None:
        raise RuntimeError("img2img needs a VAE ENCODER for the input image: pass --vae <local diffusers AutoencoderKL> (diffusers must be "
                           "installed) or inject vae_encode_fn; there is no way to start from an image without it")
    out_dir = args.image_save_path
    os.makedirs(os.path.join(out_dir, "images"), exist_ok=True)
    with open(args.caption_path, "r", encoding="utf-8") as f:
        captions = [ln.strip() for ln in f if ln.strip()]
    factor = VAE_SCALE.get(getattr(train_args, "vae", "sdxl"), 0.18215)
    ode = ODE(args.num_sampling_steps, args.solver, args.time_shifting_factor, strength=args.strength)
    if len(ode.t) < 2:
        raise ValueError(f"strength {args.strength} with {args.num_sampling_steps} steps leaves no interval to integrate")
    info: List[dict] = []
    jobs = [(res, i, c) for res in args.resolution for i, c in enumerate(captions)]
    for j, (res, idx, caption) in enumerate(jobs):
        if j % world != rank:
            continue
        cat, w, h = parse_resolution(res)
        if int(args.seed) != 0:
            torch.random.manual_seed(int(args.seed))
        img = image if image is not None else load_image(args.image, w, h, device)
        x1 = vae_encode_fn(img[None] if img.dim() == 3 else img).mul(factor)          # :179-181
        # :187 draws [1, 4, w // 8, h // 8] (width first) while the VAE latent is [1, 4, h // 8, w // 8]: identical for squares, a
        # broadcast error at the mix below for every other resolution.  Deviation: the noise takes the latent's own shape -
        # the same values for squares, and non-square resolutions work instead of dying.
        z = torch.randn([1, 4, h // 8, w // 8], device=device).to(dtype)
        if tuple(x1.shape[-2:]) != (h // 8, w // 8):
            raise ValueError(f"VAE latent {tuple(x1.shape)} does not match the requested resolution {w}x{h} (expected [..., {h // 8}, {w // 8}])")
        t0 = float(ode.t[0])
        z = z * (1 - t0) + x1.to(dtype) * t0                                          # :190
        z = z.repeat(2, 1, 1, 1)
        cap_feats, cap_mask = encode_fn([caption, ""])
        kw = dict(cap_feats=cap_feats, cap_mask=cap_mask.to(cap_feats.device), cfg_scale=args.cfg_scale,
                  proportional_attn=bool(args.proportional_attn),
                  base_seqlen=(train_args.image_size // 16) ** 2 if args.proportional_attn else None)
        if cat > 1024 and args.scaling_method == "Time-aware":
            kw.update(scale_factor=math.sqrt(w * h / train_args.image_size ** 2), scale_watershed=args.scaling_watershed)
        else:
            kw.update(scale_factor=1.0, scale_watershed=1.0)
        latent = ode.sample(z, model.forward_with_cfg, **kw)[-1][:1]
        stem = os.path.join(out_dir, "images", f"{args.solver}_{args.num_sampling_steps}_{idx}_{res.split(':')[-1]}")
        if decode_fn is not None:
            save_png(decode_fn(latent / factor)[0], stem + ".png")
            url = stem + ".png"
        else:
            torch.save(latent.cpu(), stem + ".pt")
            url = stem + ".pt"
        info.append({"caption": caption, "image_url": url, "resolution": f"res: {res.split(':')[-1]}\ntime_shift: {args.time_shifting_factor}",
                     "solver": args.solver, "num_sampling_steps": args.num_sampling_steps})
    with open(os.path.join(out_dir, f"data.rank{rank}.json" if world > 1 else "data.json"), "w") as f:
        json.dump(info, f)
    return info


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--cfg_scale", type=float, default=4.0)
    p.add_argument("--num_sampling_steps", type=int, default=250)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--solver", type=str, default="euler")
    p.add_argument("--strength", type=float, default=0.6)
    p.add_argument("--image", type=str, required=True)
    p.add_argument("--precision", type=str, choices=["fp32", "bf16"], default="bf16")
    p.add_argument("--ema", action="store_true", default=True)
    p.add_argument("--no-ema", dest="ema", action="store_false")
    p.add_argument("--image_save_path", type=str, default="samples")
    p.add_argument("--time_shifting_factor", type=float, default=1.0)
    p.add_argument("--caption_path", type=str, default="prompts.txt")
    p.add_argument("--resolution", type=str, default=[], nargs="+")
    p.add_argument("--proportional_attn", type=lambda v: str(v).lower() not in ("0", "false", "no"), default=True)
    p.add_argument("--scaling_method", type=str, default="Time-aware")
    p.add_argument("--scaling_watershed", type=float, default=0.3)
    p.add_argument("--debug", action="store_true", help="random-init weights (no checkpoint load), as in the reference")
    p.add_argument("--text_encoder", type=str, default="google/gemma-2b", help="local path of the text encoder (no network)")
    p.add_argument("--vae", type=str, default="", help="local path of the diffusers AutoencoderKL weights (encoder AND decoder)")
    # accepted for command-line compatibility with lumina_next_t2i_mini/sample_img2img.py:262-300 and ignored: one process drives one GPU
    # here (no model-parallel group), the tokenizer comes with --text_encoder, there is one attention implementation, captions run one by one
    p.add_argument("--num_gpus", type=int, default=1, help="ignored (reference: model-parallel world size, asserted == 1)")
    p.add_argument("--tokenizer_path", type=str, default="", help="ignored (the tokenizer is loaded from --text_encoder)")
    p.add_argument("--use_flash_attn", type=lambda v: str(v).lower() not in ("0", "false", "no"), default=True, help="ignored")
    p.add_argument("--batch_size", type=int, default=1, help="ignored (captions are sampled one at a time, as in the reference loop)")
    return p


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
