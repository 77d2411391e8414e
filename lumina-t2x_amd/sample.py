"""Text-to-image sampling driver on the MI355X engine - command-line compatible with the reference's
``lumina_next_t2i/sample.py`` (argument names and defaults :267-330, output layout ``<image_save_path>/images/*.png`` +
``data.json`` :144-260) for the step either side of the hot path:

    caption --[text encoder, hidden_states[-2] (:47-51)]--> cap_feats, cap_mask
    z ~ N(0, I) [1,4,h/8,w/8] repeated for cond + uncond (:201-203) --[Sampler.sample_ode on the engine]--> latent
    latent / vae_scale --[VAE decoder (:237-240)]--> png

What is ours: one process per GPU under ``python -m torch.distributed.run`` (RANK / WORLD_SIZE from the environment, captions
sharded round-robin, no collective on the data path) instead of mp.spawn + fairscale model-parallel groups; the DiT runs on
the HIP engine behind ``models.NextDiT``.  What stays third-party, exactly as in the reference: the text encoder
(``transformers.AutoModel``) and the VAE (``diffusers.AutoencoderKL``).  Both are loaded from LOCAL paths (no network here);
without ``diffusers`` the driver writes the final latents as ``.pt`` files instead of pngs and says so.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m lumina_t2x_amd.sample \\
        --ckpt /ckpts/Lumina-Next-SFT --text_encoder /ckpts/gemma-2b --vae /ckpts/sdxl-vae \\
        --caption_path prompts.txt --resolution 1024:1024x1024 --num_sampling_steps 30 --sampling-method midpoint \\
        --time_shifting_factor 4
"""
from __future__ import annotations

import argparse
import json
import math
import os
from typing import Callable, List, Optional, Tuple

import torch

from . import models
from .transport import Sampler, create_transport

VAE_SCALE = {"sdxl": 0.13025}  # everything else 0.18215 (reference sample.py:236)


def load_train_args(ckpt_dir: str):
    """``model_args.pth`` is the pickled argparse.Namespace of the training run (reference sample.py:98)."""
    return torch.load(os.path.join(ckpt_dir, "model_args.pth"), map_location="cpu", weights_only=False)


def load_checkpoint(ckpt_dir: str, ema: bool) -> dict:
    """consolidated{_ema}.00-of-01.{safetensors,pth} (single model-parallel rank: the engine holds whole weights)."""
    stem = os.path.join(ckpt_dir, f"consolidated{'_ema' if ema else ''}.00-of-01")
    if os.path.exists(stem + ".safetensors"):
        from safetensors.torch import load_file
        return load_file(stem + ".safetensors", device="cpu")
    if os.path.exists(stem + ".pth"):
        return torch.load(stem + ".pth", map_location="cpu", weights_only=True)
    import glob
    shards = sorted(glob.glob(os.path.join(ckpt_dir, f"consolidated{'_ema' if ema else ''}.*-of-*.*")))
    if shards:
        # train.py:625-634 writes one file per model-parallel rank.  They are not a split of ONE model: with qk_norm the reference builds
        # q_norm / k_norm as nn.LayerNorm(n_LOCAL_heads * head_dim) (models/model.py:211-215) - each rank normalises over ITS heads only, so an
        # MP = k checkpoint computes k grouped LayerNorms where the MP = 1 model (every released checkpoint) computes one over the full width.
        # Concatenating the Column / RowParallelLinear shards would load, and sample something else.
        raise FileNotFoundError(f"{ckpt_dir} holds model-parallel shards ({', '.join(os.path.basename(f) for f in shards)}); this engine loads "
                                "model-parallel size 1 only: with qk_norm the reference's q / k LayerNorms span the LOCAL heads of each rank "
                                "(models/model.py:211-215), so an MP > 1 checkpoint is a different function from the concatenation of its shards")
    raise FileNotFoundError(f"{stem}.safetensors / .pth not found")


def make_text_encoder(path: str, dtype, device, add_eos: bool = True) -> Tuple[Callable[[List[str]], Tuple[torch.Tensor, torch.Tensor]], int]:
    """(encode(captions) -> (feats [n, T, C], mask [n, T]), C) with the reference's tokenizer settings (sample.py:34-51;
    ``add_eos=False`` is the ``lumina_next`` CLI's tokenizer, utils/cli.py:121)."""
    from transformers import AutoModel, AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path, add_eos=True) if add_eos else AutoTokenizer.from_pretrained(path)
    tok.padding_side = "right"
    enc = AutoModel.from_pretrained(path, torch_dtype=dtype).to(device).eval()

    @torch.no_grad()
    def encode(captions):
        ti = tok(captions, padding=True, pad_to_multiple_of=8, max_length=256, truncation=True, return_tensors="pt")
        out = enc(input_ids=ti.input_ids.to(device), attention_mask=ti.attention_mask.to(device), output_hidden_states=True)
        return out.hidden_states[-2], ti.attention_mask.to(device)

    return encode, enc.config.hidden_size


def make_vae_decoder(path: Optional[str], device) -> Optional[Callable[[torch.Tensor], torch.Tensor]]:
    """latent (already divided by the scale factor) -> image in [0, 1]; None when diffusers / the weights are unavailable."""
    if not path:
        return None
    try:
        from diffusers.models import AutoencoderKL
    except ImportError:
        return None
    vae = AutoencoderKL.from_pretrained(path, torch_dtype=torch.float32).to(device).eval()

    @torch.no_grad()
    def decode(lat):
        return ((vae.decode(lat.float()).sample + 1.0) / 2.0).clamp_(0.0, 1.0)

    return decode


def parse_resolution(spec: str) -> Tuple[int, int, int]:
    """'1024:1024x1024' -> (category, w, h) (reference sample.py:193-199)."""
    cat, wh = spec.split(":")
    w, h = wh.split("x")
    return int(cat), int(w), int(h)


def save_png(img: torch.Tensor, path: str) -> None:
    """[3, H, W] in [0, 1] -> 8-bit png (PIL if present, else a minimal zlib writer: no torchvision here)."""
    arr = (img.float().clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
    try:
        from PIL import Image
        Image.fromarray(arr).save(path)
        return
    except ImportError:
        pass
    import struct
    import zlib
    h, w, _ = arr.shape
    raw = b"".join(b"\x00" + arr[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def run(args, *, encode_fn=None, cap_feat_dim=None, decode_fn=None, model=None) -> List[dict]:
    """Sample every (resolution, caption) pair of this rank's shard.  ``encode_fn`` / ``decode_fn`` / ``model`` can be injected
    (tests, or a caller that already holds the encoder / VAE); otherwise they are built from the command-line paths."""
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
    if decode_fn is None:
        decode_fn = make_vae_decoder(args.vae, device)
        if decode_fn is None and rank == 0:
            print("[sample] no VAE decoder (diffusers or --vae missing): writing final latents as .pt", flush=True)
    out_dir = args.image_save_path
    os.makedirs(os.path.join(out_dir, "images"), exist_ok=True)
    with open(args.caption_path, "r", encoding="utf-8") as f:
        captions = [ln.strip() for ln in f if ln.strip()]
    vae_name = getattr(train_args, "vae", "sdxl")
    factor = VAE_SCALE.get(vae_name, 0.18215)
    info: List[dict] = []
    jobs = [(res, i, c) for res in args.resolution for i, c in enumerate(captions)]
    for j, (res, idx, caption) in enumerate(jobs):
        if j % world != rank:
            continue
        cat, w, h = parse_resolution(res)
        transport = create_transport(args.path_type, args.prediction, args.loss_weight, args.train_eps, args.sample_eps)
        sample_fn = Sampler(transport).sample_ode(sampling_method=args.sampling_method, num_steps=args.num_sampling_steps,
                                                  atol=args.atol, rtol=args.rtol, reverse=args.reverse,
                                                  time_shifting_factor=args.time_shifting_factor)
        if int(args.seed) != 0:
            torch.manual_seed(int(args.seed))
        # note: the reference draws [1, 4, w/8, h/8] (sample.py:201-202), i.e. width first - kept
        z = torch.randn([1, 4, w // 8, h // 8], device=device).to(dtype).repeat(2, 1, 1, 1)
        cap_feats, cap_mask = encode_fn([caption, ""])
        kw = dict(cap_feats=cap_feats, cap_mask=cap_mask.to(cap_feats.device), cfg_scale=args.cfg_scale,
                  proportional_attn=bool(args.proportional_attn),
                  base_seqlen=(train_args.image_size // 16) ** 2 if args.proportional_attn else None)
        if cat > 1024 and args.scaling_method == "Time-aware":
            kw.update(scale_factor=math.sqrt(w * h / train_args.image_size ** 2), scale_watershed=args.scaling_watershed)
        else:
            kw.update(scale_factor=1.0, scale_watershed=1.0)
        latent = sample_fn(z, model.forward_with_cfg, **kw)[-1][:1]
        stem = os.path.join(out_dir, "images", f"{args.sampling_method}_{args.num_sampling_steps}_{idx}_{res.split(':')[-1]}")
        if decode_fn is not None:
            save_png(decode_fn(latent / factor)[0], stem + ".png")
            url = stem + ".png"
        else:
            torch.save(latent.cpu(), stem + ".pt")
            url = stem + ".pt"
        info.append({"caption": caption, "image_url": url, "resolution": f"res: {res.split(':')[-1]}\ntime_shift: {args.time_shifting_factor}",
                     "sampling_method": args.sampling_method, "num_sampling_steps": args.num_sampling_steps})
    with open(os.path.join(out_dir, f"data.rank{rank}.json" if world > 1 else "data.json"), "w") as f:
        json.dump(info, f)
    return info


def build_parser() -> argparse.ArgumentParser:
    def none_or_str(v):
        return None if v == "None" else v

    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--cfg_scale", type=float, default=4.0)
    p.add_argument("--num_sampling_steps", type=int, default=250)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--ckpt", type=str, required=True)
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
    p.add_argument("--vae", type=str, default="", help="local path of the diffusers AutoencoderKL weights; empty: save latents")
    g = p.add_argument_group("Transport arguments")
    g.add_argument("--path-type", type=str, default="Linear", choices=["Linear", "GVP", "VP"])
    g.add_argument("--prediction", type=str, default="velocity", choices=["velocity", "score", "noise"])
    g.add_argument("--loss-weight", type=none_or_str, default=None, choices=[None, "velocity", "likelihood"])
    g.add_argument("--sample-eps", type=float)
    g.add_argument("--train-eps", type=float)
    g = p.add_argument_group("ODE arguments")
    g.add_argument("--sampling-method", type=str, default="euler")
    g.add_argument("--atol", type=float, default=1e-6)
    g.add_argument("--rtol", type=float, default=1e-3)
    g.add_argument("--reverse", action="store_true")
    return p


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
