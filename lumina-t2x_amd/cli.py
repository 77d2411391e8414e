"""``lumina_next`` command line on the MI355X engine - grammar and behaviour of the reference's click entry point
(``lumina_next_t2i/entry_point.py:93-156`` + ``utils/cli.py:161-333``):

    python -m lumina_t2x_amd.cli [infer] "a caption" [output_dir] --ckpt DIR --ckpt_lm DIR [-c settings.yaml] [--ema] [--precision bf16]
    python -m lumina_t2x_amd.cli convert weights.pth out_dir        # .pth <-> .safetensors

``infer`` is the default command, as in the reference.  The yaml file has the reference's layout (``configs/infer/settings.yaml``:
``model`` / ``transport`` / ``ode`` / ``infer`` sections under one list item); ``plan_inference`` turns it into the sampler and
model arguments exactly as ``utils/cli.py:inference`` does, including its conventions that differ from ``sample.py``: the latent
is drawn ``[1, 4, h/8, w/8]`` (height first), "(Extrapolation)" in the resolution string switches time-aware scaling on, the
tokenizer is built without ``add_eos``.  The shipped yaml spells the watershed key ``scale_watershed`` while the reference code
reads ``scaling_watershed`` (so the reference's own default config raises a KeyError that its blanket ``except`` swallows);
both spellings are accepted here.

``--family flag`` is the same command for the Flag-DiT sub-project (``lumina_t2i/entry_point.py``, ``utils/cli.py:124-240``: the
``lumina`` console script): its yaml has ``ntk_scaling`` instead of the time-aware scaling keys, the proportional-attention base
length counts the end-of-line tokens, and the caption goes through ``tokenizer.encode`` / ``last_hidden_state`` with a bool mask.

The DiT runs on the HIP engine (``models.NextDiT``); text encoder and VAE stay third-party and load from LOCAL paths
(``--ckpt_lm``, ``--vae``): there is no network here, so the reference's hub names are not resolved.  Without ``diffusers``
the final latent is written as ``.pt``.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from typing import Optional

import torch

COMMANDS = ("infer", "convert")


# ---- convert (entry_point.py:116-156) -------------------------------------------------------------------------------------
def convert(weight_path: str, output_dir: str) -> str:
    """``.pth`` -> ``.safetensors`` or back; returns the path written (same file stem, inside ``output_dir``)."""
    from safetensors.torch import load_file, save_file

    stem, ext = os.path.splitext(weight_path)
    if ext not in (".pth", ".safetensors"):
        raise ValueError("Only ('.pth', '.safetensors') models are supported for conversion.")
    name = os.path.basename(stem)
    print(f"Loading your current `{ext}` model {weight_path}")
    os.makedirs(output_dir, exist_ok=True)
    if ext == ".pth":
        out = os.path.join(output_dir, name + ".safetensors")
        state = torch.load(weight_path, map_location="cpu", weights_only=True)
        # safetensors refuses tensors that share storage or are not contiguous; a consolidated checkpoint has neither, but a
        # state_dict saved straight from a module with tied views would - make the copy explicit instead of failing late
        save_file({k: v.contiguous().clone() if not v.is_contiguous() else v for k, v in state.items()}, out)
        print(f"Saving model with `.safetensors` format at {output_dir}")
    else:
        out = os.path.join(output_dir, name + ".pth")
        torch.save(load_file(weight_path, device="cpu"), out)
        print(f"Saving model with `{ext}` format at {output_dir}")
    print("Done.")
    return out


# ---- infer: settings -> plan (utils/cli.py:161-263) -------------------------------------------------------------------------
def load_settings(config_path: str) -> dict:
    import yaml

    with open(config_path, "r") as f:
        return yaml.safe_load(f)[0]


def plan_inference(config: dict, image_size: int, family: str = "next") -> dict:
    """Everything ``utils/cli.py:inference`` derives from the yaml before it touches a model (``family``: "next" =
    lumina_next_t2i, "flag" = lumina_t2i)."""
    tr, ode, inf = config["transport"], config["ode"], config["infer"]
    resolution = str(inf["resolution"])
    extrapolate = "Extrapolation" in resolution
    w, h = (int(v) for v in resolution.split(" ")[-1].split("x"))
    prop = bool(inf["proportional_attn"])
    model_kwargs = dict(cfg_scale=float(inf["cfg_scale"]))
    if family == "flag":  # lumina_t2i/utils/cli.py:204-214: keys are only present when switched on
        if prop:
            model_kwargs.update(proportional_attn=True, base_seqlen=(image_size // 16) ** 2 + (image_size // 16) * 2)
        if inf["ntk_scaling"]:
            model_kwargs["ntk_factor"] = ((w // 16) * (h // 16)) / ((image_size // 16) ** 2)
    else:
        watershed = inf["scaling_watershed"] if "scaling_watershed" in inf else inf.get("scale_watershed", 1.0)
        model_kwargs.update(proportional_attn=prop, base_seqlen=(image_size // 16) ** 2 if prop else None)
        if extrapolate and inf["scaling_method"] == "Time-aware":
            model_kwargs.update(scale_factor=math.sqrt(w * h / image_size ** 2), scale_watershed=float(watershed))
        else:
            model_kwargs.update(scale_factor=1.0, scale_watershed=1.0)
    # ode.likelihood is read and never used by the reference's t2i command line (utils/cli.py:175; only the ImageNet / MoE
    # sample.py scripts call sample_ode_likelihood): same here
    return dict(
        transport=dict(path_type=tr["path_type"], prediction=tr["prediction"], loss_weight=tr["loss_weight"],
                       train_eps=tr["train_eps"], sample_eps=tr["sample_eps"]),
        sampler=dict(sampling_method=inf["solver"], num_steps=int(inf["num_sampling_steps"]), atol=float(ode["atol"]),
                     rtol=float(ode["rtol"]), reverse=bool(ode["reverse"]), time_shifting_factor=inf["t_shift"]),
        latent_shape=[1, 4, h // 8, w // 8], width=w, height=h, seed=int(inf["seed"]), model_kwargs=model_kwargs)


def resolve_paths(ckpt, ckpt_lm, token, config: dict):
    """cli options win; otherwise all three come from the yaml's ``model`` section (utils/cli.py:291-304)."""
    mc = config["model"]
    if ckpt is None or ckpt_lm is None or token is None:
        if mc.get("ckpt") is None or mc.get("ckpt_lm") is None or mc.get("token") is None:
            raise ValueError("please setting correct model path in yaml config, or pass `--ckpt`, `--ckpt_lm`, `--token` as cli options.")
        return mc["ckpt"], mc["ckpt_lm"], mc["token"]
    print("> loading model path from cli options.")
    return ckpt, ckpt_lm, token


def output_name(cap: str, now: Optional[time.struct_time] = None) -> str:
    """``<caption words joined by _ up to the first '.'>_<timestamp>_lumina`` (utils/cli.py:329-333)."""
    stamp = time.strftime("%Y-%m-%d-%H-%M-%S", now or time.localtime())
    return f"{'_'.join(cap.split(' ')).split('.')[0]}_{stamp}_lumina"


def make_llama_encoder(path: str, dtype, device):
    """Flag-DiT's caption path (lumina_t2i/utils/cli.py:86, :186-202): ``tokenizer.encode`` with BOS + EOS, both prompts padded
    with zeros to the longer one, ``last_hidden_state`` of the LM, bool mask."""
    from transformers import AutoModel, AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path, add_bos_token=True, add_eos_token=True)
    tok.padding_side = "right"
    lm = AutoModel.from_pretrained(path, torch_dtype=dtype).to(device).eval()

    @torch.no_grad()
    def encode(captions):
        ids = [tok.encode(c, truncation=False) for c in captions]
        t = torch.zeros([len(ids), max(len(i) for i in ids)], dtype=torch.long, device=device)
        m = torch.zeros_like(t, dtype=torch.bool)
        for r, i in enumerate(ids):
            t[r, : len(i)] = torch.tensor(i)
            m[r, : len(i)] = True
        return lm(input_ids=t).last_hidden_state, m

    return encode, lm.config.hidden_size


def infer(text: str, output_path: str, *, ckpt, ckpt_lm, ema: bool, precision: str, config_path: str, token=False, num_gpus: int = 1,
          vae: str = "", family: str = "next", encode_fn=None, cap_feat_dim=None, decode_fn=None, model=None) -> str:
    """One caption -> one image (or latent) file; returns its path.  ``encode_fn`` / ``decode_fn`` / ``model`` can be injected."""
    from . import models
    if family == "flag":
        from .models import flag_dit as models  # noqa: F811  (each reference sub-project calls its own package `models`)
    from .sample import VAE_SCALE, load_checkpoint, load_train_args, make_text_encoder, make_vae_decoder, save_png
    from .transport import Sampler, create_transport

    if num_gpus != 1:
        raise NotImplementedError("Multi-GPU Inference is not yet supported")  # as the reference (utils/cli.py:91-92)
    torch.set_grad_enabled(False)
    print("> loading inference settings.")
    config = load_settings(config_path)
    ckpt, ckpt_lm, token = resolve_paths(ckpt, ckpt_lm, token, config)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[precision]
    train_args = load_train_args(ckpt)
    if encode_fn is None:
        encode_fn, cap_feat_dim = (make_llama_encoder(ckpt_lm, dtype, device) if family == "flag"
                                   else make_text_encoder(ckpt_lm, dtype, device, add_eos=False))
    if model is None:
        print(f"> Creating DiT model: {train_args.model}")
        model = models.__dict__[train_args.model](qk_norm=train_args.qk_norm, cap_feat_dim=cap_feat_dim)
        model.eval().to(device, dtype=dtype)
        model.load_state_dict(load_checkpoint(ckpt, ema), strict=True)
    if decode_fn is None:
        decode_fn = make_vae_decoder(vae, device)
    plan = plan_inference(config, train_args.image_size, family)
    sample_fn = Sampler(create_transport(**plan["transport"])).sample_ode(**plan["sampler"])
    if plan["seed"] != 0:
        torch.random.manual_seed(plan["seed"])
    z = torch.randn(plan["latent_shape"], device=device).to(dtype).repeat(2, 1, 1, 1)
    cap_feats, cap_mask = encode_fn([text, ""])
    print(f"> Caption: {text}\n> Num_sampling_steps: {plan['sampler']['num_steps']}\n> Cfg_scale: {plan['model_kwargs']['cfg_scale']}")
    latent = sample_fn(z, model.forward_with_cfg, cap_feats=cap_feats, cap_mask=cap_mask.to(cap_feats.device), **plan["model_kwargs"])[-1][:1]
    factor = VAE_SCALE.get(getattr(train_args, "vae", "sdxl"), 0.18215)
    print(f"> VAE factor: {factor}")
    os.makedirs(output_path, exist_ok=True)
    stem = os.path.join(output_path, output_name(text))
    if decode_fn is not None:
        save_png(decode_fn(latent / factor)[0], stem + ".png")
        out = stem + ".png"
    else:
        print("> no VAE decoder (diffusers or --vae missing): writing the final latent as .pt")
        torch.save(latent.cpu(), stem + ".pt")
        out = stem + ".pt"
    print(f"> Image saved in {output_path}.")
    return out


# ---- command line -----------------------------------------------------------------------------------------------------------
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="lumina_next", description="Lumina-Next text-to-image on the MI355X engine")
    p.add_argument("-v", "--version", action="version", version="1.0.0")
    sub = p.add_subparsers(dest="command")
    pi = sub.add_parser("infer", help="(default) sample one image for a caption")
    pi.add_argument("text", type=str)
    pi.add_argument("output_path", type=str, nargs="?", default="./")
    pi.add_argument("--num_gpus", type=int, default=1, help="number of gpus you want to use.")
    pi.add_argument("--ckpt", type=str, default=None, help="pretrained Lumina-T2X model checkpoint path.")
    pi.add_argument("--ckpt_lm", type=str, default=None, help="pretrained LLM model checkpoint path.")
    pi.add_argument("--ema", action="store_true", help="whether to load ema model.")
    pi.add_argument("--precision", type=str, choices=["bf16", "fp32"], default="bf16", help="precision of inference for model.")
    pi.add_argument("-c", "--config", type=str, default="cofing/infer/settings.yaml",  # (sic) the reference's default
                    help="setting for inference with different parameter.")
    pi.add_argument("--token", default=False, help="huggingface token (unused: weights load from local paths).")
    pi.add_argument("--vae", type=str, default="", help="local path of the diffusers AutoencoderKL weights; empty: save the latent")
    pi.add_argument("--family", type=str, choices=["next", "flag"], default="next",
                    help="next: lumina_next_t2i (`lumina_next`), flag: lumina_t2i / Flag-DiT (`lumina`)")
    pc = sub.add_parser("convert", help="convert torch model weight `.pth` into `.safetensors` (or back)")
    pc.add_argument("weight_path", type=str)
    pc.add_argument("output_dir", type=str)
    return p


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] not in COMMANDS and argv[0] not in ("-h", "--help", "-v", "--version"):
        argv.insert(0, "infer")  # default command, as the reference's DefaultGroup
    a = build_parser().parse_args(argv)
    if a.command == "convert":
        return convert(a.weight_path, a.output_dir)
    if a.command == "infer":
        return infer(a.text, a.output_path, ckpt=a.ckpt, ckpt_lm=a.ckpt_lm, ema=a.ema, precision=a.precision, config_path=a.config,
                     token=a.token, num_gpus=a.num_gpus, vae=a.vae, family=a.family)
    build_parser().print_help()
    return None


if __name__ == "__main__":
    main()
