"""ctypes binding of the C ABI declared in ``include/lumina_dit.h`` (+ the diagnostics of ``include/lumina_dit_debug.h``).

The shared library is built in-tree by ``__graft_entry__.build()`` (``make -C lumina-t2x_amd/csrc``).
There is NO fallback: if the library is missing or a symbol cannot be resolved the import of the
product path fails loudly (the oracle under ``oracle/`` is test infrastructure and is never used here).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
# LUMINA_DIT_LIB: an alternate build of the same C ABI (A/B measurements of two kernel versions on one box); still no fallback
LIB_PATH = os.environ.get("LUMINA_DIT_LIB") or os.path.join(_HERE, "lib", "liblumina_dit.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lumina_dit.h")
DEBUG_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lumina_dit_debug.h")  # options' documentation + one trace entry point
LT_OPTION_INHERIT = -2 ** 31

LT_F32, LT_BF16, LT_F16 = 0, 1, 2
LT_VARIANT_NEXT_T2I, LT_VARIANT_NEXT_IMAGENET, LT_VARIANT_FLAG_T2I, LT_VARIANT_NEXT_MOE = 0, 1, 2, 3
LT_VARIANT_NEXT_MOE_TIME, LT_VARIANT_NEXT_MOE_SPACE = 4, 5
LT_ODE_EULER, LT_ODE_MIDPOINT, LT_ODE_RK4 = 0, 1, 2
ODE_METHODS = {"euler": LT_ODE_EULER, "midpoint": LT_ODE_MIDPOINT, "rk4": LT_ODE_RK4}


class LtConfig(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("dim", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
        ("n_kv_heads", C.c_int32), ("ffn_hidden", C.c_int32), ("patch_size", C.c_int32),
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("cap_feat_dim", C.c_int32),
        ("adaln_dim", C.c_int32), ("qk_norm", C.c_int32), ("num_classes", C.c_int32),
        ("norm_eps", C.c_float), ("max_batch", C.c_int32), ("max_tokens", C.c_int32),
        ("max_text", C.c_int32), ("rope_table_len", C.c_int32), ("num_experts", C.c_int32),
    ]


class LtStepArgs(C.Structure):
    _fields_ = [
        ("cfg_scale", C.c_float), ("scale_factor", C.c_float), ("scale_watershed", C.c_float),
        ("base_seqlen", C.c_int32), ("proportional_attn", C.c_int32), ("latent_h", C.c_int32),
        ("latent_w", C.c_int32), ("batch", C.c_int32), ("io_dtype", C.c_int32), ("cfg_channels", C.c_int32),
        ("ntk_factor", C.c_float),
    ]


class LuminaLibError(RuntimeError):
    pass


def header_text(strip_comments: bool = True) -> str:
    """both headers concatenated (the drop-in boundary, then the debug header)"""
    text = ""
    for path in (HEADER_PATH, DEBUG_HEADER_PATH):
        with open(path) as f:
            text += f.read() + "\n"
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S) if strip_comments else text


def declared_symbols() -> List[str]:
    """Names of every function the two headers declare (used by the ABI-export test)."""
    return sorted(set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", header_text())))


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/lumina_dit.h + lumina_dit_debug.h one to one
_SIGNATURES: Dict[str, tuple] = {
    "lt_last_error": (C.c_char_p, []),
    "lt_version": (C.c_char_p, []),
    "lt_set_option": (_i32, [C.c_char_p, _i32]),
    "lt_engine_set_option": (_i32, [_vp, C.c_char_p, _i32]),
    "lt_engine_get_option": (_i32, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "lt_create": (_i32, [C.POINTER(LtConfig), C.POINTER(_vp)]),
    "lt_destroy": (None, [_vp]),
    "lt_set_weight": (_i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32, _vp]),
    "lt_weights_ready": (_i32, [_vp]),
    "lt_prepare_prompt": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "lt_prepare_prompt_regional": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "lt_prepare_labels": (_i32, [_vp, _vp, _i32, _vp]),
    "lt_forward": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(LtStepArgs), _vp]),
    "lt_forward_packed": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_i32), _vp, C.POINTER(_vp), C.POINTER(LtStepArgs), _vp]),
    "lt_forward_cfg": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(LtStepArgs), _vp]),
    "lt_sample_ode": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(_f32), _i32, _i32, _i32, _i32, C.POINTER(LtStepArgs), _vp]),
    "lt_last_nfe": (_i64, [_vp]),
    "lt_graph_replays": (_i64, [_vp]),
    "lt_moe_routing_record": (_i32, [_vp, _i32]),
    "lt_moe_routing_read": (_i32, [_vp, C.POINTER(_i32), _i32]),
    "lt_moe_routing_force": (_i32, [_vp, C.POINTER(_i32), _i32]),
    "lt_profile_enable": (_i32, [_vp, _i32]),
    "lt_profile_read": (_i32, [_vp, _i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "lt_profile_reset": (_i32, [_vp]),
    "lt_profile_enable_mask": (_i32, [_vp, _i32]),
    "lt_profile_set_budget": (_i32, [_vp, _i32, _i64]),
    "lt_profile_set_window": (_i32, [_vp, _i32, _i64, _i64]),
    "lt_op_gemm_bf16": (_i32, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_pair_layout": (_i32, [_vp, _i64, _i32, _i32, _vp]),
    "lt_op_gemm_bf16_pair": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_gemm_vt": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_gemm_qkv": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_gemm_qkv_fusable": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "lt_op_gemm_describe": (_i32, [_i32, _i32, _i32, _i32, _i32, C.c_char_p, _i32]),
    "lt_op_moe_plan": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "lt_op_gemm_splitk": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "lt_op_gemm_splitk_auto": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "lt_op_gemm_grouped": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_gemm_grouped_tail": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "lt_op_gemm_grouped_gather": (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_pack_w13": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "lt_op_rmsnorm_mod": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lt_op_gated_residual_norm": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "lt_op_prep_mod": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, C.c_uint32, C.c_uint32, _i32, _vp]),
    "lt_op_qk_norm_rope": (_i32, [_vp, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _vp]),
    "lt_op_v_transpose": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_attention": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lt_op_qkv_attention_small": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp]),
    "lt_op_attention_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_attention_trace": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "lt_op_linear_small_m": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "lt_op_rope_table_2d": (_i32, [_vp, _i32, _i32, _f32, _f32, _vp]),
    "lt_op_rope_table_2d_pair": (_i32, [_vp, _vp, _i32, _i32, _f32, _f32, _vp]),
    "lt_op_proj_gated_residual_norm": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "lt_op_qkv_qstat": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp]),
    "lt_op_attention_qraw": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP extension; raises LuminaLibError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LuminaLibError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C lumina-t2x_amd/csrc`). There is no CPU/PyTorch fallback for the denoising path."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime, wrong arch, ...
        raise LuminaLibError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise LuminaLibError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().lt_last_error()
        raise LuminaLibError(f"{what or 'lumina_dit call'} failed (rc={rc}): {msg.decode() if msg else '?'}")
