"""Helpers for the -m gpu parity tests: call the C-ABI operator entry points with torch tensors."""
import ctypes as C

import torch

import lumina_t2x_amd  # noqa: F401
from lumina_t2x_amd import _lib


def P(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf(t):
    return t.to(device="cuda", dtype=torch.bfloat16).contiguous()


def r16(t):
    """bf16 round trip in fp32 (the reference's rounding points)"""
    return t.to(torch.bfloat16).to(torch.float32)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def lib():
    return _lib.load()


def ok(rc, what=""):
    _lib.check(rc, what)


def set_option(name, value):
    ok(lib().lt_set_option(name.encode(), int(value)), f"lt_set_option({name})")
