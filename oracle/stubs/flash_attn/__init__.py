"""Stub of ``flash_attn`` (not installed).  TEST INFRASTRUCTURE ONLY.  The fp32 CPU path of the reference
never reaches these (model.py:378 takes the SDPA branch); they raise if called."""


def _absent(*a, **k):
    raise RuntimeError("flash_attn is not available in the oracle harness (fp32 path must not call it)")


flash_attn_func = _absent
flash_attn_varlen_func = _absent
