"""CPU: the C-ABI library loads and exports every symbol include/lumina_dit.h declares; argument validation
errors come back as codes + messages (no GPU work is issued here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import lumina_t2x_amd  # noqa: F401  (import shim)
from lumina_t2x_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    declared = _lib.declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lumina_dit.h but not exported"
    # the python binding table covers the header one to one
    assert sorted(_lib._SIGNATURES) == declared


def test_binding_table_matches_header_prototypes():
    """argument COUNT and coarse kind (pointer / 32-bit / 64-bit / float) of every ctypes signature against the prototype in
    include/lumina_dit.h - an ABI drift between the header and the Python binding would otherwise only show up as memory
    corruption on the GPU box."""
    text = _lib.header_text()
    protos = dict(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(protos) == set(_lib._SIGNATURES)

    def kind_of_c(param):
        p = " ".join(param.split())
        if p in ("void", ""):
            return None
        if "*" in p:
            return "ptr"
        if p.startswith("int64_t"):
            return "i64"
        if p.startswith("float"):
            return "f32"
        if p.startswith(("int32_t", "uint32_t", "int ")):
            return "i32"
        raise AssertionError(f"unclassified C parameter '{p}'")

    def kind_of_ctype(t):
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents"):
            return "ptr"
        return {C.c_int32: "i32", C.c_uint32: "i32", C.c_int64: "i64", C.c_float: "f32"}[t]

    for name, params in protos.items():
        want = [k for k in (kind_of_c(p) for p in params.split(",")) if k]
        got = [kind_of_ctype(t) for t in _lib._SIGNATURES[name][1]]
        assert got == want, (name, got, want)


def test_version_and_error_channel(lib):
    assert lib.lt_version().decode().startswith("lumina_dit gfx950")
    cfg = _lib.LtConfig(variant=0, dim=100, n_layers=1, n_heads=3, n_kv_heads=3, ffn_hidden=256, patch_size=2,
                        in_channels=4, out_channels=8, cap_feat_dim=64, adaln_dim=100, qk_norm=1, num_classes=0,
                        norm_eps=1e-5, max_batch=2, max_tokens=64, max_text=64, rope_table_len=384)
    handle = C.c_void_p()
    rc = lib.lt_create(C.byref(cfg), C.byref(handle))
    assert rc != 0 and not handle.value
    assert b"n_heads" in lib.lt_last_error() or b"head_dim" in lib.lt_last_error()
    with pytest.raises(_lib.LuminaLibError):
        _lib.check(rc, "lt_create")


def test_bad_variant_rejected(lib):
    cfg = _lib.LtConfig(variant=7, dim=576, n_layers=1, n_heads=8, n_kv_heads=8, ffn_hidden=1536, patch_size=2,
                        in_channels=4, out_channels=8, cap_feat_dim=128, adaln_dim=576, qk_norm=1, num_classes=0,
                        norm_eps=1e-5, max_batch=2, max_tokens=64, max_text=64, rope_table_len=384)
    handle = C.c_void_p()
    assert lib.lt_create(C.byref(cfg), C.byref(handle)) != 0
    assert b"variant" in lib.lt_last_error()


@pytest.mark.parametrize("field,value,needle", [("n_heads", 0, b"positive"), ("n_kv_heads", 0, b"positive"), ("n_layers", 0, b"positive"),
                                                ("patch_size", 0, b"positive"), ("max_tokens", 0, b"positive"), ("dim", -64, b"positive"),
                                                ("max_batch", 9, b"max_batch"), ("cap_feat_dim", 100, b"cap_feat_dim")])
def test_degenerate_configs_rejected_before_any_allocation(lib, field, value, needle):
    """A zero head count used to reach `dim % n_heads`; every degenerate field is an error message now, not a SIGFPE."""
    kw = dict(variant=0, dim=576, n_layers=1, n_heads=8, n_kv_heads=8, ffn_hidden=1536, patch_size=2, in_channels=4, out_channels=8,
              cap_feat_dim=128, adaln_dim=576, qk_norm=1, num_classes=0, norm_eps=1e-5, max_batch=2, max_tokens=64, max_text=64,
              rope_table_len=384)
    kw[field] = value
    cfg = _lib.LtConfig(**kw)
    handle = C.c_void_p()
    assert lib.lt_create(C.byref(cfg), C.byref(handle)) != 0 and not handle.value
    assert needle in lib.lt_last_error(), lib.lt_last_error()


def test_null_arguments_are_errors_not_crashes(lib):
    assert lib.lt_forward(None, None, None, None, None, None) != 0
    assert lib.lt_op_gemm_bf16(None, None, None, -1, None, 1, 8, 64, 0, 0, None) != 0
    assert lib.lt_weights_ready(None) != 0
    assert lib.lt_forward_packed(None, None, None, None, None, None, None) != 0
    assert lib.lt_prepare_prompt_regional(None, None, 1, None, 3, 16, None, None, 16, 1, 2, None) != 0
    assert lib.lt_op_gemm_grouped(None, None, None, 0, None, 256, 8, 64, 0, 0, None) != 0
    assert b"null" in lib.lt_last_error()
    assert lib.lt_set_option(b"no_such_knob", 1) != 0 and b"no_such_knob" in lib.lt_last_error()


def test_product_path_never_imports_the_oracle():
    """A product path that routes through oracle/ voids parity claims - enforce statically."""
    pkg = os.path.join(REPO, "lumina-t2x_amd")
    bad = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "import_module(\"oracle" in text:
                    bad.append(os.path.join(root, f))
                # ... nor the travelling copy of the reference modules (oracle/_ref, the CPU-baseline leg of bench.py) or the checkout
                if "_ref" in text and re.search(r"oracle[/.\\]_ref", text):
                    bad.append(os.path.join(root, f))
                if "/root/reference" in text and f.endswith(".py"):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_reference_archive_for_the_cpu_baseline_is_byte_identical_or_absent():
    """oracle/_ref/reference_files.tar (git-ignored build artefact, packed by oracle/build_ref.py from __graft_entry__.build()): when present
    the archive must be the one its manifest describes and every member must still carry the sha256 of the reference file it was read from;
    in the authoring container the sources are compared directly.  No loose reference source may sit in the tree, nothing of it in git."""
    import hashlib
    import json
    from oracle import build_ref
    if not os.path.exists(build_ref.MANIFEST):
        pytest.skip("oracle/_ref not built (no reference checkout was present when build() ran)")
    assert build_ref.verify()
    root = build_ref.extract()
    man = json.load(open(build_ref.MANIFEST))["sha256"]
    for rel, h in man.items():
        assert hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest() == h, rel
        if os.path.isdir(os.path.join(build_ref.REF_SRC, "lumina_next_t2i")):
            assert hashlib.sha256(open(os.path.join(build_ref.REF_SRC, rel), "rb").read()).hexdigest() == h, rel
    assert sorted(os.listdir(build_ref.REF_DST)) == ["MANIFEST.json", "reference_files.tar"], os.listdir(build_ref.REF_DST)
    tracked = subprocess.run(["git", "-C", REPO, "ls-files", "oracle/_ref"], capture_output=True, text=True).stdout.strip()
    assert tracked == "", "reference sources must never enter git history: " + tracked


def test_missing_library_fails_loudly(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import lumina_t2x_amd\n"
        "from lumina_t2x_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "try:\n"
        "    _lib.load()\n"
        "except _lib.LuminaLibError as e:\n"
        "    print('LOUD', e)\n" % (REPO, str(tmp_path / "nope.so"))
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "LOUD" in out.stdout and "no CPU/PyTorch fallback" in out.stdout, out.stdout + out.stderr


def _option_table():
    """(name, lo, hi, default, boolean) rows of kLtOptDesc in csrc/options.hip"""
    src = open(os.path.join(REPO, "lumina-t2x_amd", "csrc", "options.hip")).read()
    body = src[src.index("const LtOptDesc kLtOptDesc"):]
    body = body[:body.index("};")]
    rows = re.findall(r'\{"([a-z_0-9]+)",\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(true|false)\}', body)
    return [(n, int(lo), int(hi), int(d), b == "true") for n, lo, hi, d, b in rows]


def test_every_option_is_documented_in_the_debug_header_with_its_range_and_default():
    """the options live in ONE table (csrc/options.hip); include/lumina_dit_debug.h is where a reader learns what they do.  Every
    name must appear there with the same (range, default); the boundary header documents none of them (they are not part of the
    drop-in surface) but declares the three option entry points."""
    table = _option_table()
    names = {r[0] for r in table}
    assert {"attn_q_fused", "gemm_prefetch", "graph", "attention_variant", "gemm_splitk", "gemm_w4q_grouped", "rmsnorm_apex"} <= names, names
    debug = open(_lib.DEBUG_HEADER_PATH).read()
    for n, lo, hi, d, _ in table:
        assert re.search(r'"%s"\s*\(%d\.\.%d, %d\)' % (n, lo, hi, d), debug), f'option "{n}" ({lo}..{hi}, {d}) is not documented as such in lumina_dit_debug.h'
    enum = open(os.path.join(REPO, "lumina-t2x_amd", "csrc", "options.h")).read()
    assert len(table) == len(re.findall(r"\bOPT_[A-Z0-9_]+", enum[enum.index("enum LtOpt"):enum.index("LT_OPT_COUNT")]))
    for n in ("gemm_pipeline", "gemm_pp_tail", "gemm_persist"):  # accepted with value 0 only; named as removed
        assert n in debug
    boundary = open(_lib.HEADER_PATH).read()
    assert all(f'"{n}"' not in boundary for n in names - {"graph"})  # (lt_graph_replays' comment names the one option it depends on)
    assert all(fn in boundary for fn in ("lt_set_option", "lt_engine_set_option", "lt_engine_get_option"))


def test_options_validate_ranges_and_engine_overrides_do_not_leak(lib):
    """gemm_prefetch / gemm_splitk / gemm_w4q_grouped took any integer.  Every option now validates against the table; an
    engine override is visible through that engine only and LT_OPTION_INHERIT drops it (no GPU needed: lt_create fails here without a
    device, so the engine-side half runs with e = NULL -> process default, and on the GPU box in tests/test_gpu_model.py)."""
    val = C.c_int32(-7)
    for n, lo, hi, d, boolean in _option_table():
        assert lib.lt_engine_get_option(None, n.encode(), C.byref(val)) == 0 and val.value == d, (n, val.value, d)
        if not boolean:
            assert lib.lt_set_option(n.encode(), hi + 1) != 0 and n.encode() in lib.lt_last_error(), n
            assert lib.lt_set_option(n.encode(), lo - 1) != 0
        assert lib.lt_engine_get_option(None, n.encode(), C.byref(val)) == 0 and val.value == d  # a refused value changes nothing
        assert lib.lt_set_option(n.encode(), hi) == 0
        assert lib.lt_engine_get_option(None, n.encode(), C.byref(val)) == 0 and val.value == hi
        assert lib.lt_set_option(n.encode(), d) == 0
    assert lib.lt_set_option(b"attention_variant", 5) != 0 and b"removed" in lib.lt_last_error()
    assert lib.lt_set_option(b"gemm_prefetch", 2) != 0 and b"removed" in lib.lt_last_error()
    assert lib.lt_set_option(b"gemm_persist", 0) == 0 and lib.lt_set_option(b"gemm_persist", 1) != 0
    assert lib.lt_engine_set_option(None, b"graph", 0) != 0 and b"null engine" in lib.lt_last_error()
    assert lib.lt_engine_get_option(None, b"no_such", C.byref(val)) != 0
