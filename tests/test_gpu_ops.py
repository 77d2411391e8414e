"""-m gpu: every HIP kernel against a plain PyTorch fp32 reference of the same op, through the C ABI.

Tolerances (stated here, justified in DESIGN.md): GEMM / attention outputs are bf16, so one output ulp is
2^-8 relative; we require rel-L2 <= 4e-3 against an fp32-accumulated reference rounded once to bf16 and a
max-abs error of a few ulp of the largest output.  Pure data-movement kernels must be bit exact.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import P, bf, lib, max_abs, ok, r16, rel_l2, set_option, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_kernel_variants():
    yield
    set_option("attention_variant", 4)
    set_option("gemm_variant", 0)


PRODUCT_VARIANTS = [1, 2, 3, 7, 8]
REMOVED_VARIANTS = [4, 5, 6, 9, 10, 11, 12, 13, 14, 17, 18]  # the study kernels of rounds 1-3 (csrc/experimental/, deleted in round 5; git history keeps them)


def test_removed_gemm_variants_are_refused_by_name():
    A = bf(torch.zeros(256, 128))
    W = bf(torch.zeros(256, 128))
    out = torch.empty(256, 256, device="cuda", dtype=torch.bfloat16)
    for v in REMOVED_VARIANTS:
        assert lib().lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(out), 256, 256, 128, 0, v, stream()) != 0
        assert b"removed" in lib().lt_last_error()


def _variant_params():
    return list(PRODUCT_VARIANTS)


def _gemm(A, W, bias=None, epilogue=0, variant=0):
    M, K = A.shape
    N = W.shape[0]
    out = torch.full((M, N // 2 if epilogue else N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_bf16(P(A), P(W), P(bias), 1, P(out), M, N, K, epilogue, variant, stream()), "gemm")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 576, 576), (300, 256, 128), (1024, 2304, 2304),
                                   (130, 32, 576), (8192, 6912, 2304), (256, 1152, 128), (8192, 2304, 6144)])
def test_gemm_plain(M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    out = _gemm(A, W)
    ref = (A.float() @ W.float().t())  # fp32 reference on the device (rocBLAS), inputs are the same bf16 values
    assert not torch.isnan(out.float()).any(), "unwritten outputs"
    assert rel_l2(out, ref) < 4e-3, rel_l2(out, ref)
    assert max_abs(out, ref) < 0.04 * float(ref.abs().max())


@pytest.mark.parametrize("variant", _variant_params())
@pytest.mark.parametrize("M,N,K", [(8192, 2304, 2304), (300, 576, 128), (1024, 6912, 2304), (256, 288, 64), (257, 296, 192),
                                   (512, 512, 6144)])
def test_gemm_tile_variants(variant, M, N, K):
    """both tile shapes (256x256 / 8 waves, 256x288 / 12 waves), the 8-wave ping-pong loop (3) and the small-M tiles (7 / 8) on
    tile-multiple and ragged problems; K = 64 ... 6144 covers 2 ... 192 ring slabs."""
    if variant == 9 and K < 96:
        pytest.skip("the persistent kernel carries a 3-slab prefetch across tiles: K >= 96")
    g = torch.Generator(device="cpu").manual_seed(M + N + K + variant)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = bf(torch.randn(N, generator=g))
    out = _gemm(A, W, b, variant=variant)
    ref = A.float() @ W.float().t() + b.float()
    assert not torch.isnan(out.float()).any(), "unwritten outputs"
    assert rel_l2(out, ref) < 4e-3, rel_l2(out, ref)


@pytest.mark.parametrize("variant", _variant_params() + [15, 16])
def test_gemm_identity_asymmetric(variant):
    """A = I with an asymmetric W catches a transposed / permuted C-write (guide 5.4 rule 16)."""
    K, N = 320, 576
    A = bf(torch.eye(320, K))
    W = bf((torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 64.0)
    out = _gemm(A, W, variant=variant)
    assert torch.equal(out.float().cpu(), W.float().t().cpu())


def test_gemm_bias_and_edges():
    g = torch.Generator().manual_seed(5)
    A = bf(torch.randn(200, 128, generator=g))
    W = bf(torch.randn(40, 128, generator=g) / 11)
    b = bf(torch.randn(40, generator=g))
    out = _gemm(A, W, b)
    ref = A.float() @ W.float().t() + b.float()
    assert rel_l2(out, ref) < 4e-3
    # guard: a larger buffer around the output must stay untouched (no stray stores past M or N)
    big = torch.full((264, 40), 7.0, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_bf16(P(A), P(W), P(b), 1, P(big), 200, 40, 128, 0, 0, stream()))
    torch.cuda.synchronize()
    assert torch.all(big[200:] == 7.0)


@pytest.mark.parametrize("variant", [0, 1, 3, 7, 15])  # auto, classic loop, 8-wave ping-pong, small tiles, persistent 4 waves (16x16x32 MFMA)
@pytest.mark.parametrize("M,F_,K", [(256, 128, 64), (300, 1536, 576), (4096, 6144, 2304), (8192, 6144, 2304)])
def test_gemm_swiglu(M, F_, K, variant):
    if variant in (14, 15) and K < 128:
        pytest.skip("persistent 4-wave kernels: K >= 128")
    g = torch.Generator().manual_seed(F_ + K)
    A = bf(torch.randn(M, K, generator=g))
    w1 = bf(torch.randn(F_, K, generator=g) / math.sqrt(K))
    w3 = bf(torch.randn(F_, K, generator=g) / math.sqrt(K))
    packed = torch.empty(2 * F_, K, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_pack_w13(P(w1), P(w3), P(packed), F_, K, stream()))
    out = _gemm(A, packed, None, 1, variant=variant)
    a = r16(A.float() @ w1.float().t())
    b = r16(A.float() @ w3.float().t())
    ref = r16(r16(F.silu(a)) * b)
    assert rel_l2(out, ref) < 6e-3, rel_l2(out, ref)


@pytest.mark.parametrize("M,N,K,epi", [(8192, 12288, 2304, 1), (8300, 6912, 2304, 0), (2100, 1280, 128, 1), (512, 512, 128, 0),
                                       (300, 576, 192, 0), (16384, 2304, 6144, 0), (70000, 520, 256, 0), (256, 131072, 128, 1),
                                       (8320, 3072, 3072, 0), (8192, 2304, 2304, 0)])
@pytest.mark.parametrize("variant", [15, 16])
def test_gemm_4wave_persistent_16x16x32(M, N, K, epi, variant):
    """gemm_bf16_w4q: the persistent 4-wave structure on v_mfma_f32_16x16x32_bf16 with 256-wide (15) or 288-wide (16) tiles -
    XOR-swizzled 16-row fragment reads, v_permlane16_swap epilogue, ragged M / N against both tile widths, several tiles per
    CU and fewer tiles than CUs, K = 128 (four slabs).  One K = 32 MFMA per slab instead of two K = 16 ones: equal to the classic
    kernel up to fp32 summation order (compared after the same bf16 rounding), and to the fp32 reference at the GEMM tolerance."""
    if epi == 1 and variant == 16:
        pytest.skip("SwiGLU pairs 32-column groups: 256-wide tiles only")
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    got = _gemm(A, W, None, epi, variant=variant)
    ref = _gemm(A, W, None, epi, variant=1)
    assert not torch.isnan(got.float()).any(), "unwritten outputs"
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)
    assert float((got.float() != ref.float()).float().mean()) < 0.05  # different summation order flips the odd last bit, no more
    if epi == 0:
        assert rel_l2(got, A.float() @ W.float().t()) < 4e-3


@pytest.mark.parametrize("variant", [0, 3, 7])
@pytest.mark.parametrize("epilogue", [0, 1])
def test_gemm_grouped_gather_on_load(variant, epilogue):
    """round 3: the experts' GEMM reads its rows through the routing plan's inverse map (sorted position -> token row, -1 = padding)
    instead of a gathered copy - must equal, bit for bit, the same kernel on a gathered copy; every token appears twice (top-2),
    ragged segments, a padding-only tile, padding rows inside real tiles leave zeros x W = 0 rows (they are never read back)."""
    E, K, N, T = 4, 1536, 512, 600
    te = [2, 2, 0, -1, 3, 1, 1, -1]
    M = 256 * len(te)
    g = torch.Generator().manual_seed(41 + variant + epilogue)
    X = bf(torch.randn(T, K, generator=g))
    W = bf(torch.randn(E, N, K, generator=g) / math.sqrt(K))
    row_map = torch.full((M,), -1, dtype=torch.int32)
    fill = {0: 256, 1: 256, 2: 200, 4: 131, 5: 256, 6: 97}  # entries per real tile
    perm = torch.randperm(T, generator=g)
    cursor = 0
    for t_, cnt in fill.items():
        idx = torch.cat([perm, perm])[cursor:cursor + cnt]
        cursor += cnt
        row_map[256 * t_: 256 * t_ + cnt] = idx.to(torch.int32)
    assert cursor == 2 * T - 4  # (almost) every token twice
    gathered = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
    valid = row_map >= 0
    gathered[valid.cuda()] = X[row_map[valid].long().cuda()]
    tile_expert = torch.tensor(te, dtype=torch.int32, device="cuda")
    No = N // 2 if epilogue else N
    want = torch.full((M, No), 3.0, device="cuda", dtype=torch.bfloat16)
    got = torch.full((M, No), 3.0, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_grouped(P(gathered), P(W), P(tile_expert), N * K, P(want), M, N, K, epilogue, variant, stream()), "grouped")
    ok(lib().lt_op_gemm_grouped_gather(P(X), T, P(row_map.cuda()), P(W), P(tile_expert), N * K, P(got), M, N, K, epilogue, variant, stream()),
       "grouped_gather")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.all(got[256 * 3: 256 * 4] == 3.0) and torch.all(got[256 * 7:] == 3.0)
    y = X[row_map[256 * 4: 256 * 4 + 131].long().cuda()].float() @ W[3].float().t()
    if not epilogue:
        assert rel_l2(got[256 * 4: 256 * 4 + 131], y) < 4e-3


@pytest.mark.parametrize("M,N,K,force", [(512, 1536, 1536, False), (512, 1536, 4096, False), (500, 1530 // 8 * 8, 1024, False), (64, 128, 1024, False),
                                         (1024, 2304, 2048, True), (512, 1536, 768, False)])
def test_gemm_splitk_small_m(M, N, K, force):
    """round 4: the 512-row O / W2 projections of the 600M models with their K range split over two workgroups per 64 x 128 tile
    (lt_op_gemm_splitk: the second-arriving half adds the first one's fp32 partial, counter per tile).  Against the unsplit kernel on
    the same tile (equal to fp32 rounding of one addition) and the fp32 reference; ragged M / N; a shape that must NOT split (K = 768:
    runs unsplit, bit-identical); `force` = more tiles than one round (gemm_splitk 2); the counters are back at zero afterwards and a
    second launch on the same workspace gives the same bits (the result cannot depend on arrival order)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    tiles = ((M + 63) // 64) * ((N + 127) // 128)
    part = torch.full((tiles * 2 * 64 * 128,), float("nan"), device="cuda", dtype=torch.float32)
    cnt = torch.zeros(tiles, device="cuda", dtype=torch.int32)
    plain = _gemm(A, W, variant=8)
    if force:
        set_option("gemm_splitk", 2)
    try:
        outs = []
        for _ in range(2):
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            ok(lib().lt_op_gemm_splitk(P(A), P(W), P(out), M, N, K, P(part), P(cnt), tiles, stream()), "gemm_splitk")
            torch.cuda.synchronize()
            outs.append(out)
            assert int(cnt.abs().sum()) == 0, "a tile counter was left non-zero"
    finally:
        set_option("gemm_splitk", 1)
    assert torch.equal(outs[0], outs[1])
    split_ran = not torch.isnan(part).all()
    assert split_ran == (K >= 1024 and K % 512 == 0 and (force or 2 * tiles <= 256))
    if not split_ran:
        assert torch.equal(outs[0], plain)
    ref = A.float() @ W.float().t()
    assert rel_l2(outs[0], ref) < 4e-3, rel_l2(outs[0], ref)
    assert rel_l2(outs[0], plain) < 2e-3, rel_l2(outs[0], plain)


@pytest.mark.parametrize("M,N,K,parts", [(512, 1536, 4096, 4),    # the 600M models' w2: 48 tiles of 128 x 128, four K quarters each = 192 workgroups
                                          (500, 1528, 8192, 4),    # ragged M / N, 64 slabs per quarter
                                          (512, 1536, 1536, 2),    # their O projection: K below the four-way bound -> two halves on 64 x 128 tiles
                                          (1024, 2304, 4096, 0)])  # 8 x 18 tiles x 4 > #CUs: unsplit
def test_gemm_splitk_four_ways_on_128_tiles(M, N, K, parts):
    """round 5 (option gemm_splitk4): K split four ways on 128 x 128 tiles, the last arriver sums the four fp32 partials in K order.  Through
    the launcher's own decision (lt_op_gemm_splitk_auto = what the engine calls): against the unsplit kernel (fp32 rounding of three
    additions) and the fp32 reference; counters back at zero; the same bits from a second launch and from 40 more (arrival order varies);
    option off -> the round-4 two-way form."""
    g = torch.Generator().manual_seed(M + N + K + 5)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    slots = 256
    part = torch.full((slots * 2 * 64 * 128,), float("nan"), device="cuda", dtype=torch.float32)
    cnt = torch.zeros(slots, device="cuda", dtype=torch.int32)
    plain = _gemm(A, W, variant=7)

    def run():
        out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ok(lib().lt_op_gemm_splitk_auto(P(A), P(W), P(out), M, N, K, P(part), P(cnt), slots, stream()), "gemm_splitk_auto")
        return out

    first = run()
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0, "a tile counter was left non-zero"
    touched = int((~torch.isnan(part)).sum())
    t128, t64 = ((M + 127) // 128) * ((N + 127) // 128), ((M + 63) // 64) * ((N + 127) // 128)
    assert touched == {4: t128 * 4 * 128 * 128, 2: t64 * 2 * 64 * 128, 0: 0}[parts], (touched, parts)
    for _ in range(40):
        assert torch.equal(run(), first)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    assert not torch.isnan(first.float()).any()
    assert rel_l2(first, A.float() @ W.float().t()) < 4e-3
    assert rel_l2(first, plain) < 2e-3, rel_l2(first, plain)
    if parts == 4:
        set_option("gemm_splitk4", 0)
        try:
            part.fill_(float("nan"))
            two = run()
            torch.cuda.synchronize()
        finally:
            set_option("gemm_splitk4", 1)
        assert int((~torch.isnan(part)).sum()) in (0, t64 * 2 * 64 * 128)  # two halves on 64 x 128 tiles where those fit one round, else unsplit
        assert rel_l2(two, first) < 2e-3


def test_gemm_splitk_handoff_stress():
    """the split-K halves of a tile hand their fp32 partials over between two workgroups that may sit on different XCDs (sc0 sc1
    stores -> s_waitcnt vmcnt(0) -> relaxed system-scope counter -> sc0 sc1 loads, no L2-wide fence).  A stale read would silently corrupt
    the O / W2 outputs of the 512-row models.  3000 back-to-back split launches (both engine shapes + a forced two-round shape: uneven
    arrival) while a second stream streams 256 MiB copies through every L2; EVERY output word of EVERY launch must equal the first
    launch's (the sum of two fp32 partials does not depend on which half arrives second) and agree with the unsplit kernel to fp32
    rounding of one addition."""
    thrash_src = torch.empty(1 << 26, device="cuda", dtype=torch.float32).normal_()  # 256 MiB: eight times all L2s together
    thrash_dst = torch.empty_like(thrash_src)
    side = torch.cuda.Stream()
    bad = torch.zeros(1, device="cuda", dtype=torch.int32)
    for M, N, K, force in ((512, 1536, 1536, False), (512, 1536, 4096, False), (1024, 2304, 2048, True), (512, 1536, 4096, "four")):
        g = torch.Generator().manual_seed(M + N + K + 1)
        A = bf(torch.randn(M, K, generator=g))
        W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
        tiles = ((M + 63) // 64) * ((N + 127) // 128)
        four = force == "four"  # round 5: the same problem as the launcher runs it by default - four K quarters on 128 x 128 tiles, summed in K order
        if four:
            tiles = 256
        part = torch.zeros(tiles * 2 * 64 * 128, device="cuda", dtype=torch.float32)
        cnt = torch.zeros(tiles, device="cuda", dtype=torch.int32)
        plain = _gemm(A, W, variant=8)
        outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(100)]
        if force is True:
            set_option("gemm_splitk", 2)
        try:
            first = None
            for rnd in range(10):
                with torch.cuda.stream(side):
                    for _ in range(2):
                        thrash_dst.copy_(thrash_src, non_blocking=True)
                for o in outs:
                    o.fill_(float("nan"))
                for o in outs:
                    if four:
                        ok(lib().lt_op_gemm_splitk_auto(P(A), P(W), P(o), M, N, K, P(part), P(cnt), tiles, stream()), "gemm_splitk_auto")
                    else:
                        ok(lib().lt_op_gemm_splitk(P(A), P(W), P(o), M, N, K, P(part), P(cnt), tiles, stream()), "gemm_splitk")
                if first is None:
                    first = outs[0].clone()
                    assert not torch.isnan(first.float()).any() and rel_l2(first, plain) < 2e-3
                for o in outs:
                    bad += (o.view(torch.int16) != first.view(torch.int16)).any().int()
            torch.cuda.synchronize()
            assert int(cnt.abs().sum()) == 0
        finally:
            set_option("gemm_splitk", 1)
        assert int(bad.item()) == 0, (M, N, K, int(bad.item()))


@pytest.mark.parametrize("epilogue", [0, 1])
@pytest.mark.parametrize("K,N,ntile,gather", [(512, 4096, 44, True), (1536, 640, 9, True), (256, 2048, 40, False), (4096, 1536, 36, False)])
def test_gemm_grouped_persistent_kernel(K, N, ntile, gather, epilogue):
    """round 4: grouped (MoE expert) mode of the persistent 16x16x32 kernel (variant 15 with a tile -> expert table; the engine picks
    it for >= 2 tiles per CU, Next-DiT-MoE at 1024^2).  Holes in the table (padding segments anywhere, not only behind the last
    expert), several tiles per workgroup (44 x 16 = 704 tiles on 256 CUs: the LDS gather-map slots alternate and the DMA stream crosses
    tile boundaries with per-tile lane offsets), ragged N (640 = 2.5 tiles), the shortest K the mode accepts (256), gather-on-load with
    padding rows inside real tiles, every token twice.  Against the 8-wave ping-pong kernel on a gathered copy (same products, other
    accumulation order inside a 32-deep slab: equal to fp32 rounding) and an fp32 reference; padding segments stay untouched; the
    gathered and the copy form of the SAME kernel are bit-identical."""
    E = 4
    g = torch.Generator().manual_seed(K + N + ntile + epilogue)
    te = torch.randint(0, E, (ntile,), generator=g).tolist()
    for hole in (1, ntile // 2, ntile - 1):
        te[hole] = -1
    M = 256 * ntile
    T = M // 2 - 37
    X = bf(torch.randn(T, K, generator=g))
    W = bf(torch.randn(E, N, K, generator=g) / math.sqrt(K))
    row_map = torch.full((M,), -1, dtype=torch.int32)
    src = torch.cat([torch.randperm(T, generator=g), torch.randperm(T, generator=g)]).to(torch.int32)
    cursor = 0
    for t_, ex in enumerate(te):
        if ex < 0:
            continue
        cnt = 256 if t_ % 3 else 256 - 7 * (t_ % 11) - 1  # ragged fill: padding rows behind the entries of some real tiles
        cnt = min(cnt, src.numel() - cursor)
        row_map[256 * t_: 256 * t_ + cnt] = src[cursor:cursor + cnt]
        cursor += cnt
    valid = (row_map >= 0).cuda()
    gathered = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
    gathered[valid] = X[row_map[row_map >= 0].long().cuda()]
    tile_expert = torch.tensor(te, dtype=torch.int32, device="cuda")
    No = N // 2 if epilogue else N
    fill = lambda: torch.full((M, No), 3.0, device="cuda", dtype=torch.bfloat16)
    pp, copy, got = fill(), fill(), fill()
    ok(lib().lt_op_gemm_grouped(P(gathered), P(W), P(tile_expert), N * K, P(pp), M, N, K, epilogue, 3, stream()), "grouped pp")
    ok(lib().lt_op_gemm_grouped(P(gathered), P(W), P(tile_expert), N * K, P(copy), M, N, K, epilogue, 15, stream()), "grouped w4q")
    if gather:
        ok(lib().lt_op_gemm_grouped_gather(P(X), T, P(row_map.cuda()), P(W), P(tile_expert), N * K, P(got), M, N, K, epilogue, 15, stream()),
           "grouped_gather w4q")
    torch.cuda.synchronize()
    if gather:
        assert torch.equal(got, copy)
    for t_, ex in enumerate(te):
        rows = slice(256 * t_, 256 * t_ + 256)
        if ex < 0:
            assert torch.all(copy[rows] == 3.0), "padding segment was written"
            continue
        assert rel_l2(copy[rows], pp[rows]) < 4e-3, (t_, ex, rel_l2(copy[rows], pp[rows]))
        if t_ % 7 == 0:
            y = gathered[rows].float() @ W[ex].float().t()
            if epilogue:
                y = y.view(256, N // 64, 2, 32)
                y = r16(r16(F.silu(r16(y[:, :, 0]))) * r16(y[:, :, 1])).reshape(256, No)
            assert rel_l2(copy[rows], y) < 6e-3, (t_, ex, rel_l2(copy[rows], y))


@pytest.mark.parametrize("K,N,ntile,holes", [(4096, 1536, 68, 2), (4096, 1536, 64, 0), (1024, 768, 100, 3), (512, 256, 300, 5)])
def test_gemm_grouped_tail_split(K, N, ntile, holes):
    """round 6: the grouped persistent kernel cuts the tiles of a partial LAST round of its walk along K (2 / 4 parts,
    picked on the device from the number of valid tiles) and the last part of a tile to arrive sums the fp32 parts in K order.  Shapes: the
    MoE W2 launch at 1024^2 (66 / 64 valid row tiles x 6 = 1.55 / 1.5 rounds of 256 CUs: 4 / 2 parts), three column tiles with 97 valid row
    tiles (291 = 1.14 rounds: 35 tail tiles), one column tile and 295 row tiles.  Against the same kernel without the split (another
    summation order along K: equal to fp32 rounding), an fp32 reference, twice on one workspace (bit-identical: the order of arrival
    does not enter; the counters are back at zero) and with padding segments left untouched."""
    E = 4
    g = torch.Generator().manual_seed(K + N + ntile)
    te = torch.randint(0, E, (ntile,), generator=g).tolist()
    for h in range(holes):
        te[(h * 37 + 1) % ntile] = -1
    M = 256 * ntile
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(E, N, K, generator=g) / math.sqrt(K))
    tile_expert = torch.tensor(te, dtype=torch.int32, device="cuda")
    cap = 4 * 256
    ws = torch.full((cap, 256 * 256), float("nan"), device="cuda", dtype=torch.float32)
    cnt = torch.zeros(256, device="cuda", dtype=torch.int32)
    fill = lambda: torch.full((M, N), 3.0, device="cuda", dtype=torch.bfloat16)
    plain, split, again = fill(), fill(), fill()
    ok(lib().lt_op_gemm_grouped(P(A), P(W), P(tile_expert), N * K, P(plain), M, N, K, 0, 15, stream()), "grouped w4q")
    ok(lib().lt_op_gemm_grouped_tail(P(A), P(W), P(tile_expert), N * K, P(split), M, N, K, P(ws), P(cnt), cap, stream()), "grouped tail")
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    used = int(torch.isfinite(ws[:, 0]).sum())
    valid = sum(1 for x in te if x >= 0) * ((N + 255) // 256)
    tail = valid % 256 if valid > 256 else 0
    assert used in ((0,) if tail == 0 else (2 * tail, 4 * tail)), (used, tail)  # parts written = tail tiles x the split the device picked
    ok(lib().lt_op_gemm_grouped_tail(P(A), P(W), P(tile_expert), N * K, P(again), M, N, K, P(ws), P(cnt), cap, stream()), "grouped tail again")
    torch.cuda.synchronize()
    assert torch.equal(again, split) and int(cnt.abs().sum()) == 0
    for t_, ex in enumerate(te):
        rows = slice(256 * t_, 256 * t_ + 256)
        if ex < 0:
            assert torch.all(split[rows] == 3.0), "padding segment was written"
            continue
        assert rel_l2(split[rows], plain[rows]) < 3e-3, (t_, ex, rel_l2(split[rows], plain[rows]))
        if t_ % 9 == 0 or t_ >= ntile - 3:
            y = A[rows].float() @ W[ex].float().t()
            assert rel_l2(split[rows], y) < 4e-3, (t_, ex, rel_l2(split[rows], y))
    if tail:
        assert not torch.equal(split, plain)  # (the split really ran: another summation order along K)


@pytest.mark.parametrize("tokens,B,kvh,hd,K,variant", [(4096, 2, 32, 72, 2304, 0), (64, 3, 2, 72, 128, 1), (128, 2, 8, 72, 576, 2),
                                                       (4160, 2, 32, 96, 3072, 0), (1024, 2, 32, 48, 1536, 0), (64, 1, 4, 72, 64, 2)])
def test_gemm_vt_epilogue_matches_gemm_plus_transpose(tokens, B, kvh, hd, K, variant):
    """epilogue 2: the V projection written straight into the attention kernels' transposed, key-permuted V image
    [B, kv heads, hd, tokens] must equal (bit for bit: same products in the same order, one bf16 rounding) the plain GEMM
    followed by lt_op_v_transpose - on both tile shapes, ragged N (288-wide tiles over N = 2304 / 576 / 3072), a ragged last
    row tile (M = 8320), heads that straddle 32-column MFMA tiles (hd 72), and every sample boundary inside the launch."""
    M, N = B * tokens, kvh * hd
    g = torch.Generator().manual_seed(tokens + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    plain = _gemm(A, W, variant=1)
    want = torch.full((B, kvh, hd, tokens), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(plain), N, 0, P(want), B, tokens, tokens, kvh, hd, stream()))
    guard = 128
    buf = torch.full((B * kvh * hd * tokens + 2 * guard,), 7.0, device="cuda", dtype=torch.bfloat16)
    got = buf[guard:-guard].view(B, kvh, hd, tokens)
    ok(lib().lt_op_gemm_vt(P(A), P(W), P(got), M, N, K, tokens, hd, variant, stream()), "gemm_vt")
    torch.cuda.synchronize()
    assert torch.equal(got, want), rel_l2(got, want)
    assert torch.all(buf[:guard] == 7.0) and torch.all(buf[-guard:] == 7.0), "stray store outside the V^T image"


@pytest.mark.parametrize("tokens,B,H,Hkv,hd,K", [(4096, 2, 32, 32, 72, 2304), (4096, 2, 32, 8, 72, 2304), (1024, 8, 16, 16, 72, 1152), (16384, 1, 32, 8, 72, 256),
                                                 (4160, 2, 32, 32, 96, 3072), (320, 16, 16, 16, 96, 256), (320, 24, 16, 16, 72, 192),
                                                 (4096, 2, 32, 32, 48, 1536)])  # round 4: the 600M models at 1024^2 (256-wide tiles, 2.25 rounds)
def test_gemm_fused_qkv_matches_separate_launches(tokens, B, H, Hkv, hd, K):
    """one launch of the persistent 256 x 288 (or 256 x 256: Flag-DiT's 3072-wide Q, K, V) kernel for the whole QKV projection
    (lt_op_gemm_qkv: plain tiles for the Q | K columns, swapped-operand V^T tiles for the V columns) against the same kernel run as a
    plain GEMM + lt_op_v_transpose: identical MFMA sequences per output element -> bit-identical; MHA and GQA splits, several tiles
    per CU, short K; round 3: samples that end inside a 256-row tile (4160 = Flag-DiT's 64 x 65 tokens, 320) and a ragged last
    row tile (M = 8320)"""
    d, dkv = H * hd, Hkv * hd
    M, N, split = B * tokens, H * hd + 2 * Hkv * hd, H * hd + Hkv * hd
    g = torch.Generator().manual_seed(tokens + Hkv)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    plain = _gemm(A, W, variant=16)
    vt_ref = torch.empty(B, Hkv, hd, tokens, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(plain), N, split, P(vt_ref), B, tokens, tokens, Hkv, hd, stream()))
    C = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    vt = torch.full((B, Hkv, hd, tokens), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_qkv(P(A), P(W), P(C), P(vt), M, N, K, split, tokens, hd, stream()), "gemm_qkv")
    torch.cuda.synchronize()
    assert torch.equal(C[:, :split], plain[:, :split])
    assert torch.isnan(C[:, split:].float()).all()   # the V columns of C are not written
    assert torch.equal(vt, vt_ref), rel_l2(vt, vt_ref)


@pytest.mark.parametrize("variant", [0, 1, 3, 7])
@pytest.mark.parametrize("epilogue", [0, 1])
def test_gemm_grouped_expert_segments(variant, epilogue):
    """grouped (MoE) mode: every 256-row segment of the expert-sorted rows multiplies with ITS expert's weight
    (Next-DiT-MoE/models/models2.py:470-476 per-expert loop); padding segments (-1) must leave their output rows untouched."""
    E, K, N = 4, 192, 384
    te = [2, 0, -1, 3, 3, 1]
    M = 256 * len(te)
    g = torch.Generator().manual_seed(17 + variant + epilogue)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(E, N, K, generator=g) / math.sqrt(K))
    tile_expert = torch.tensor(te, dtype=torch.int32, device="cuda")
    No = N // 2 if epilogue else N
    out = torch.full((M, No), 3.0, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_grouped(P(A), P(W), P(tile_expert), N * K, P(out), M, N, K, epilogue, variant, stream()), "grouped")
    torch.cuda.synchronize()
    for t, ex in enumerate(te):
        rows = slice(256 * t, 256 * t + 256)
        if ex < 0:
            assert torch.all(out[rows] == 3.0), "padding segment was written"
            continue
        y = A[rows].float() @ W[ex].float().t()
        if epilogue:  # packed layout: 32-row groups alternate w1 / w3 (lt_op_pack_w13)
            y = y.view(256, N // 64, 2, 32)
            ref = r16(r16(F.silu(r16(y[:, :, 0]))) * r16(y[:, :, 1])).reshape(256, No)
        else:
            ref = y
        assert rel_l2(out[rows], ref) < 6e-3, (t, ex, rel_l2(out[rows], ref))


def test_rmsnorm_mod():
    B, N, d = 2, 70, 2304
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(B * N, d, generator=g) * 3)
    w = bf(1 + 0.1 * torch.randn(d, generator=g))
    ld = 3 * d
    mod = bf(torch.randn(B, ld, generator=g) * 0.5)
    out = torch.empty_like(x)
    ok(lib().lt_op_rmsnorm_mod(P(x), P(w), P(mod[:, d:]), None, ld, P(out), B, N, d, 1e-5, 0, stream()))
    torch.cuda.synchronize()
    xf = x.float().cpu()
    n = r16(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5))
    n = r16(n * w.float().cpu())
    sc = mod.float().cpu()[:, d:2 * d].repeat_interleave(N, dim=0)
    ref = r16(n * r16(1 + sc))
    assert rel_l2(out, ref) < 2e-3, rel_l2(out, ref)
    assert max_abs(out, ref) <= 0.07  # a 1-ulp flip of an intermediate bf16 at |x| ~ 8


@pytest.mark.parametrize("d", [576, 2304])
def test_rmsnorm_apex_order_option(d):
    """option rmsnorm_apex (include/lumina_dit_debug.h): the weight of the RMSNorms multiplies in fp32 BEFORE the one bf16 rounding
    (components.py:6-9 with apex, as SURVEY.md 8c states it) instead of after a rounding of its own (vanilla class, :11-54).  Both
    row kernels, against torch emulations of the two orders: the option's output must sit closer to its own emulation than to the
    other one's, and really differ from the default (d = 2304 has specialised instantiations: the option routes around them)."""
    B, N = 2, 40
    g = torch.Generator().manual_seed(d)
    x = bf(torch.randn(B * N, d, generator=g) * 3)
    y = bf(torch.randn(B * N, d, generator=g) * 2)
    w = bf(1 + 0.1 * torch.randn(d, generator=g))
    nw = bf(1 + 0.1 * torch.randn(d, generator=g))
    ld = 4 * d
    mod = bf(torch.randn(B, ld, generator=g) * 0.5)
    ok(lib().lt_op_prep_mod(P(mod), B, ld, 1, 4, d, 0b0010, 0b0100, -1, stream()))

    def run():
        out = torch.empty_like(x)
        ok(lib().lt_op_rmsnorm_mod(P(x), P(w), None, None, ld, P(out), B, N, d, 1e-5, 0, stream()))
        x2, h2 = x.clone(), torch.empty_like(x)
        ok(lib().lt_op_gated_residual_norm(P(x2), P(y), P(w), P(mod[:, d:]), 1, 0, P(nw), P(mod[:, 2 * d:]), None, 1, ld, P(h2), B, N, d,
                                           1e-5, 1e-6, 1, stream()))
        torch.cuda.synchronize()
        return out.float().cpu(), x2.float().cpu(), h2.float().cpu()

    base = run()
    set_option("rmsnorm_apex", 1)
    try:
        apex = run()
    finally:
        set_option("rmsnorm_apex", 0)
    assert torch.equal(run()[0], base[0])  # and back
    xf, yf, wf, nwf = x.float().cpu(), y.float().cpu(), w.float().cpu(), nw.float().cpu()
    rs = lambda v: torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)
    emu = {0: r16(r16(xf * rs(xf)) * wf), 1: r16(xf * rs(xf) * wf)}
    assert rel_l2(base[0], emu[0]) < rel_l2(base[0], emu[1]) and rel_l2(apex[0], emu[1]) < rel_l2(apex[0], emu[0])
    assert rel_l2(apex[0], emu[1]) < 1e-3 and not torch.equal(apex[0], base[0])
    m = mod.float().cpu()
    gate, scale1 = m[:, d:2 * d].repeat_interleave(N, dim=0), m[:, 2 * d:3 * d].repeat_interleave(N, dim=0)  # prepared: tanh(gate), 1 + scale
    for k, got in ((0, base), (1, apex)):
        yn = r16(r16(yf * rs(yf)) * wf) if k == 0 else r16(yf * rs(yf) * wf)
        xn = r16(xf + r16(gate * yn))
        hn = r16(r16(xn * rs(xn)) * nwf) if k == 0 else r16(xn * rs(xn) * nwf)
        assert rel_l2(got[1], xn) < 2e-3 and rel_l2(got[2], r16(hn * scale1)) < 3e-3, k
    assert not torch.equal(apex[2], base[2])


@pytest.mark.parametrize("next_mode", [0, 1, 2])
def test_gated_residual_norm(next_mode):
    B, N, d = 2, 33, 576
    g = torch.Generator().manual_seed(2 + next_mode)
    x = bf(torch.randn(B * N, d, generator=g))
    y = bf(torch.randn(B * N, d, generator=g) * 2)
    pw = bf(1 + 0.1 * torch.randn(d, generator=g))
    nw = bf(1 + 0.1 * torch.randn(d, generator=g))
    ld = 4 * d
    mod = bf(torch.randn(B, ld, generator=g))
    x_dev = x.clone()
    h = torch.full_like(x, float("nan"))
    ok(lib().lt_op_gated_residual_norm(P(x_dev), P(y), P(pw), P(mod[:, d:]), 1, 1, P(nw) if next_mode == 1 else None,
                                       P(mod[:, 2 * d:]) if next_mode else None, None, next_mode, ld, P(h), B, N, d,
                                       1e-5, 1e-6, 0, stream()))
    # engine path: gates / scales prepared once (tanh, 1 + scale) by prep_mod, row kernel with gate_mode 0 + scale_pre 1
    # -> must be bit identical to the in-kernel form
    mod2, x2, h2 = mod.clone(), x.clone(), torch.full_like(x, float("nan"))
    ok(lib().lt_op_prep_mod(P(mod2), B, ld, 1, 4, d, 0b0010, 0b0100, -1, stream()))
    ok(lib().lt_op_gated_residual_norm(P(x2), P(y), P(pw), P(mod2[:, d:]), 1, 0, P(nw) if next_mode == 1 else None,
                                       P(mod2[:, 2 * d:]) if next_mode else None, None, next_mode, ld, P(h2), B, N, d,
                                       1e-5, 1e-6, 1, stream()))
    torch.cuda.synchronize()
    assert torch.equal(x2, x_dev)
    if next_mode:
        assert torch.equal(h2, h)
    m = mod.float().cpu()
    gate = r16(torch.tanh(m[:, d:2 * d])).repeat_interleave(N, dim=0)
    scale = m[:, 2 * d:3 * d].repeat_interleave(N, dim=0)
    yf = y.float().cpu()
    yn = r16(r16(yf * torch.rsqrt(yf.pow(2).mean(-1, keepdim=True) + 1e-5)) * pw.float().cpu())
    xn = r16(x.float().cpu() + r16(gate * yn))
    assert rel_l2(x_dev, xn) < 2e-3
    if next_mode == 1:
        hn = r16(r16(xn * torch.rsqrt(xn.pow(2).mean(-1, keepdim=True) + 1e-5)) * nw.float().cpu())
        ref = r16(hn * r16(1 + scale))
        assert rel_l2(h, ref) < 3e-3, rel_l2(h, ref)
    elif next_mode == 2:
        ref = r16(F.layer_norm(xn, (d,), None, None, 1e-6) * r16(1 + scale))
        assert rel_l2(h, ref) < 3e-3, rel_l2(h, ref)


@pytest.mark.parametrize("d", [1536, 2304, 3072, 576])
@pytest.mark.parametrize("post_mode,next_mode", [(1, 1), (1, 2), (0, 1), (0, 2)])
def test_gated_residual_norm_specialised_is_bit_identical(d, post_mode, next_mode):
    """the engine's mode combinations on the compile-time-specialised instantiations (d = 576 has none: falls through to the
    generic kernel) against the generic kernel on the same inputs"""
    B, N = 2, 70
    g = torch.Generator().manual_seed(d + 10 * post_mode + next_mode)
    x = bf(torch.randn(B * N, d, generator=g))
    y = bf(torch.randn(B * N, d, generator=g) * 2)
    pw, nw = bf(1 + 0.1 * torch.randn(d, generator=g)), bf(1 + 0.1 * torch.randn(d, generator=g))
    ld = 4 * d
    mod = bf(torch.randn(B, ld, generator=g) * 0.3)
    outs = []
    for spec in (0, 1):
        xs, hs = x.clone(), torch.full_like(x, float("nan"))
        set_option("norm_specialize", spec)
        try:
            ok(lib().lt_op_gated_residual_norm(P(xs), P(y), P(pw) if post_mode else None, P(mod[:, d:]), post_mode, 0,
                                               P(nw) if next_mode == 1 else None, P(mod[:, 2 * d:]), P(mod[:, 3 * d:]), next_mode, ld, P(hs),
                                               B, N, d, 1e-5, 1e-6, 1, stream()))
        finally:
            set_option("norm_specialize", 1)  # the default
        torch.cuda.synchronize()
        outs.append((xs, hs))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][1].float()).all()


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("d,K,N", [(2304, 2304, 4096), (2304, 6144, 4096), (1536, 1536, 8192), (2304, 2304, 4099)])
def test_proj_gated_residual_norm_ystat(d, K, N, form):
    """round 6 (option grn_ystat): the O / W2 projection leaves the rows' sum-of-squares partials behind (GemmArgs::ystat) and the row kernel
    streams on them.  Against the same two launches without it (the row kernel reduces y itself: same statements, only the fp32 summation
    order of the statistic differs -> equal up to rare one-ulp flips) and against an fp32 restatement with the reference's rounding points
    (model.py:597-610, components.py:40-54)."""
    B = 3 if N % 2 else 2  # (an odd row count: the last wave of the two-rows-per-wave form holds one row)
    M = B * N
    g = torch.Generator().manual_seed(d + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(d, K, generator=g) * K ** -0.5)
    x = bf(torch.randn(M, d, generator=g))
    pw, nw = bf(1 + 0.1 * torch.randn(d, generator=g)), bf(1 + 0.1 * torch.randn(d, generator=g))
    ld = 3 * d
    mod = bf(torch.randn(B, ld, generator=g) * 0.3)
    cap = 2 * ((d + 255) // 256)
    outs = []
    for use in (0, 1):
        xs, hs, ys = x.clone(), torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        ws = torch.full((M, cap), float("nan"), device="cuda", dtype=torch.float32)
        set_option("grn_ystat", form)  # 1: one row per wave; 2: two rows per wave (odd row counts end on a half-filled wave)
        try:
            ok(lib().lt_op_proj_gated_residual_norm(P(A), P(W), P(ys), P(ws), cap, K, P(xs), P(pw), P(mod[:, :d]), P(nw), P(mod[:, d:]), ld, P(hs),
                                                    B, N, d, 1e-5, use, stream()))
        finally:
            set_option("grn_ystat", 1)
        torch.cuda.synchronize()
        outs.append((ys, xs, hs, ws))
    (y0, x0, h0, _), (y1, x1, h1, ws) = outs
    assert torch.equal(y0, y1)  # the GEMM's outputs do not change
    yf = y1.float()
    # the launch packs its ns partials per row densely, [M][ns] at the front of the workspace; ns = 2 x column tiles of the tile width the
    # dispatcher picked for this shape (288 or 256): the NaN fill says how many were written
    nfin = int(torch.isfinite(ws).sum())
    ns = nfin // M
    assert nfin == M * ns and ns in (2 * ((d + 287) // 288), 2 * ((d + 255) // 256)), (nfin, M, ns)
    slots = ws.flatten()[: M * ns].view(M, ns)
    assert torch.isfinite(slots).all()
    assert rel_l2(slots.sum(-1), yf.pow(2).sum(-1)) < 1e-5  # the partials are the row's sum of squares
    gate = mod[:, :d].float().repeat_interleave(N, dim=0)
    scale = mod[:, d:2 * d].float().repeat_interleave(N, dim=0)
    yn = r16(r16(yf * torch.rsqrt(yf.pow(2).mean(-1, keepdim=True) + 1e-5)) * pw.float())
    xn = r16(x.float() + r16(gate * yn))
    hn = r16(r16(r16(xn * torch.rsqrt(xn.pow(2).mean(-1, keepdim=True) + 1e-5)) * nw.float()) * scale)
    for xs_, hs_ in ((x0, h0), (x1, h1)):  # each form against the reference first: says WHICH one is off when they disagree
        assert rel_l2(xs_, xn) < 2e-3 and rel_l2(hs_, hn) < 3e-3, (rel_l2(xs_, xn), rel_l2(hs_, hn))
    assert rel_l2(x1, x0) < 2e-4 and rel_l2(h1, h0) < 2e-4
    assert (x1 != x0).float().mean() < 2e-3 and (h1 != h0).float().mean() < 2e-3


@pytest.mark.parametrize("heads,hd,qk_norm", [(8, 72, True), (2, 72, True), (32, 72, False), (32, 48, True)])
def test_qk_norm_rope(heads, hd, qk_norm):
    from oracle import nextdit_oracle as O
    B, Hp, Wp = 2, 6, 10
    N = Hp * Wp
    width = heads * hd
    g = torch.Generator().manual_seed(heads + hd)
    ld = width + 64
    src = bf(torch.randn(B * N, ld, generator=g))
    w = bf(1 + 0.1 * torch.randn(width, generator=g))
    b = bf(0.1 * torch.randn(width, generator=g))
    table = torch.empty(2, 384, hd // 4, 2, device="cuda", dtype=torch.float32)
    ok(lib().lt_op_rope_table_2d(P(table), 384, hd, 10000.0, 2.0, stream()))
    dst = torch.empty(B, heads, N, hd, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_qk_norm_rope(P(src), ld, 64, P(w) if qk_norm else None, P(b) if qk_norm else None, 1e-5, P(dst), B, N,
                                heads, hd, 1, P(table[1]), Wp, 1.0, stream()))
    # out_scale: applied in fp32 before the single bf16 rounding (how the engine folds softmax_scale * log2 e into K)
    dst_s = torch.empty_like(dst)
    ok(lib().lt_op_qk_norm_rope(P(src), ld, 64, P(w) if qk_norm else None, P(b) if qk_norm else None, 1e-5, P(dst_s), B, N,
                                heads, hd, 1, P(table[1]), Wp, 0.1875, stream()))
    torch.cuda.synchronize()
    xs = src.float().cpu()[:, 64:]
    if qk_norm:
        xs = F.layer_norm(xs, (width,), w.float().cpu(), b.float().cpu(), 1e-5)
    freqs = O.rope_table(hd, 384, scale_factor=2.0, scale_watershed=0.3, timestep=0.9)[:Hp, :Wp].flatten(0, 1).unsqueeze(0)
    ref = r16(O.apply_rotary(xs.view(B, N, heads, hd), freqs)).permute(0, 2, 1, 3)
    assert rel_l2(dst, ref) < 3e-3, rel_l2(dst, ref)
    ref_s = r16(O.apply_rotary(xs.view(B, N, heads, hd), freqs) * 0.1875).permute(0, 2, 1, 3)
    assert rel_l2(dst_s, ref_s) < 3e-3, rel_l2(dst_s, ref_s)
    # table itself: branch 0 = linear interpolation, branch 1 = NTK (model.py:944-952)
    lin = O.rope_table(hd, 384, scale_factor=2.0, scale_watershed=0.3, timestep=0.1)
    tb = table.cpu()
    got_row = torch.complex(tb[0, :, :, 0], tb[0, :, :, 1])  # [pos, freq]
    assert (got_row - lin[:, 0, 0::2]).abs().max() < 2e-4
    ntk = O.rope_table(hd, 384, scale_factor=2.0, scale_watershed=0.3, timestep=0.9)
    got_row = torch.complex(tb[1, :, :, 0], tb[1, :, :, 1])
    assert (got_row - ntk[0, :, 1::2]).abs().max() < 2e-4


@pytest.mark.parametrize("N,kvh,hd", [(64, 2, 72), (100, 8, 72), (256, 4, 48)])
def test_v_transpose_is_exact(N, kvh, hd):
    B = 2
    Npad = (N + 63) // 64 * 64
    ld = kvh * hd + 16
    src = bf(torch.randn(B * N, ld))
    dst = torch.full((B, kvh, hd, Npad), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(src), ld, 16, P(dst), B, N, Npad, kvh, hd, stream()))
    torch.cuda.synchronize()
    v = src.cpu()[:, 16:].view(B, N, kvh, hd).permute(0, 2, 3, 1)  # [B,kvh,hd,N]
    want = torch.zeros(B, kvh, hd, Npad, dtype=torch.bfloat16)
    idx = torch.arange(Npad)
    pos = (idx & ~12) | ((idx & 4) << 1) | ((idx & 8) >> 1)
    valid = idx < N
    want[..., pos[valid]] = v[..., idx[valid]]
    assert torch.equal(dst.cpu().view(torch.int16), want.view(torch.int16))


def _attn_ref(q, k, v, scale, bias=None):
    """exact softmax attention in fp32, one (batch, head) at a time to bound memory"""
    q, k, v = q.cuda().float(), k.cuda().float(), v.cuda().float()
    rep = q.shape[1] // k.shape[1]
    out = torch.empty_like(q)
    for b in range(q.shape[0]):
        for h in range(q.shape[1]):
            s = (q[b, h] @ k[b, h // rep].t()) * scale
            if bias is not None:
                s = s + bias[b].cuda()[None, :]
            out[b, h] = torch.softmax(s, -1) @ v[b, h // rep]
    return out.cpu()


def _run_attn(q, k, v, scale, bias=None, gate=None, prev=None, fold_scale=False):
    """fold_scale: hand the kernel K * scale * log2(e) (rounded once to bf16) with k_prescaled = 1 - the engine's path"""
    B, H, N, hd = q.shape
    if fold_scale:
        k = (k.float() * (scale * 1.4426950408889634)).to(torch.bfloat16)
    Hkv, Nk = k.shape[1], k.shape[2]
    Nkpad = (Nk + 63) // 64 * 64
    vsrc = v.permute(0, 2, 1, 3).reshape(B * Nk, Hkv * hd).contiguous()
    vt = torch.empty(B, Hkv, hd, Nkpad, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(vsrc), Hkv * hd, 0, P(vt), B, Nk, Nkpad, Hkv, hd, stream()))
    out = prev.clone() if prev is not None else torch.full((B, N, H * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
    bias_dev = None
    if bias is not None:
        bias_dev = torch.full((B, Nkpad), float("-inf"), device="cuda", dtype=torch.float32)
        bias_dev[:, :Nk] = bias.cuda()
    ok(lib().lt_op_attention(P(q), P(k), P(vt), P(bias_dev), P(out), P(gate), 1 if gate is not None else 0, B, H, Hkv, N,
                             Nk, Nkpad, hd, scale, 1 if fold_scale else 0, stream()), "attention")
    torch.cuda.synchronize()
    return out.view(B, N, H, hd).permute(0, 2, 1, 3)


@pytest.mark.parametrize("variant", [1, 2, 3, 4])  # 4: one wave per SIMD x 64 query rows (hd 72 / 96 / 48, whole tiles; else variant 3)
@pytest.mark.parametrize("B,H,Hkv,N,hd", [(1, 8, 8, 128, 72), (2, 8, 2, 320, 72), (2, 4, 4, 200, 72), (1, 3, 3, 64, 72),
                                          (2, 32, 32, 4096, 72), (1, 8, 8, 256, 48), (1, 8, 8, 192, 96),
                                          (1, 8, 8, 1000, 72), (1, 2, 2, 40, 72),
                                          # hd 96 (Flag-DiT 5B): variant 3 = the ping-pong kernel with VALU-side max / row sum and the
                                          # XOR-swizzled K image; 4160 = 64 rows x 65 tokens (cfg 3), ragged tails, GQA, one tile
                                          (2, 32, 32, 4160, 96), (2, 8, 2, 321, 96), (1, 4, 4, 1000, 96), (1, 3, 3, 64, 96), (1, 2, 2, 40, 96)])
@pytest.mark.parametrize("fold", [False, True])
def test_attention_self(variant, B, H, Hkv, N, hd, fold):
    set_option("attention_variant", variant)
    g = torch.Generator().manual_seed(N + hd)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    scale = math.sqrt(math.log(N, 64) / hd) if N > 64 else 1 / math.sqrt(hd)
    out = _run_attn(q, k, v, scale, fold_scale=fold)
    ref = _attn_ref(q.cpu(), k.cpu(), v.cpu(), scale)
    assert not torch.isnan(out.float()).any()
    assert rel_l2(out, ref) < 6e-3, rel_l2(out, ref)


@pytest.mark.parametrize("hd", [72, 96, 48])
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 6])
def test_attention_softmax_outlier_keys(variant, hd):
    """forces large running-max jumps mid-sequence, above and below the deferred-rescale threshold (guide 5.4
    rule 26); v1 (rescale every tile), v2 (threshold 8 in log2 units) and v3 (same threshold; hd 72: max folded into the
    QK^T MFMA as a bf16 pair, hd 96: explicit VALU max with O / l rescale) must all match the exact softmax"""
    set_option("attention_variant", variant)
    B, H, N = 1, 8, 256
    g = torch.Generator().manual_seed(9)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, H, N, hd, generator=g))
    v = bf(torch.randn(B, H, N, hd, generator=g))
    k[:, :, 131] = q[:, :, 7] * 4.0
    k[:, :, 3] = q[:, :, 200] * 2.0
    k[:, :, 70] = q[:, :, 100] * 0.35   # raises the row max by less than the threshold: stays un-rescaled
    k[:, :, :64] -= q[:, :, 50:51] * 3.0  # first tile far BELOW later ones for row 50 (and shifted for all rows)
    out = _run_attn(q, k, v, 1 / math.sqrt(hd))
    ref = _attn_ref(q.cpu(), k.cpu(), v.cpu(), 1 / math.sqrt(hd))
    assert rel_l2(out, ref) < 6e-3, rel_l2(out, ref)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("T,valid1", [(16, 8), (13, 5), (128, 8), (77, 77), (200, 130)])
@pytest.mark.parametrize("fold", [False, True])
def test_attention_text_accumulate(variant, T, valid1, fold):
    set_option("attention_variant", variant)
    B, H, Hkv, N, hd = 2, 8, 2, 96, 72
    g = torch.Generator().manual_seed(T)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, T, hd, generator=g))
    v = bf(torch.randn(B, Hkv, T, hd, generator=g))
    gate = bf(torch.randn(H, generator=g))
    prev = bf(torch.randn(B, N, H * hd, generator=g))
    mask = torch.ones(B, T)
    mask[1, valid1:] = 0
    bias = torch.where(mask > 0, 0.0, float("-inf"))
    out = _run_attn(q, k, v, 1 / math.sqrt(hd), bias=bias, gate=gate, prev=prev, fold_scale=fold)
    oy = r16(_attn_ref(q.cpu(), k.cpu(), v.cpu(), 1 / math.sqrt(hd), bias))
    gt = r16(torch.tanh(gate.float().cpu())).view(1, H, 1, 1)
    ref = r16(prev.float().cpu().view(B, N, H, hd).permute(0, 2, 1, 3) + r16(oy * gt))
    assert rel_l2(out, ref) < 4e-3, rel_l2(out, ref)


@pytest.mark.parametrize("hd", [72, 96])
@pytest.mark.parametrize("B,H,Hkv,N,T,valid1", [(2, 8, 8, 320, 128, 8), (2, 8, 2, 200, 77, 30), (1, 4, 4, 4096, 256, 256),
                                                  (2, 4, 4, 96, 300, 130), (2, 4, 2, 512, 128, 100)])
@pytest.mark.parametrize("variant", [3, 4])
def test_attention_fused_text(B, H, Hkv, N, T, valid1, hd, variant):
    """one launch = self-attention + gated text cross-attention (model.py:392-434), both K pre-scaled (engine path)"""
    set_option("attention_variant", variant)
    g = torch.Generator().manual_seed(N + T)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    tk = bf(torch.randn(B, Hkv, T, hd, generator=g))
    tv = bf(torch.randn(B, Hkv, T, hd, generator=g))
    gate = bf(torch.randn(H, generator=g))
    mask = torch.ones(B, T)
    mask[B - 1, valid1:] = 0
    bias = torch.where(mask > 0, 0.0, float("-inf"))
    s_self = math.sqrt(math.log(N, 64) / hd) if N > 64 else 1 / math.sqrt(hd)
    s_txt = 1 / math.sqrt(hd)
    L2E = 1.4426950408889634
    kf = (k.float() * (s_self * L2E)).to(torch.bfloat16)
    tkf = (tk.float() * (s_txt * L2E)).to(torch.bfloat16)
    Npad, Tpad = (N + 63) // 64 * 64, (T + 63) // 64 * 64
    vt = torch.empty(B, Hkv, hd, Npad, device="cuda", dtype=torch.bfloat16)
    tvt = torch.empty(B, Hkv, hd, Tpad, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(v.permute(0, 2, 1, 3).reshape(B * N, Hkv * hd).contiguous()), Hkv * hd, 0, P(vt), B, N, Npad, Hkv, hd, stream()))
    ok(lib().lt_op_v_transpose(P(tv.permute(0, 2, 1, 3).reshape(B * T, Hkv * hd).contiguous()), Hkv * hd, 0, P(tvt), B, T, Tpad, Hkv, hd, stream()))
    bias_dev = torch.full((B, Tpad), float("-inf"), device="cuda", dtype=torch.float32)
    bias_dev[:, :T] = bias.cuda()
    out = torch.full((B, N, H * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_attention_fused(P(q), P(kf), P(vt), P(tkf), P(tvt), P(bias_dev), P(gate), P(out), B, H, Hkv, N, N, Npad, T, Tpad,
                                   hd, stream()), "attention_fused")
    torch.cuda.synchronize()
    got = out.view(B, N, H, hd).permute(0, 2, 1, 3)
    o_self = r16(_attn_ref(q.cpu(), k.cpu(), v.cpu(), s_self))
    o_txt = r16(_attn_ref(q.cpu(), tk.cpu(), tv.cpu(), s_txt, bias))
    gt = r16(torch.tanh(gate.float().cpu())).view(1, H, 1, 1)
    ref = r16(o_self + r16(o_txt * gt))
    assert not torch.isnan(got.float()).any()
    assert rel_l2(got, ref) < 6e-3, rel_l2(got, ref)


@pytest.mark.parametrize("B,H,Hkv,N,outliers", [(2, 32, 32, 4096, False), (1, 8, 2, 1024, True), (1, 4, 4, 64, False), (1, 2, 2, 200 * 64, True),
                                                 # tile counts 2, 3, 5, 6, 7 (mod 4 remainders of the ring-depth unrolled loop), ragged last q-block
                                                 (1, 4, 4, 128, False), (1, 4, 2, 192, True), (1, 4, 4, 320, True), (2, 2, 2, 384, False), (1, 2, 1, 448, True)])
def test_attention_v4_is_bit_identical_to_v3(B, H, Hkv, N, outliers):
    """attn_fwd_kernel_v4 (4 waves x 64 query rows, asm-owned AGPRs) runs the same arithmetic in the same order as the ping-pong
    kernel: per query row the two are the same sequence of MFMAs, exp2 and max-moves (a wave-wide `any` only decides WHEN the rare
    branch runs, its per-row effect is the identity for rows that do not raise) -> outputs must be equal bit for bit, with max
    moves late in the sequence included"""
    hd = 72
    g = torch.Generator().manual_seed(N + H)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    if outliers:
        rep = H // Hkv
        k[:, :, N // 2 + 3] = q[:, ::rep, 7] * 4.0
        k[:, :, 5] = q[:, ::rep, N - 9] * 2.0
        k[:, :, N - 64:] += q[:, ::rep, 40:41] * 1.5   # last tile above everything before it for row 40
        k[:, :, :64] -= q[:, ::rep, 50:51] * 3.0
    scale = math.sqrt(math.log(N, 64) / hd) if N > 64 else 1 / math.sqrt(hd)
    outs = []
    for variant in (3, 4):
        set_option("attention_variant", variant)
        outs.append(_run_attn(q, k, v, scale, fold_scale=True).clone())
    assert torch.equal(outs[0], outs[1]), rel_l2(outs[1], outs[0])
    ref = _attn_ref(q.cpu(), k.cpu(), v.cpu(), scale)
    assert rel_l2(outs[1], ref) < 6e-3


@pytest.mark.parametrize("B,H,Hkv,N,outliers", [(2, 32, 32, 4096, False), (1, 8, 2, 1024, True), (1, 4, 4, 64, False), (1, 2, 2, 200 * 64, True), (2, 32, 32, 256, False),
                                                 # tile counts 2, 3, 5, 6, 7 (mod 4 remainders of the ring-depth unrolled loop), ragged last q-block
                                                 (1, 4, 4, 128, False), (1, 4, 2, 192, True), (1, 4, 4, 320, True), (2, 2, 2, 384, False), (1, 2, 1, 448, True)])
@pytest.mark.parametrize("fold", [True, False])
def test_attention_v4_hd48_against_the_v2_kernel_and_fp32(B, H, Hkv, N, outliers, fold):
    """attn_fwd_kernel_v4h48 (round 4: the one-wave-per-SIMD structure at head_dim 48 - a fourth, pad-only k-step carries the folded
    maximum, 64-row O^T with the row of ones, six-gap overhang of the softmax window into the block's own PV segment) against the
    round-1 kernel (variant 2: fp32 maximum / row sum on the VALU - other rounding on purpose, as between the two hd-96 kernels) and
    the exact softmax; the 600M models' shapes (32 heads, 256 and 4096 tokens), max moves late in the sequence, GQA, every remainder
    of the unrolled tile loop, scale folded into K or applied to Q."""
    hd = 48
    g = torch.Generator().manual_seed(N + H + hd)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    if outliers:
        rep = H // Hkv
        k[:, :, N // 2 + 3] = q[:, ::rep, 7] * 4.0
        k[:, :, 5] = q[:, ::rep, N - 9] * 2.0
        k[:, :, N - 64:] += q[:, ::rep, 40:41] * 1.5   # last tile above everything before it for row 40
        k[:, :, :64] -= q[:, ::rep, 50:51] * 3.0
    scale = math.sqrt(math.log(N, 64) / hd) if N > 64 else 1 / math.sqrt(hd)
    outs = []
    for variant in (2, 6):  # 6 = 4 with the hd-48 kernel forced (the dispatcher keeps the round-1 kernel below ~200 workgroups)
        set_option("attention_variant", variant)
        outs.append(_run_attn(q, k, v, scale, fold_scale=fold).clone())
    ref = _attn_ref(q.cpu(), k.cpu(), v.cpu(), scale)
    assert not torch.isnan(outs[1].float()).any()
    assert rel_l2(outs[1], ref) < 6e-3, rel_l2(outs[1], ref)
    # (measured: v4 3.0e-3, v2 1.8 - 2.4e-3 from the fp32 softmax, 4.1e-3 from each other: the folded maximum is a bf16 pair and
    #  the row sum adds the bf16-rounded P the PV MFMA multiplies - the hd-72 / hd-96 one-wave kernels do the same)
    assert rel_l2(outs[1], outs[0]) < 5e-3, rel_l2(outs[1], outs[0])



@pytest.mark.parametrize("B,H,Hkv,N,outliers", [(2, 32, 32, 4160, False), (1, 8, 2, 1024, True), (1, 4, 4, 64, False), (1, 2, 2, 100 * 64, True),
                                                 # tile counts 2, 3, 5, 6, 7 (mod 4 remainders of the ring-depth unrolled loop), ragged last q-block
                                                 (1, 4, 4, 128, False), (1, 4, 2, 192, True), (1, 4, 4, 320, True), (2, 2, 2, 384, False), (1, 2, 1, 448, True)])
def test_attention_v4_hd96_against_the_ping_pong_kernel_and_fp32(B, H, Hkv, N, outliers):
    """attn_fwd_kernel_v4h96 (round 3: the one-wave-per-SIMD structure at head_dim 96 - running maximum through a seventh, constant-K
    k-step, row sum by v_dot2 on the packed P, XOR-swizzled 192-byte K rows) against attn_fwd_kernel_v3<96> and the exact softmax.
    The two kernels round differently on purpose (v3<96>: fp32 maximum and fp32 row sum on the VALU; v4: bf16-pair folded maximum and
    the sum of the bf16-rounded P the PV MFMA multiplies - as both hd-72 kernels do), so they agree to a few bf16 ulps of the
    output, not bit for bit; max moves late in the sequence, GQA and every remainder of the unrolled tile loop included."""
    hd = 96
    g = torch.Generator().manual_seed(N + H + 96)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    if outliers:
        rep = H // Hkv
        k[:, :, N // 2 + 3] = q[:, ::rep, 7] * 4.0
        k[:, :, 5] = q[:, ::rep, N - 9] * 2.0
        k[:, :, N - 64:] += q[:, ::rep, 40:41] * 1.5   # last tile above everything before it for row 40
        k[:, :, :64] -= q[:, ::rep, 50:51] * 3.0
    scale = math.sqrt(math.log(N, 64) / hd) if N > 64 else 1 / math.sqrt(hd)
    outs = {}
    for variant in (3, 4):
        set_option("attention_variant", variant)
        for fold in (True, False):
            outs[variant, fold] = _run_attn(q, k, v, scale, fold_scale=fold).clone()
    ref = _attn_ref(q.cpu(), k.cpu(), v.cpu(), scale)
    for key, o in outs.items():
        assert not torch.isnan(o.float()).any(), key
        assert rel_l2(o, ref) < 6e-3, (key, rel_l2(o, ref))
    assert rel_l2(outs[4, True], outs[3, True]) < 3e-3, rel_l2(outs[4, True], outs[3, True])
    assert max_abs(outs[4, True], ref) < 0.06 * float(ref.abs().max()) + 0.02


@pytest.mark.parametrize("B,H,Hkv,N,T,valid1", [(2, 8, 8, 320, 128, 8), (2, 8, 2, 192, 77, 30), (1, 4, 4, 4096, 256, 256), (2, 4, 2, 512, 40, 33),
                                                  (2, 4, 4, 64, 200, 130)])
def test_attention_v4_fused_text_is_bit_identical_to_v3(B, H, Hkv, N, T, valid1):
    """the text phase of variant 4 runs the text keys through the SAME tile pipeline as the image keys (the additive mask rides in
    a pad slot of the QK^T MFMA: 1.0 x {0, -inf} per key) - same values as the ping-pong kernel's `scores + mask`, bit for bit"""
    hd = 72
    g = torch.Generator().manual_seed(N + T + 1)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    tk = bf(torch.randn(B, Hkv, T, hd, generator=g))
    tv = bf(torch.randn(B, Hkv, T, hd, generator=g))
    gate = bf(torch.randn(H, generator=g))
    mask = torch.ones(B, T)
    mask[B - 1, valid1:] = 0
    bias = torch.where(mask > 0, 0.0, float("-inf"))
    L2E = 1.4426950408889634
    kf = (k.float() * (L2E / math.sqrt(hd))).to(torch.bfloat16)
    tkf = (tk.float() * (L2E / math.sqrt(hd))).to(torch.bfloat16)
    Npad, Tpad = (N + 63) // 64 * 64, (T + 63) // 64 * 64
    vt = torch.empty(B, Hkv, hd, Npad, device="cuda", dtype=torch.bfloat16)
    tvt = torch.empty(B, Hkv, hd, Tpad, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(v.permute(0, 2, 1, 3).reshape(B * N, Hkv * hd).contiguous()), Hkv * hd, 0, P(vt), B, N, Npad, Hkv, hd, stream()))
    ok(lib().lt_op_v_transpose(P(tv.permute(0, 2, 1, 3).reshape(B * T, Hkv * hd).contiguous()), Hkv * hd, 0, P(tvt), B, T, Tpad, Hkv, hd, stream()))
    bias_dev = torch.full((B, Tpad), float("-inf"), device="cuda", dtype=torch.float32)
    bias_dev[:, :T] = bias.cuda()
    outs = []
    for variant in (3, 4):
        set_option("attention_variant", variant)
        out = torch.full((B, N, H * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
        ok(lib().lt_op_attention_fused(P(q), P(kf), P(vt), P(tkf), P(tvt), P(bias_dev), P(gate), P(out), B, H, Hkv, N, N, Npad, T, Tpad,
                                       hd, stream()), "attention_fused")
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[1].float()).any()
    assert torch.equal(outs[0], outs[1]), rel_l2(outs[1], outs[0])


def _to_pair_ref(m):
    """row-pair-interleaved image of a [rows][cols] matrix (include/lumina_dit.h): (r, k) -> (r >> 1) * 2 cols + (k >> 5) * 64 + (r & 1) * 32 + (k & 31)"""
    rows, cols = m.shape
    return m.view(rows // 2, 2, cols // 32, 32).permute(0, 2, 1, 3).reshape(rows, cols).contiguous()


@pytest.mark.parametrize("rows,cols", [(2, 32), (6, 2304), (256, 6144), (12288, 2304), (8320, 3072), (64, 16384)])
def test_pair_layout_in_place_conversion(rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    m = bf(torch.randn(rows, cols, generator=g))
    buf = torch.full((rows * cols + 64,), 7.0, device="cuda", dtype=torch.bfloat16)
    w = buf[:rows * cols].view(rows, cols)
    w.copy_(m)
    ok(lib().lt_op_pair_layout(P(w), rows, cols, 1, stream()))
    torch.cuda.synchronize()
    assert torch.equal(w, _to_pair_ref(m))
    ok(lib().lt_op_pair_layout(P(w), rows, cols, 0, stream()))
    torch.cuda.synchronize()
    assert torch.equal(w, m) and torch.all(buf[rows * cols:] == 7.0)


@pytest.mark.parametrize("M,N,K,epi", [(8192, 2304, 2304, 0), (8192, 2304, 6144, 0), (8192, 12288, 2304, 1), (8192, 6912, 2304, 0), (8320, 3072, 3072, 0),
                                       (8320, 16384, 3072, 1), (8190, 2304, 2304, 0), (16384, 12288, 2304, 1)])
def test_gemm_pair_layout_is_bit_identical(M, N, K, epi):
    """round 6: the persistent GEMM reading A and W in the row-pair-interleaved layout (whole 128-byte lines per LDS-DMA request) multiplies
    the same slabs in the same order as on row-major operands - outputs equal bit for bit; with the SwiGLU epilogue the output can be
    written in the pair layout too (it is the W2 projection's A operand)"""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    No = N // 2 if epi else N
    ref = torch.full((M, No), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_bf16(P(A), P(W), P(None), 1, P(ref), M, N, K, epi, 0, stream()), "gemm")
    Ap, Wp = A.clone(), W.clone()
    ok(lib().lt_op_pair_layout(P(Ap), M, K, 1, stream()))
    ok(lib().lt_op_pair_layout(P(Wp), N, K, 1, stream()))
    for pair_c in ((0, 1) if epi else (0,)):
        buf = torch.full((M * No + 64,), float("nan"), device="cuda", dtype=torch.bfloat16)
        out = buf[:M * No].view(M, No)
        buf[M * No:] = 7.0
        ok(lib().lt_op_gemm_bf16_pair(P(Ap), P(Wp), P(out), M, N, K, epi, pair_c, stream()), "gemm pair")
        if pair_c:
            ok(lib().lt_op_pair_layout(P(out), M, No, 0, stream()))
        torch.cuda.synchronize()
        assert torch.all(buf[M * No:] == 7.0)
        assert not torch.isnan(out.float()).any()
        assert torch.equal(out, ref), (pair_c, rel_l2(out, ref))


def _fused_text_inputs(B, H, Hkv, N, T, valid, hd, seed):
    """inputs of lt_op_attention_fused as the engine hands them over (both K pre-scaled); valid[b] = valid text keys of sample b"""
    g = torch.Generator().manual_seed(seed)
    q = bf(torch.randn(B, H, N, hd, generator=g))
    k = bf(torch.randn(B, Hkv, N, hd, generator=g))
    v = bf(torch.randn(B, Hkv, N, hd, generator=g))
    tk = bf(torch.randn(B, Hkv, T, hd, generator=g))
    tv = bf(torch.randn(B, Hkv, T, hd, generator=g))
    gate = bf(torch.randn(H, generator=g))
    mask = torch.ones(B, T)
    for b_, n_ in enumerate(valid):
        mask[b_, n_:] = 0
    bias = torch.where(mask > 0, 0.0, float("-inf"))
    s_self = math.sqrt(math.log(N, 64) / hd) if N > 64 else 1 / math.sqrt(hd)
    s_txt = 1 / math.sqrt(hd)
    L2E = 1.4426950408889634
    kf = (k.float() * (s_self * L2E)).to(torch.bfloat16)
    tkf = (tk.float() * (s_txt * L2E)).to(torch.bfloat16)
    Npad, Tpad = (N + 63) // 64 * 64, (T + 63) // 64 * 64
    vt = torch.empty(B, Hkv, hd, Npad, device="cuda", dtype=torch.bfloat16)
    tvt = torch.empty(B, Hkv, hd, Tpad, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_v_transpose(P(v.permute(0, 2, 1, 3).reshape(B * N, Hkv * hd).contiguous()), Hkv * hd, 0, P(vt), B, N, Npad, Hkv, hd, stream()))
    ok(lib().lt_op_v_transpose(P(tv.permute(0, 2, 1, 3).reshape(B * T, Hkv * hd).contiguous()), Hkv * hd, 0, P(tvt), B, T, Tpad, Hkv, hd, stream()))
    bias_dev = torch.full((B, Tpad), float("-inf"), device="cuda", dtype=torch.float32)
    bias_dev[:, :T] = bias.cuda()

    def run():
        out = torch.full((B, N, H * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
        ok(lib().lt_op_attention_fused(P(q), P(kf), P(vt), P(tkf), P(tvt), P(bias_dev), P(gate), P(out), B, H, Hkv, N, N, Npad, T, Tpad,
                                       hd, stream()), "attention_fused")
        torch.cuda.synchronize()
        return out

    def ref():
        o_self = r16(_attn_ref(q.cpu(), k.cpu(), v.cpu(), s_self))
        o_txt = r16(_attn_ref(q.cpu(), tk.cpu(), tv.cpu(), s_txt, bias))
        gt = r16(torch.tanh(gate.float().cpu())).view(1, H, 1, 1)
        return r16(o_self + r16(o_txt * gt)).permute(0, 2, 1, 3).reshape(B, N, H * hd)

    return run, ref


@pytest.mark.parametrize("hd", [72, 96])
@pytest.mark.parametrize("B,H,Hkv,N,T,valid", [(2, 8, 8, 320, 128, (128, 8)), (2, 4, 2, 512, 256, (70, 8)), (2, 4, 4, 256, 200, (60, 130)),
                                                (1, 4, 4, 1024, 256, (1,)), (2, 4, 4, 192, 128, (128, 64))])
def test_attention_text_tile_skip_is_bit_identical(B, H, Hkv, N, T, valid, hd):
    """round 6, option attn_text_skip: the one-wave kernels do not run the text tiles behind a sample's last valid key (the unconditional
    half of a CFG pair: 8 valid keys of 128).  Skipped keys are all masked -> exp2(-inf) = 0 into the row sum and O^T, the maximum
    untouched: the outputs must be equal bit for bit (and right: checked against fp32)"""
    set_option("attention_variant", 4)
    run, ref = _fused_text_inputs(B, H, Hkv, N, T, valid, hd, seed=N + T + hd)
    outs = []
    try:
        for skip in (0, 1):
            set_option("attn_text_skip", skip)
            outs.append(run())
    finally:
        set_option("attn_text_skip", 1)
    assert not torch.isnan(outs[1].float()).any()
    assert torch.equal(outs[0], outs[1]), rel_l2(outs[1], outs[0])
    assert rel_l2(outs[1], ref()) < 6e-3


@pytest.mark.parametrize("B,H,Hkv,N,T,valid,parts", [(2, 8, 8, 4160, 128, (128, 8), 4), (1, 8, 2, 2176, 77, (77,), 4), (1, 4, 4, 1088, 64, (64,), 2),
                                                      (2, 4, 4, 2112, 256, (256, 130), 3), (1, 16, 16, 4160, 0, (), 4), (1, 4, 1, 2688, 0, (), 4)])
def test_attention_hd96_tail_split(B, H, Hkv, N, T, valid, parts):
    """round 6, option attn_tail_split: the partial last query block of a head_dim-96 head (64 or 128 rows: Flag-DiT's
    4160 tokens) runs as `parts` workgroups over disjoint key ranges + a merge launch.  Rows of the whole blocks: bit-identical to the
    unsplit launch.  Tail rows: a different summation order of the same flash recurrence -> compared at the attention tolerance against
    fp32, and against the unsplit kernel at a bf16 ulp's worth"""
    hd = 96
    set_option("attention_variant", 4)
    if T:
        run, ref = _fused_text_inputs(B, H, Hkv, N, T, valid, hd, seed=N + T)
    else:
        g = torch.Generator().manual_seed(N)
        q = bf(torch.randn(B, H, N, hd, generator=g))
        k = bf(torch.randn(B, Hkv, N, hd, generator=g))
        v = bf(torch.randn(B, Hkv, N, hd, generator=g))
        rep = H // Hkv
        k[:, :, N - 70] = q[:, ::rep, N - 5] * 3.0   # a late maximum for a tail row: the parts' maxima differ by a lot
        k[:, :, 3] = q[:, ::rep, N - 40] * 2.0
        scale = math.sqrt(math.log(N, 64) / hd)
        run = lambda: _run_attn(q, k, v, scale, fold_scale=True).permute(0, 2, 1, 3).reshape(B, N, H * hd).clone()
        ref = lambda: _attn_ref(q.cpu(), k.cpu(), v.cpu(), scale).permute(0, 2, 1, 3).reshape(B, N, H * hd)
    outs = {}
    try:
        for ts in (0, parts):
            set_option("attn_tail_split", ts)
            outs[ts] = run()
    finally:
        set_option("attn_tail_split", 4)
    whole = (N // 256) * 256
    assert N % 256 in (64, 128)
    assert not torch.isnan(outs[parts].float()).any()
    assert torch.equal(outs[0][:, :whole], outs[parts][:, :whole])
    r = ref()
    assert rel_l2(outs[parts], r) < 6e-3, rel_l2(outs[parts], r)
    assert rel_l2(outs[parts][:, whole:], r[:, whole:]) < 6e-3, rel_l2(outs[parts][:, whole:], r[:, whole:])
    assert rel_l2(outs[parts][:, whole:], outs[0][:, whole:]) < 4e-3, rel_l2(outs[parts][:, whole:], outs[0][:, whole:])
    assert not torch.equal(outs[parts][:, whole:], torch.zeros_like(outs[parts][:, whole:]))


@pytest.mark.parametrize("M,N,K,act", [(2, 1000, 256, 0), (2, 9216, 1024, 1), (1, 37, 128, 1), (8, 64, 2048, 0), (3, 1001, 1024, 1), (5, 7, 64, 0),
                                       (2, 101376, 1024, 1)])
def test_linear_small_m(M, N, K, act):
    """eight output columns per wave since round 3: N not a multiple of 8 / 32, fewer columns than one wave's eight, every register
    height (2 / 4 / 8 rows), the adaLN GEMV of cfg 1 (101 376 columns); a guarded buffer catches stores past [M, N]"""
    g = torch.Generator().manual_seed(N)
    a = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = bf(torch.randn(N, generator=g))
    buf = torch.full((M * N + 64,), 7.0, device="cuda", dtype=torch.bfloat16)
    y = buf[:M * N].view(M, N)
    ok(lib().lt_op_linear_small_m(P(a), P(w), P(b), P(y), M, N, K, act, stream()))
    torch.cuda.synchronize()
    assert torch.all(buf[M * N:] == 7.0)
    af = a.float().cpu()
    if act:
        af = r16(F.silu(af))
    ref = af @ w.float().cpu().t() + b.float().cpu()
    assert rel_l2(y, ref) < 4e-3


@pytest.mark.parametrize("B,tokens,H,Hkv,grid_w", [(2, 4096, 32, 32, 64), (2, 4096, 32, 8, 128)])
def test_qkv_qstat_then_attention_qraw_equal_the_separate_passes(B, tokens, H, Hkv, grid_w):
    """round 4, the attn_q_fused path of the engine at the op level (model.py:355-371 + :392-405), head_dim 72:
    (1) lt_op_qkv_qstat: the fused QKV launch leaves C / V^T bit-identical to lt_op_gemm_qkv, its K pass equals lt_op_qk_norm_rope on the K
        columns bit for bit, and the per-row (mean, rstd) reduced from the epilogue's partial sums equal torch's statistics of the
        bf16-rounded Q columns to fp32 accuracy (the variance comes from E[x^2] - mean^2: a few 1e-6 relative);
    (2) lt_op_attention_qraw with exactly those statistics equals lt_op_attention on the queries lt_op_qk_norm_rope makes, up to the
        bf16 ulps that the two forms of the row variance move (two-pass there, sums here): <= 2e-3 of the output norm; fed the two-pass
        statistics instead, the prologue is qk_norm_rope's arithmetic and the outputs agree to <= 1e-3."""
    hd = 72
    d, dkv = H * hd, Hkv * hd
    M, Nall, K = B * tokens, d + 2 * dkv, d
    if not lib().lt_op_gemm_qkv_fusable(M, Nall, K, d + dkv, tokens, hd):
        pytest.skip("this box does not take the fused QKV launch for the shape")
    g = torch.Generator().manual_seed(H * 7 + Hkv)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(Nall, K, generator=g) / math.sqrt(K))
    W[:d] += bf(0.02 * torch.ones(1, K))  # a row mean that is not negligible beside the spread: exercises E[x^2] - mean^2
    qw, qb = bf(1 + 0.1 * torch.randn(d, generator=g)), bf(0.1 * torch.randn(d, generator=g))
    kw, kb = bf(1 + 0.1 * torch.randn(dkv, generator=g)), bf(0.1 * torch.randn(dkv, generator=g))
    table = torch.empty(2, 384, hd // 4, 2, device="cuda", dtype=torch.float32)
    table_t = torch.empty(2, hd // 4, 384, 2, device="cuda", dtype=torch.float32)
    ok(lib().lt_op_rope_table_2d_pair(P(table), P(table_t), 384, hd, 10000.0, 1.0, stream()))
    ref_table = torch.empty_like(table)
    ok(lib().lt_op_rope_table_2d(P(ref_table), 384, hd, 10000.0, 1.0, stream()))
    torch.cuda.synchronize()
    assert torch.equal(table, ref_table) and torch.equal(table_t, table.permute(0, 2, 1, 3).contiguous())
    scale = math.sqrt(math.log(tokens, 4096) / hd) if tokens > 4096 else 1 / math.sqrt(hd)
    kscale = scale * 1.4426950408889634
    # the separate passes
    C0 = torch.empty(M, Nall, device="cuda", dtype=torch.bfloat16)
    vt0 = torch.empty(B, Hkv, hd, tokens, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_gemm_qkv(P(A), P(W), P(C0), P(vt0), M, Nall, K, d + dkv, tokens, hd, stream()))
    q0 = torch.empty(B, H, tokens, hd, device="cuda", dtype=torch.bfloat16)
    k0 = torch.empty(B, Hkv, tokens, hd, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_qk_norm_rope(P(C0), Nall, 0, P(qw), P(qb), 1e-5, P(q0), B, tokens, H, hd, 1, P(table[1]), grid_w, 1.0, stream()))
    ok(lib().lt_op_qk_norm_rope(P(C0), Nall, d, P(kw), P(kb), 1e-5, P(k0), B, tokens, Hkv, hd, 1, P(table[1]), grid_w, kscale, stream()))
    out0 = torch.full((B, tokens, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_attention(P(q0), P(k0), P(vt0), None, P(out0), None, 0, B, H, Hkv, tokens, tokens, tokens, hd, scale, 1, stream()))
    # the fused path
    C1, vt1, k1 = torch.empty_like(C0), torch.empty_like(vt0), torch.empty_like(k0)
    ws = torch.empty(M, 32, 2, device="cuda", dtype=torch.float32)
    qmr = torch.empty(M, 2, device="cuda", dtype=torch.float32)
    ok(lib().lt_op_qkv_qstat(P(A), P(W), P(C1), P(vt1), M, Nall, K, d + dkv, tokens, hd, d, P(kw), P(kb), P(table[1]), grid_w, kscale,
                             P(k1), P(ws), P(qmr), stream()), "qkv_qstat")
    out1 = torch.full_like(out0, float("nan"))
    ok(lib().lt_op_attention_qraw(P(C1), Nall, 0, P(qmr), P(qw), P(qb), P(table), P(table_t), 384, grid_w, P(k1), P(vt1), P(out1),
                                  B, H, Hkv, tokens, tokens, hd, stream()), "attention_qraw")
    torch.cuda.synchronize()
    assert torch.equal(C1[:, :d + dkv], C0[:, :d + dkv]) and torch.equal(vt1, vt0)  # (the V columns of C are not written: V leaves as vt)
    if not torch.equal(k1, k0):
        bad = (k1.view(torch.int16) != k0.view(torch.int16))
        idx = bad.nonzero()
        raise AssertionError(f"k differs: rel {rel_l2(k1, k0):.3e}, {int(bad.sum())} of {bad.numel()} elements; first {idx[:6].tolist()} last {idx[-3:].tolist()}; "
                             f"rows hit {sorted(set(idx[:, 2].tolist()))[:20]}")
    x = C0[:, :d].float()
    mean, var = x.mean(-1), x.var(-1, unbiased=False)
    assert (mean.abs() > 0.2 * var.sqrt()).float().mean() > 0.5  # (the draw really has rows with a sizeable mean)
    assert ((qmr[:, 0] - mean).abs() / var.sqrt()).max() < 1e-5
    assert (qmr[:, 1] * torch.sqrt(var + 1e-5) - 1).abs().max() < 2e-5
    assert not torch.isnan(out1.float()).any()
    assert rel_l2(out1, out0) < 2e-3, rel_l2(out1, out0)
    # and with the statistics of the separate pass's form (two-pass variance, computed here) the prologue is the same arithmetic:
    # whatever still differs is fp32 contraction inside two differently compiled kernels, far below a bf16 ulp of q on average
    qmr2 = torch.stack([mean, torch.rsqrt(var + 1e-5)], -1).contiguous()
    out2 = torch.full_like(out0, float("nan"))
    ok(lib().lt_op_attention_qraw(P(C1), Nall, 0, P(qmr2), P(qw), P(qb), P(table), P(table_t), 384, grid_w, P(k1), P(vt1), P(out2),
                                  B, H, Hkv, tokens, tokens, hd, stream()), "attention_qraw (two-pass statistics)")
    torch.cuda.synchronize()
    assert rel_l2(out2, out0) < 1e-3, rel_l2(out2, out0)


@pytest.mark.parametrize("B,tokens,H,Hkv,K,grid_w", [(2, 256, 32, 32, 1536, 16),   # the 600M ImageNet / MoE models at 256^2: 512 rows, 128 x 128 tiles
                                                      (2, 64, 8, 8, 384, 8),        # the tiny test models: 64 x 128 tiles, one key tile
                                                      (1, 512, 16, 8, 768, 32),     # the longest sequence the kernel keeps resident, GQA
                                                      (3, 192, 16, 8, 384, 12)])    # three key tiles, odd batch x heads still a multiple of 8, two query heads per kv head
def test_qkv_rowstat_then_fused_small_attention_equals_the_separate_passes(B, tokens, H, Hkv, K, grid_w):
    """round 5, the attn_small_fused path of the engine at the op level (Next-DiT-ImageNet/models/models.py:358-404), head_dim 48:
    (1) the small-M QKV GEMM with GemmArgs::rowstat writes C bit-identically to the plain launch and leaves, per row and 128-column
        tile, (sum, sum of squares) of the bf16-rounded outputs - checked against torch on every tile;
    (2) ONE kernel then does q_norm, k_norm, RoPE, scale fold, V transpose and attention: against the three separate passes
        (lt_op_qk_norm_rope x 2, lt_op_v_transpose, lt_op_attention) up to the bf16 ulps the two forms of the row variance move, and
        against an fp32 softmax attention on torch-made q / k."""
    from oracle import nextdit_oracle as O
    hd = 48
    d, dkv = H * hd, Hkv * hd
    M, Nall = B * tokens, d + 2 * dkv
    g = torch.Generator().manual_seed(tokens * 3 + H + Hkv)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(Nall, K, generator=g) / math.sqrt(K))
    W[:d] += bf(0.02 * torch.ones(1, K))  # a row mean that is not negligible beside the spread: exercises E[x^2] - mean^2
    qw, qb = bf(1 + 0.1 * torch.randn(d, generator=g)), bf(0.1 * torch.randn(d, generator=g))
    kw, kb = bf(1 + 0.1 * torch.randn(dkv, generator=g)), bf(0.1 * torch.randn(dkv, generator=g))
    table = torch.empty(2, 384, hd // 4, 2, device="cuda", dtype=torch.float32)
    ok(lib().lt_op_rope_table_2d(P(table), 384, hd, 10000.0, 1.0, stream()))
    scale = 1 / math.sqrt(hd)
    kscale = scale * 1.4426950408889634
    # the separate passes
    C0 = _gemm(A, W)
    q0 = torch.empty(B, H, tokens, hd, device="cuda", dtype=torch.bfloat16)
    k0 = torch.empty(B, Hkv, tokens, hd, device="cuda", dtype=torch.bfloat16)
    vt0 = torch.empty(B, Hkv, hd, tokens, device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_qk_norm_rope(P(C0), Nall, 0, P(qw), P(qb), 1e-5, P(q0), B, tokens, H, hd, 1, P(table[1]), grid_w, 1.0, stream()))
    ok(lib().lt_op_qk_norm_rope(P(C0), Nall, d, P(kw), P(kb), 1e-5, P(k0), B, tokens, Hkv, hd, 1, P(table[1]), grid_w, kscale, stream()))
    ok(lib().lt_op_v_transpose(P(C0), Nall, d + dkv, P(vt0), B, tokens, tokens, Hkv, hd, stream()))
    out0 = torch.full((B, tokens, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib().lt_op_attention(P(q0), P(k0), P(vt0), None, P(out0), None, 0, B, H, Hkv, tokens, tokens, tokens, hd, scale, 1, stream()))
    # the fused path
    slots = (Nall + 127) // 128
    C1 = torch.full((M, Nall), float("nan"), device="cuda", dtype=torch.bfloat16)
    ws = torch.full((M, slots, 2), float("nan"), device="cuda", dtype=torch.float32)
    out1 = torch.full_like(out0, float("nan"))
    ok(lib().lt_op_qkv_attention_small(P(A), P(W), P(C1), M, K, H, Hkv, tokens, hd, P(qw), P(qb), P(kw), P(kb), P(table), 384, grid_w, kscale,
                                       P(ws), P(out1), stream()), "qkv_attention_small")
    torch.cuda.synchronize()
    assert torch.equal(C1, C0)
    x = C0.float().view(M, -1)
    pad = slots * 128 - Nall
    xt = F.pad(x, (0, pad)).view(M, slots, 128)
    assert not torch.isnan(ws).any()
    assert (ws[..., 0] - xt.sum(-1)).abs().max() < 2e-3 * xt.abs().sum(-1).max()
    assert ((ws[..., 1] - xt.pow(2).sum(-1)).abs() / xt.pow(2).sum(-1).clamp_min(1e-3)).max() < 1e-5
    assert not torch.isnan(out1.float()).any()
    assert rel_l2(out1, out0) < 3e-3, rel_l2(out1, out0)
    # fp32 reference: torch LayerNorm + the oracle's rotary on the bf16 projection, one bf16 rounding of q and k, exact softmax
    xc = C0.float().cpu()
    Hp = tokens // grid_w
    freqs = O.rope_table(hd, 384)[:Hp, :grid_w].flatten(0, 1).unsqueeze(0)
    qr = r16(O.apply_rotary(F.layer_norm(xc[:, :d], (d,), qw.float().cpu(), qb.float().cpu(), 1e-5).view(B, tokens, H, hd), freqs)).permute(0, 2, 1, 3)
    kr = r16(O.apply_rotary(F.layer_norm(xc[:, d:d + dkv], (dkv,), kw.float().cpu(), kb.float().cpu(), 1e-5).view(B, tokens, Hkv, hd), freqs)).permute(0, 2, 1, 3)
    vr = xc[:, d + dkv:].view(B, tokens, Hkv, hd).permute(0, 2, 1, 3)
    ref = _attn_ref(qr, kr, vr, scale).permute(0, 2, 1, 3).reshape(B, tokens, d)
    assert rel_l2(out1, ref) < 8e-3, rel_l2(out1, ref)
    assert rel_l2(out0, ref) < 8e-3
