"""Per-kernel numerics on the MI355X: every C-ABI entry point against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
from pathlib import Path
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    from vggt_qwen3_amd import ops as _ops
    from vggt_qwen3_amd import _lib
    _lib.load()
    return _ops


def _rand(shape, scale=1.0, dtype=BF16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=F32) * scale).to(dtype).cuda()


def _relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _maxerr(a, b):
    return (a.float() - b.float()).abs().max().item()


# ------------------------------------------------------------------------------------------ GEMM
def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric B catches swapped row/col maps and permuted k order."""
    K = 128
    A = torch.eye(K, dtype=F32).to(BF16).cuda()
    Bm = (torch.arange(200 * K, dtype=F32).reshape(200, K) % 251 - 125).to(BF16).cuda()  # exact small ints
    out = ops.linear(A, Bm, out_dtype=F32)  # [K, 200] = I @ B^T
    assert torch.equal(out, Bm.float().t().contiguous()), f"max err {_maxerr(out, Bm.float().t())}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1200, 2560, 4096), (1200, 6144, 2560), (77, 200, 192),
                                   (24, 1000, 2560), (6174, 1024, 1024), (300, 151937 // 16, 256)])
def test_gemm_shapes(ops, M, N, K):
    A = _rand((M, K), 1.0, seed=1)
    W = _rand((N, K), 1.0, seed=2)
    ref = A.float() @ W.float().t()
    out32 = ops.linear(A, W, out_dtype=F32)
    assert _relerr(out32, ref) < 2e-6 * math.sqrt(K) + 1e-6, f"f32-out rel err {_relerr(out32, ref)}"
    out16 = ops.linear(A, W)
    assert _relerr(out16, ref) < 4e-3, f"bf16-out rel err {_relerr(out16, ref)}"


def test_gemm_epilogues(ops):
    M, N, K = 300, 520, 128
    A = _rand((M, K), 0.5, seed=3); W = _rand((N, K), 0.5, seed=4)
    bias = _rand((N,), 1.0, F32, seed=5); cs = _rand((N,), 1.0, F32, seed=6)
    R16 = _rand((M, N), 1.0, seed=7); R32 = R16.float()
    base = A.float() @ W.float().t()
    # f32 out: bias + gelu + colscale + residual, alpha
    out = ops.linear(A, W, bias=bias, colscale=cs, residual=R32, act=ops.ACT_GELU, out_dtype=F32, alpha=0.5)
    ref = torch.nn.functional.gelu(base * 0.5 + bias) * cs + R32
    assert _relerr(out, ref) < 1e-5, _relerr(out, ref)
    # bf16 out with the PyTorch rounding points
    out = ops.linear(A, W, bias=bias, residual=R16, act=ops.ACT_SILU)
    t = (base + bias).to(BF16)
    t = torch.nn.functional.silu(t.float()).to(BF16)
    ref = (t.float() + R16.float()).to(BF16)
    assert _maxerr(out, ref) <= 0.0625 and _relerr(out, ref) < 3e-3, (_maxerr(out, ref), _relerr(out, ref))
    # accumulate
    C = _rand((M, N), 1.0, seed=8)
    C0 = C.clone()
    ops.linear(A, W, out=C, accumulate=True)
    assert _relerr(C, base + C0.float()) < 4e-3
    C = _rand((M, N), 1.0, F32, seed=9); C0 = C.clone()
    ops.linear(A, W, out=C, accumulate=True)
    assert _relerr(C, base + C0) < 1e-5


@pytest.mark.parametrize("cfg", [20, 21, 22, 24, 25, 11, 13, 7, 9, 15, 16, 17])
def test_gemm_forced_configs_with_epilogues(ops, cfg):
    """Every tile configuration behind vq3_gemm_bf16_nt (20 = the 256x256 8-phase kernel; 11 / 13 / 7 / 9 = loader-wave and
    2-stage kernels) through the whole epilogue surface - bias, GELU, LayerScale, residual, accumulate, f32 output - with M and N
    edges inside a tile, an odd number of K tiles, strided C and batches: all must agree with the automatic choice's contract."""
    try:
        ops.gemm_force_config(cfg)
        shapes = ((300, 520, 192), (1029, 1024, 1024), (257, 264, 64))
        if cfg in (21, 22, 24, 25):
            # more tiles than CUs: the two-phase kernels are persistent (a workgroup walks several tiles and requests the next tile's
            # first K tiles during the current epilogue) - 65 x 8 / 129 x 4 tiles, ragged last row tile, 1 / 3 / 4 K tiles per tile
            shapes += ((16500, 1024, 256), (16500, 1000, 64), (33000, 520, 192))
        for (M, N, K) in shapes:
            A = _rand((M, K), 0.5, seed=3); W = _rand((N, K), 0.5, seed=4)
            bias = _rand((N,), 1.0, F32, seed=5); cs = _rand((N,), 1.0, F32, seed=6)
            R16 = _rand((M, N), 1.0, seed=7)
            base = A.float() @ W.float().t()
            out = ops.linear(A, W)
            assert _relerr(out, base) < 4e-3, (cfg, M, N, K, _relerr(out, base))
            out = ops.linear(A, W, bias=bias, colscale=cs, residual=R16, act=ops.ACT_GELU, alpha=0.5)
            t = (base * 0.5 + bias).to(BF16)
            t = torch.nn.functional.gelu(t.float()).to(BF16)
            t = (t.float() * cs).to(BF16)
            ref = (t.float() + R16.float()).to(BF16)
            # 2 bf16 ulps at the largest magnitude: a 1-ulp flip of the GEMM result can survive the two later roundings
            assert _relerr(out, ref) < 4e-3 and _maxerr(out, ref) <= 2 ** -6 * ref.float().abs().max().item(), \
                (cfg, M, N, K, _relerr(out, ref), _maxerr(out, ref))
            C = _rand((M, N), 1.0, seed=8); C0 = C.clone()
            ops.linear(A, W, out=C, accumulate=True)
            assert _relerr(C, base + C0.float()) < 4e-3
            out32 = ops.linear(A, W, bias=bias, out_dtype=F32)
            assert _relerr(out32, base + bias) < 1e-5
            # strided output (ldc > N) with a guard band that must stay untouched, in-place residual (R is C)
            big = torch.full((M, N + 24), 7.0, device="cuda", dtype=BF16)
            view = big[:, 8:8 + N]
            view.copy_(R16)
            ops.gemm_raw(A, W, view, M, N, K, K, K, N + 24, R=view, ldr=N + 24, c_off=0, r_off=0)
            assert _relerr(view, (base.to(BF16).float() + R16.float())) < 4e-3
            assert bool((big[:, :8] == 7).all()) and bool((big[:, 8 + N:] == 7).all())
        # batched: 3 x [200, 256] . [256, 128]^T
        A = _rand((3, 200, 128), 1.0, seed=20); W = _rand((3, 256, 128), 1.0, seed=21)
        C = torch.empty((3, 200, 256), device="cuda", dtype=BF16)
        ops.gemm_raw(A, W, C, 200, 256, 128, 128, 128, 256, nb1=3, sA=(200 * 128, 0), sB=(256 * 128, 0), sC=(200 * 256, 0))
        assert _relerr(C, torch.einsum("bmk,bnk->bmn", A.float(), W.float())) < 4e-3
    finally:
        ops.gemm_force_config(-3)


@pytest.mark.parametrize("M,N,K", [(9600, 2560, 4096), (1200, 2560, 2560), (2000, 2560, 1088), (5000, 2560, 1024), (9600, 2560, 9728)])
def test_gemm_last_round_split(ops, M, N, K):
    """cfg 25 (gemm6.hip): the tiles of a last round that is at most half full are cut along K over the idle CUs - 9600 x 2560 is 380 tiles of
    256 x 256 = one round of 256 + 124 tiles x 2 halves; 1200 x 2560 is 50 tiles x 4 slices; K = 1088 is 17 K tiles (9 + 8); 5000 x 2560 is 200
    tiles (no split: the plain launch). Partial sums meet in f32, so the result equals the unsplit kernel's up to the order of the f32 sum;
    the whole epilogue surface behind it; the bounded wait never gives up."""
    full, rem, sl = ops.gemm_split_plan(M, N, K)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    assert (sl == 0) == (M == 5000) and (sl == 0 or (full + rem == tiles and rem * sl <= 256 and K // 64 // sl >= 8))
    A = _rand((M, K), 0.5, seed=3); W = _rand((N, K), 0.5, seed=4)
    bias = _rand((N,), 1.0, F32, seed=5); cs = _rand((N,), 1.0, F32, seed=6)
    R16 = _rand((M, N), 1.0, seed=7)
    try:
        ops.gemm_force_config(20)
        ref_plain = ops.linear(A, W)
        ref_epi = ops.linear(A, W, bias=bias, colscale=cs, residual=R16, act=ops.ACT_GELU, alpha=0.5)
        ops.gemm_force_config(25)
        for _ in range(3):                                     # (the arrival counts are zeroed per launch: repeatable)
            out = ops.linear(A, W)
            assert _relerr(out, ref_plain) < 2e-3 and _maxerr(out, ref_plain) <= 2 ** -7 * ref_plain.float().abs().max().item()
        base = A.float() @ W.float().t()
        assert _relerr(out, base) < 4e-3
        out = ops.linear(A, W, bias=bias, colscale=cs, residual=R16, act=ops.ACT_GELU, alpha=0.5)
        assert _relerr(out, ref_epi) < 2e-3
        C = _rand((M, N), 1.0, seed=8); C0 = C.clone()
        ops.linear(A, W, out=C, accumulate=True)
        assert _relerr(C, base + C0.float()) < 4e-3
        view = R16.clone()                                     # in-place residual (R is C), as the transformer blocks call it
        ops.gemm_raw(A, W, view, M, N, K, K, K, N, R=view, ldr=N, c_off=0, r_off=0)
        assert _relerr(view, (base.to(BF16).float() + R16.float())) < 4e-3
        assert not ops.gemm_split_gave_up()
    finally:
        ops.gemm_force_config(-3)


def test_gemm_split_error_word_is_sticky(ops):
    """a split reducer whose bounded wait expires must leave a signal that SURVIVES later launches. The wait bound is
    shrunk to 1 poll (vq3_gemm_split_debug_spin_bound; on a healthy device the partial tiles of a 76-K-tile slice cannot have arrived by
    then), so the give-up path really runs: the process-wide error word - host-mapped, outside the per-launch memset of the arrival counts -
    is set, stays set across healthy split launches on this and on another stream, is visible to a host poll WITHOUT synchronisation once
    the launch completed, and clears only when read with clear. Stage1Trainer.check_kernels raises on it."""
    M, N, K = 1200, 2560, 9728
    assert ops.gemm_split_plan(M, N, K)[2] >= 2
    A = _rand((M, K), 0.5, seed=3); W = _rand((N, K), 0.5, seed=4)
    try:
        ops.gemm_force_config(25)
        good = ops.linear(A, W)
        assert not ops.gemm_split_gave_up()
        ops.gemm_split_debug_spin_bound(1)
        ops.linear(A, W)                                       # reducers give up at once (the result of this launch is incomplete)
        torch.cuda.synchronize()
        ops.gemm_split_debug_spin_bound(0)
        assert ops.gemm_split_poll(clear=False)                # host read, no synchronisation needed any more
        again = ops.linear(A, W)                               # a healthy launch: zeroes its arrival counts, NOT the error word
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ops.linear(A, W)                                   # another stream's workspace: same word
        torch.cuda.synchronize()
        assert torch.equal(again, good)
        assert ops.gemm_split_poll(clear=False)                # still there
        from vggt_qwen3_amd.trainer import Stage1Trainer
        with pytest.raises(RuntimeError, match="split-K"):
            Stage1Trainer.check_kernels(type("T", (), {"micro": 7})())
        assert not ops.gemm_split_poll(clear=False) and not ops.gemm_split_gave_up()      # read with clear: gone
    finally:
        ops.gemm_split_debug_spin_bound(0)
        ops.gemm_force_config(-3)


def test_gemm_whole_rounds_plus_row_tail(ops):
    """cfg 30 (gemm.hip: launch_split_rows): the 256 x 256 kernel on the row tiles that fill whole rounds of the chip + a second launch
    for the remaining rows - here 65 x 4 = 260 tiles = one round of 256 CUs + 4: rows 0..16383 and a 116-row tail. Same contract as
    every other configuration, epilogue operands offset with the rows."""
    try:
        ops.gemm_force_config(30)
        M, N, K = 16500, 1024, 256
        A = _rand((M, K), 0.5, seed=3); W = _rand((N, K), 0.5, seed=4)
        bias = _rand((N,), 1.0, F32, seed=5); cs = _rand((N,), 1.0, F32, seed=6)
        R16 = _rand((M, N), 1.0, seed=7)
        base = A.float() @ W.float().t()
        out = ops.linear(A, W)
        assert _relerr(out, base) < 4e-3
        assert _relerr(out[-116:], base[-116:]) < 4e-3 and _relerr(out[16300:16400], base[16300:16400]) < 4e-3
        out = ops.linear(A, W, bias=bias, colscale=cs, residual=R16, act=ops.ACT_GELU, alpha=0.5)
        t = (base * 0.5 + bias).to(BF16)
        t = torch.nn.functional.gelu(t.float()).to(BF16)
        t = (t.float() * cs).to(BF16)
        ref = (t.float() + R16.float()).to(BF16)
        assert _relerr(out, ref) < 4e-3 and _relerr(out[-116:], ref[-116:]) < 4e-3
        C = _rand((M, N), 1.0, seed=8); C0 = C.clone()
        ops.linear(A, W, out=C, accumulate=True)
        assert _relerr(C, base + C0.float()) < 4e-3 and _relerr(C[-116:], (base + C0.float())[-116:]) < 4e-3
        out32 = ops.linear(A, W, bias=bias, out_dtype=F32)
        assert _relerr(out32, base + bias) < 1e-5
        # a shape the split does not apply to falls back to the plain 256 x 256 launch
        A2 = _rand((300, 192), 0.5, seed=9); W2 = _rand((520, 192), 0.5, seed=10)
        assert _relerr(ops.linear(A2, W2), A2.float() @ W2.float().t()) < 4e-3
    finally:
        ops.gemm_force_config(-3)


@pytest.mark.parametrize("cfg", [-3, 20, 21, 22, 24, 25, 11, 13, 7, 30])
def test_gemm_layernorm_fold(ops, cfg):
    """LayerNorm folded into the GEMMs either side of it (vq3_gemm_bf16_nt_ln): the producer (residual GEMM) leaves per-row
    (sum, sum of squares) over 128-column groups of what it stored, the consumer reads the RAW rows and applies
    rstd (x.B^T - mu colsum) + bias in its epilogue. Against: torch LayerNorm (f32 statistics of the bf16 rows) -> bf16 -> GEMM. Every tile
    configuration; M with a ragged last tile; GELU on the consumer; statistics also from vq3_rowstats128 and bit-identical twice."""
    try:
        ops.gemm_force_config(cfg)
        # (cfg 25: 65 x 4 = 260 tiles of 256 x 256 = a round of 256 + 4 tiles cut into two K halves - the reducer runs the folded epilogue)
        M, C, N2 = (16500 if cfg in (30, 21, 22, 24, 25) else 1029), 1024, (1024 if cfg in (21, 22, 24, 25) else 512)
        if cfg == 25:
            assert ops.gemm_split_plan(M, N2, C)[2] == 2
        eps = 1e-5
        h = _rand((M, 256), 0.5, seed=1); Wp = _rand((C, 256), 0.3, seed=2)
        R = _rand((M, C), 1.0, seed=3) + 0.7                      # non-zero row means
        gamma = (torch.rand(C, device="cuda") + 0.5); beta = torch.randn(C, device="cuda") * 0.2
        W = _rand((N2, C), 0.05, seed=4).float(); b = torch.randn(N2, device="cuda") * 0.1
        # producer: x = R + h.Wp^T, statistics on the way out
        st = torch.full((M, C // 128, 2), float("nan"), device="cuda", dtype=F32)
        x = ops.linear(h, Wp, residual=R, ln_fold=ops.ln_fold(stats_out=st))
        xf = x.float()
        assert torch.equal(x, ops.linear(h, Wp, residual=R))      # the statistics do not change the product
        assert torch.allclose(st[:, :, 0], xf.view(M, C // 128, 128).sum(-1), rtol=1e-5, atol=1e-3)
        assert torch.allclose(st[:, :, 1], (xf * xf).view(M, C // 128, 128).sum(-1), rtol=1e-5, atol=1e-3)
        st2 = ops.rowstats128(x)
        assert torch.allclose(st2, st, rtol=1e-5, atol=1e-3)
        st_again = torch.empty_like(st)
        ops.linear(h, Wp, residual=R, ln_fold=ops.ln_fold(stats_out=st_again))
        assert torch.equal(st, st_again)
        # consumer
        Bw = (W * gamma[None, :]).to(BF16)
        colsum = Bw.float().sum(1).contiguous()
        dvec = (b + W @ beta).contiguous()
        y = ops.linear(x, Bw, bias=dvec, act=ops.ACT_GELU, ln_fold=ops.ln_fold(stats_in=st, eps=eps, colsum=colsum))
        xn = torch.nn.functional.layer_norm(xf, (C,), gamma, beta, eps)
        ref = torch.nn.functional.gelu((xn @ W.t() + b).to(BF16).float())
        assert _relerr(y, ref) < 6e-3, (cfg, _relerr(y, ref))
        # the unfused form on the same kernels (LayerNorm output rounded to bf16, as the reference does) agrees as closely
        y0 = ops.linear(xn.to(BF16), W.to(BF16), bias=b, act=ops.ACT_GELU)
        assert _relerr(y, y0) < 8e-3
    finally:
        ops.gemm_force_config(-3)


def test_gemm_layernorm_fold_adversarial_rows(ops):
    """the folded LayerNorm takes the variance as sum(x^2)/C - mu^2 (f32, one pass) and its output as the difference of two
    large terms, rstd (x.B^T - mu colsum). Rows a real residual stream has and random initialisation does not: a large common offset
    (mean 50, std 0.5), massive-activation channels (a few columns at +-1e3) and both together - against LayerNorm in f64 statistics ->
    bf16 -> GEMM. Tolerance: 2e-2 relative (the offset rows keep only ~3 significant bits of their deviation in bf16 - for both paths)."""
    M, C, N2, eps = 2048, 1024, 512, 1e-5
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(M, C, device="cuda", generator=g) * 0.5
    x[: M // 4] += 50.0                                           # common offset
    x[M // 4: M // 2, [3, 517, 900]] = torch.tensor([1e3, -1e3, 7e2], device="cuda")    # massive activations
    x[M // 2: 3 * M // 4] += 50.0
    x[M // 2: 3 * M // 4, [3, 517, 900]] = torch.tensor([1e3, -1e3, 7e2], device="cuda")
    x = x.to(BF16)
    gamma = (torch.rand(C, device="cuda", generator=g) + 0.5); beta = torch.randn(C, device="cuda", generator=g) * 0.2
    W = _rand((N2, C), 0.05, seed=4).float(); b = torch.randn(N2, device="cuda", generator=g) * 0.1
    st = ops.rowstats128(x)
    Bw = (W * gamma[None, :]).to(BF16)
    colsum = Bw.float().sum(1).contiguous()
    dvec = (b + W @ beta).contiguous()
    y = ops.linear(x, Bw, bias=dvec, ln_fold=ops.ln_fold(stats_in=st, eps=eps, colsum=colsum))
    xd = x.double()
    mu = xd.mean(1, keepdim=True); var = xd.var(1, unbiased=False, keepdim=True)
    xn = ((xd - mu) / torch.sqrt(var + eps) * gamma.double() + beta.double())
    ref = (xn.float().to(BF16).float() @ W.to(BF16).float().t() + b)
    for lo, hi, what in ((0, M // 4, "offset"), (M // 4, M // 2, "outliers"), (M // 2, 3 * M // 4, "both"), (3 * M // 4, M, "plain")):
        e = _relerr(y[lo:hi], ref[lo:hi])
        assert e < 2e-2, (what, e)


def test_colsum_multi_tall_slabs(ops):
    """vq3_colsum_multi on the slabs a merged pass of 8 micro-batches leaves (9600 x 128 from the q/k-prep backward, 480 x 2560 from
    RMSNorm): rows split over several workgroups, combined by the last arriver in split order - equal to the f32 column sums,
    bit-identical when repeated (tickets reset themselves), accumulate honoured; small slabs keep the one-workgroup path."""
    torch.manual_seed(0)
    for shapes in (((9600, 128), (9600, 128), (480, 2560), (480, 2560)), ((300, 2560), (1200, 128)), ((20000, 64),)):
        parts = [torch.randn(r, c, device="cuda") for r, c in shapes]
        outs = [torch.randn(c, device="cuda").to(BF16) for _, c in shapes]
        old = [o.clone() for o in outs]
        acc = [i % 2 == 1 for i in range(len(shapes))]
        ops.colsum_flush([(p, p.shape[0], p.shape[1], o, a) for p, o, a in zip(parts, outs, acc)])
        for p, o, o0, a in zip(parts, outs, old, acc):
            ref = p.double().sum(0) + (o0.double() if a else 0.0)
            assert _relerr(o, ref.float()) < 4e-3, (p.shape, _relerr(o, ref.float()))
        outs2 = [o0.clone() for o0 in old]
        ops.colsum_flush([(p, p.shape[0], p.shape[1], o, a) for p, o, a in zip(parts, outs2, acc)])
        for o, o2 in zip(outs, outs2):
            assert torch.equal(o, o2)


@pytest.mark.parametrize("sched", [102, 103, 105, 106])
def test_gemm_kmajor_layouts_all_schedules(ops, sched):
    """The any-layout kernel (gemm3.hip) in each of its schedules - 128x128 two-stage, 128x128 loader ring, 256x128 loader ring -
    for every operand layout (NT, k-major B = dgrad, k-major A and B = wgrad), K not a multiple of 64 (tail zero-fill), M / N edges
    inside tiles, accumulate into bf16 (the wgrad form) and a residual."""
    try:
        ops.gemm_force_config(sched)
        # (106 = both operands k-major on the 256 x 256 8-phase kernel, gemm6.hip: the other layouts and f32 output fall back to 105 under it;
        # its own shapes add a weight-gradient-sized product with a ragged last K tile and one without)
        shapes = ((1200, 520, 200), (304, 1032, 1208), (2560, 1024, 1200))
        if sched == 106:
            shapes += ((6144, 2560, 2408), (2560, 4096, 1920))
        for (M, N, K) in shapes:
            A = _rand((M, K), 0.5, seed=40); Bm = _rand((N, K), 0.5, seed=41)
            At, Bt = A.t().contiguous(), Bm.t().contiguous()              # k-major copies [K, M], [K, N]
            ref = A.float() @ Bm.float().t()
            for tA, tB in ((False, True), (True, True), (True, False)):
                C = torch.empty((M, N), device="cuda", dtype=BF16)
                ops.gemm_raw(At if tA else A, Bt if tB else Bm, C, M, N, K, M if tA else K, N if tB else K, N, transA=tA, transB=tB)
                assert _relerr(C, ref) < 4e-3, (sched, M, N, K, tA, tB, _relerr(C, ref))
            C = _rand((M, N), 1.0, seed=42); C0 = C.clone()
            ops.gemm_raw(At, Bt, C, M, N, K, M, N, N, transA=True, transB=True, accumulate=True)
            assert _relerr(C, ref + C0.float()) < 4e-3
            R = _rand((M, N), 1.0, seed=43)
            C = torch.empty((M, N), device="cuda", dtype=BF16)
            ops.gemm_raw(A, Bt, C, M, N, K, K, N, N, transB=True, R=R, ldr=N)
            assert _relerr(C, ref.to(BF16).float() + R.float()) < 4e-3
            C32 = torch.empty((M, N), device="cuda", dtype=F32)
            ops.gemm_raw(At, Bt, C32, M, N, K, M, N, N, transA=True, transB=True)
            assert _relerr(C32, ref) < 1e-5
    finally:
        ops.gemm_force_config(-3)


@pytest.mark.parametrize("transB", [True, False])
def test_gemm_swiglu_bwd_epilogue_equals_two_launches(ops, transB):
    """vq3_gemm_swiglu_bwd (down_proj dgrad with the SwiGLU backward in its epilogue) against the two launches it replaces -
    vq3_gemm_bf16_nt then vq3_silu_mul_bwd - for the k-major weight layout the backward uses and for the W^T (NT) form."""
    M, H, I = 300, 256, 520                                   # edges inside tiles in both M and N
    dY = _rand((M, H), 1.0, seed=30)
    W = _rand((H, I), 0.1, seed=31)                           # down_proj.weight [out = H, in = I]
    gu = _rand((M, 2 * I), 1.5, seed=32)
    Wop = W if transB else W.t().contiguous()
    d_act = torch.empty((M, I), device="cuda", dtype=BF16)
    if transB:
        ops.gemm_raw(dY, W, d_act, M, I, H, H, I, I, transB=True)
    else:
        ops.linear(dY, Wop, out=d_act)
    ref = ops.silu_mul_bwd(d_act, gu)
    got = ops.gemm_swiglu_bwd(dY, Wop, gu, transB=transB)
    assert torch.equal(got, ref)                              # same staged bf16 d(act), same arithmetic
    # and both agree with autograd of silu(g) * u
    g32 = gu[:, :I].float().requires_grad_(True); u32 = gu[:, I:].float().requires_grad_(True)
    (torch.nn.functional.silu(g32) * u32).backward(dY.float() @ W.float())
    assert _relerr(got[:, :I], g32.grad) < 1e-2 and _relerr(got[:, I:], u32.grad) < 1e-2


@pytest.mark.parametrize("cfg", [-3, 20, 21, 22, 24])
@pytest.mark.parametrize("M,H,I", [(300, 256, 384), (1200, 2560, 9728), (77, 128, 128)])
def test_gemm_swiglu_fwd_epilogue_equals_two_launches(ops, cfg, M, H, I):
    """vq3_gemm_swiglu_fwd (gate|up projection with silu(gate) * up in its epilogue) against the two launches it replaces - vq3_gemm_bf16_nt
    then vq3_silu_mul_fwd: gu and act bit-identical (same k order per element, same rounding points); act written into a row block of a
    longer slab, as the trainer's deferred weight-gradient operands are; ragged M inside a tile."""
    x = _rand((M, H), 1.0, seed=50)
    W = _rand((2 * I, H), 0.05, seed=51)                      # [gate rows | up rows]
    ops.gemm_force_config(cfg)
    try:
        gu_ref = ops.linear(x, W)
        act_ref = ops.silu_mul_fwd(gu_ref)
        slab = torch.full((M + 5, I), 7.0, device="cuda", dtype=BF16)
        gu, act = ops.gemm_swiglu_fwd(x, W, act_out=slab[2:2 + M])
    finally:
        ops.gemm_force_config(-3)
    assert act.data_ptr() == slab[2:].data_ptr()
    if cfg == -3:                  # (the tuner may give the two paths different tile shapes: compare the fused pair with itself)
        assert _relerr(gu, gu_ref) < 1e-3
        assert torch.equal(act, ops.silu_mul_fwd(gu))
    else:
        assert torch.equal(gu, gu_ref)
        assert torch.equal(act, act_ref)
    assert (slab[:2] == 7.0).all() and (slab[2 + M:] == 7.0).all()            # nothing outside the row block
    ref = torch.nn.functional.silu(gu_ref[:, :I].float()).to(BF16).float() * gu_ref[:, I:].float()
    assert _relerr(act, ref) < 4e-3
    # without the gate|up output (a forward that no backward follows): the same act, bit for bit
    try:
        ops.gemm_force_config(cfg)
        none, act_only = ops.gemm_swiglu_fwd(x, W, keep_gu=False)
    finally:
        ops.gemm_force_config(-3)
    assert none is None and (torch.equal(act_only, act) if cfg != -3 else _relerr(act_only, act) < 1e-3)
    with pytest.raises(RuntimeError):
        ops.gemm_swiglu_fwd(x, _rand((2 * 200, H), 0.05, seed=52))             # I % 128 != 0


def test_gemm_batched_strided(ops):
    """Attention-shaped use: batch (b, h) with K/V shared by groups of heads and a strided output."""
    Bz, Hq, Hkv, L, D = 2, 8, 2, 200, 128
    Q = _rand((Bz, Hq, L, D), 1.0, seed=10); Kt = _rand((Bz, Hkv, L, D), 1.0, seed=11)
    S = torch.empty((Bz, Hq, L, 256), device="cuda", dtype=F32).fill_(float("nan"))
    ops.gemm_raw(Q, Kt, S, L, L, D, D, D, 256, nb1=Bz, nb2=Hq, b2divB=Hq // Hkv, sA=(Hq * L * D, L * D),
                 sB=(Hkv * L * D, L * D), sC=(Hq * L * 256, L * 256), alpha=D ** -0.5)
    ref = torch.einsum("bhld,bhmd->bhlm", Q.float(), Kt.float().repeat_interleave(Hq // Hkv, 1)) * D ** -0.5
    assert _relerr(S[..., :L], ref) < 1e-5
    assert torch.isnan(S[..., L:]).all(), "columns beyond N must not be written"
    # P @ V with V^T operand and output scattered into [B, L, Hq*D]
    P = _rand((Bz, Hq, L, 256), 0.1, seed=12); P[..., L:] = 0
    Vt = _rand((Bz, Hkv, D, 256), 1.0, seed=13)
    O = torch.zeros((Bz, L, Hq * D), device="cuda", dtype=BF16)
    ops.gemm_raw(P, Vt, O, L, D, 256, 256, 256, Hq * D, nb1=Bz, nb2=Hq, b2divB=Hq // Hkv, sA=(Hq * L * 256, L * 256),
                 sB=(Hkv * D * 256, D * 256), sC=(L * Hq * D, D))
    ref = torch.einsum("bhlm,bhdm->blhd", P.float(), Vt.float().repeat_interleave(Hq // Hkv, 1)).reshape(Bz, L, Hq * D)
    assert _relerr(O, ref) < 4e-3


def test_gemm_rejects_bad_args(ops):
    from vggt_qwen3_amd._lib import Vq3Error
    A = _rand((16, 100), seed=1); W = _rand((16, 100), seed=2)
    with pytest.raises(Vq3Error):
        ops.linear(A, W)  # K % 8 != 0


# ------------------------------------------------------------------------------------------ norms
def test_rmsnorm_fwd_bwd(ops):
    rows, cols, eps = 1200, 2560, 1e-6
    x = _rand((rows, cols), 2.0, seed=20); w = _rand((cols,), 1.0, seed=21)
    y, rstd = ops.rmsnorm_fwd(x, w, eps, want_rstd=True)
    xf = x.float()
    rs = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    ref = (w.float() * (xf * rs).to(BF16).float()).to(BF16)
    assert _relerr(y, ref) < 2e-3 and _maxerr(rstd, rs.squeeze(-1)) < 1e-5
    # backward vs autograd (fp32, no intermediate rounding)
    dy = _rand((rows, cols), 1.0, seed=22); dres = _rand((rows, cols), 1.0, seed=23)
    xa = xf.clone().requires_grad_(True); wa = w.float().clone().requires_grad_(True)
    ya = wa * (xa * torch.rsqrt(xa.pow(2).mean(-1, keepdim=True) + eps))
    ya.backward(dy.float())
    dw = _rand((cols,), 1.0, seed=24); dw0 = dw.clone()
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dres, dw, True)
    assert _relerr(dx, xa.grad + dres.float()) < 4e-3
    assert _relerr(dw, wa.grad + dw0.float()) < 4e-3
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dres, dw, False)
    assert _relerr(dw, wa.grad) < 4e-3


@pytest.mark.parametrize("xdt", [BF16, F32])
def test_layernorm_fwd(ops, xdt):
    rows, cols, eps = 770, 4096, 1e-5
    x = _rand((rows, cols), 2.0, xdt, seed=30); res = _rand((rows, cols), 1.0, xdt, seed=31)
    w = _rand((cols,), 1.0, F32, seed=32); b = _rand((cols,), 1.0, F32, seed=33)
    yb, yf = ops.layernorm_fwd(x, w, b, eps, res=res, want_bf16=True, want_f32=True)
    s = x.float() + res.float()
    if xdt == BF16:
        s = s.to(BF16).float()
    ref = torch.nn.functional.layer_norm(s, (cols,), w, b, eps)
    assert _relerr(yf, ref) < 1e-5 and _relerr(yb, ref) < 3e-3
    yb2, _ = ops.layernorm_fwd(x, w, b, eps)
    assert _relerr(yb2, torch.nn.functional.layer_norm(x.float(), (cols,), w, b, eps)) < 3e-3


# ------------------------------------------------------------------------------------------ element-wise
def test_silu_mul_fwd_bwd(ops):
    rows, inter = 300, 9728
    gu = _rand((rows, 2 * inter), 1.5, seed=40)
    act = ops.silu_mul_fwd(gu)
    g, u = gu[:, :inter].float(), gu[:, inter:].float()
    ref = torch.nn.functional.silu(g).to(BF16).float() * u
    assert _relerr(act, ref) < 3e-3
    dact = _rand((rows, inter), 1.0, seed=41)
    ga = g.clone().requires_grad_(True); ua = u.clone().requires_grad_(True)
    (torch.nn.functional.silu(ga) * ua).backward(dact.float())
    dgu = ops.silu_mul_bwd(dact, gu)
    assert _relerr(dgu[:, :inter], ga.grad) < 3e-3 and _relerr(dgu[:, inter:], ua.grad) < 3e-3


def test_transpose(ops):
    x = _rand((200, 128), seed=50)
    t = ops.transpose2d(x, pad_to=64)
    assert t.shape == (128, 256)
    assert torch.equal(t[:, :200], x.t()) and (t[:, 200:] == 0).all()
    # batched with strides: [2,3,70,40] -> [2,3,40,(70 -> 128)]
    x = _rand((2, 3, 70, 40), seed=51)
    out = torch.full((2, 3, 40, 128), 7.0, device="cuda", dtype=BF16)
    ops.transpose_raw(x, out, 70, 40, 128, 40, 128, n=(1, 2, 3), s=(0, 3 * 70 * 40, 70 * 40), d=(0, 3 * 40 * 128, 40 * 128))
    assert torch.equal(out[..., :70], x.transpose(-1, -2)) and (out[..., 70:] == 0).all()


def test_cast_and_acc(ops):
    x = _rand((1000003,), 3.0, F32, seed=60)
    assert torch.equal(ops.cast(x, BF16), x.to(BF16))
    xb = x.to(BF16)
    assert torch.equal(ops.cast(xb, F32), xb.float())
    acc = _rand((1000003,), 1.0, BF16, seed=61); acc0 = acc.clone()
    ops.f32_to_bf16_acc(x, acc, True)
    assert torch.equal(acc, (x + acc0.float()).to(BF16))


def test_gather_scatter(ops):
    src = _rand((1200, 2560), seed=70)
    idx = torch.tensor([5, 1199, 0, 77, 640], dtype=torch.int32, device="cuda")
    out = ops.gather_rows(src, idx, 5, 16)
    assert torch.equal(out[:5], src[idx.long()]) and (out[5:] == 0).all()
    dst = torch.zeros_like(src)
    ops.scatter_rows(out, idx, dst, 5, False)
    assert torch.equal(dst[idx.long()], src[idx.long()])
    ops.scatter_rows(out, idx, dst, 5, True)
    assert torch.equal(dst[idx.long()], (2 * src[idx.long()].float()).to(BF16))


# ------------------------------------------------------------------------------------------ qwen attention pieces
def _rope_tables(L, D, theta=5e6):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=F32) / D))
    fr = torch.arange(L, dtype=F32)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    return emb.cos().to(BF16).cuda(), emb.sin().to(BF16).cuda()


def _rot_half(x):
    return torch.cat([-x[..., 64:], x[..., :64]], -1)


def _qkprep_ref(qkv, qw, kw, cos, sin, B, L, Hq, Hkv, D, eps, exact_rounding):
    x = qkv.float().reshape(B, L, Hq + 2 * Hkv, D)
    q, k, v = x[:, :, :Hq], x[:, :, Hq:Hq + Hkv], x[:, :, Hq + Hkv:]

    def norm(t, w):
        n = t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + eps)
        if exact_rounding:
            n = n.to(BF16).float()
        o = w.float() * n
        return o.to(BF16).float() if exact_rounding else o

    def rope(t):
        c, s = cos.float()[None, :, None, :], sin.float()[None, :, None, :]
        if exact_rounding:
            return ((t * c).to(BF16).float() + (_rot_half(t) * s).to(BF16).float()).to(BF16).float()
        return t * c + _rot_half(t) * s

    return rope(norm(q, qw)).transpose(1, 2), rope(norm(k, kw)).transpose(1, 2), v.transpose(1, 2)


def test_qkprep_fwd_bwd(ops):
    B, L, Hq, Hkv, D, eps = 2, 200, 8, 2, 128, 1e-6
    qkv = _rand((B * L, (Hq + 2 * Hkv) * D), 1.0, seed=80)
    qw = (_rand((D,), 0.2, seed=81).float() + 1).to(BF16); kw = (_rand((D,), 0.2, seed=82).float() + 1).to(BF16)
    cos, sin = _rope_tables(L, D)
    Q, K, V, qr, kr = ops.qwen_qkprep_fwd(qkv, qw, kw, cos, sin, B, L, Hq, Hkv, D, eps)
    rq, rk, rv = _qkprep_ref(qkv, qw, kw, cos, sin, B, L, Hq, Hkv, D, eps, True)
    assert _relerr(Q, rq) < 2e-3 and _relerr(K, rk) < 2e-3 and torch.equal(V.float(), rv)
    # backward vs autograd of the un-rounded function
    xa = qkv.float().clone().requires_grad_(True); qwa = qw.float().clone().requires_grad_(True)
    kwa = kw.float().clone().requires_grad_(True)
    aq, ak, av = _qkprep_ref(xa, qwa, kwa, cos, sin, B, L, Hq, Hkv, D, eps, False)
    dQ = _rand((B, Hq, L, D), 1.0, seed=83); dK = _rand((B, Hkv, L, D), 1.0, seed=84); dV = _rand((B, Hkv, L, D), 1.0, seed=85)
    ((aq * dQ.float()).sum() + (ak * dK.float()).sum() + (av * dV.float()).sum()).backward()
    dqw = torch.zeros(D, device="cuda", dtype=BF16); dkw = torch.zeros(D, device="cuda", dtype=BF16)
    dqkv = ops.qwen_qkprep_bwd(dQ, dK, dV, qkv, qw, kw, cos, sin, qr, kr, dqw, dkw, False, B, L, Hq, Hkv, D)
    assert _relerr(dqkv, xa.grad) < 4e-3, _relerr(dqkv, xa.grad)
    assert _relerr(dqw, qwa.grad) < 4e-3 and _relerr(dkw, kwa.grad) < 4e-3
    # dK / dV handed over as two partial slabs (what the split dK/dV attention pass writes): same result
    half = lambda t: torch.stack([(t.float() * 0.5).to(BF16), (t.float() - (t.float() * 0.5).to(BF16).float()).to(BF16)])
    dqw2 = torch.zeros(D, device="cuda", dtype=BF16); dkw2 = torch.zeros(D, device="cuda", dtype=BF16)
    dqkv2 = ops.qwen_qkprep_bwd(dQ, half(dK), half(dV), qkv, qw, kw, cos, sin, qr, kr, dqw2, dkw2, False, B, L, Hq, Hkv, D)
    assert _relerr(dqkv2, dqkv) < 4e-3 and _relerr(dkw2, dkw) < 4e-3


def test_softmax_fwd_bwd(ops):
    nb, Lq, Lk, ld = 12, 200, 200, 256
    S = _rand((nb, Lq, ld), 3.0, F32, seed=90)
    km = (torch.rand(3, Lk) > 0.2).to(torch.uint8); km[:, 0] = 1; km = km.cuda()
    P = ops.softmax_fwd(S, km, 4, Lk, ld, True)
    mask = torch.tril(torch.ones(Lq, Lk, device="cuda", dtype=torch.bool))[None] & km.bool().repeat_interleave(4, 0)[:, None, :]
    ref = torch.softmax(S[..., :Lk].masked_fill(~mask, float("-inf")), -1)
    assert _maxerr(P[..., :Lk], ref) < 4e-3 and (P[..., Lk:] == 0).all()
    dP = _rand((nb, Lq, ld), 1.0, F32, seed=91)
    dS = ops.softmax_bwd(P, dP, Lk, 0.25)
    Pf = P.float()[..., :Lk]
    refd = 0.25 * Pf * (dP[..., :Lk] - (Pf * dP[..., :Lk]).sum(-1, keepdim=True))
    assert _relerr(dS[..., :Lk], refd) < 4e-3 and (dS[..., Lk:] == 0).all()
    # non-causal, no mask
    P2 = ops.softmax_fwd(S, None, 1, Lk, ld, False)
    assert _maxerr(P2[..., :Lk], torch.softmax(S[..., :Lk], -1)) < 4e-3


# ------------------------------------------------------------------------------------------ embedding / loss / optimiser
def test_embed_splice_fwd_bwd(ops):
    B, L, H, S, V = 3, 200, 2560, 136, 1000
    g = torch.Generator().manual_seed(100)
    ids = torch.randint(0, V, (B, L), generator=g).cuda()
    table = _rand((V, H), seed=101); feat = _rand((B, S, H), seed=102)
    srcmap = torch.full((B, L), -1, dtype=torch.int32)
    for b, pos in enumerate([10, 20, 64]):
        srcmap[b, pos:pos + S] = torch.arange(S, dtype=torch.int32)
    srcmap[2] = -1  # row without <image>
    srcmap = srcmap.cuda()
    out = ops.embed_splice_fwd(ids, table, feat, srcmap, B, L, H, S)
    ref = table[ids]
    for b in range(B):
        sel = srcmap[b] >= 0
        ref[b, sel] = feat[b, srcmap[b, sel].long()]
    assert torch.equal(out, ref)
    dout = _rand((B, L, H), seed=103)
    sorted_ids, order = torch.sort(ids.reshape(-1), stable=True)
    dtable = _rand((V, H), 0.1, seed=104); dtable0 = dtable.clone()
    dfeat = torch.zeros((B, S, H), device="cuda", dtype=F32)
    ops.embed_splice_bwd(sorted_ids, order, srcmap, dout, dtable, dfeat, B, L, H, S)
    reft = torch.zeros((V, H), device="cuda", dtype=F32)
    keep = (srcmap < 0).reshape(-1)
    reft.index_add_(0, ids.reshape(-1)[keep], dout.reshape(-1, H).float()[keep])
    assert _relerr(dtable, dtable0.float() + reft) < 4e-3
    reff = torch.zeros_like(dfeat)
    for b in range(B):
        sel = srcmap[b] >= 0
        reff[b, srcmap[b, sel].long()] = dout[b, sel].float()
    assert torch.equal(dfeat, reff)


def test_cross_entropy(ops):
    n, V = 24, 151937
    ld = (V + 7) // 8 * 8
    logits = torch.zeros((n, ld), device="cuda", dtype=BF16)
    logits[:, :V] = _rand((n, V), 2.0, seed=110)
    tgt = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(111)).to(torch.int32).cuda()
    la = logits[:, :V].float().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(la, tgt.long(), reduction="sum")
    ref.backward()
    loss = torch.zeros(1, device="cuda", dtype=F32)
    ops.cross_entropy_fwd_bwd(logits, tgt, loss, n, V, 1.0 / n)
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-5
    assert _relerr(logits[:, :V], la.grad / n) < 4e-3 and (logits[:, V:] == 0).all()


def test_adamw(ops):
    n = 1000003
    p = _rand((n,), 1.0, F32, seed=120); g = _rand((n,), 0.1, BF16, seed=121)
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    m = torch.zeros_like(p); v = torch.zeros_like(p); w = torch.empty(n, device="cuda", dtype=BF16)
    for step in (1, 2, 3):
        ref_p.grad = g.float()
        opt.step()
        ops.adamw_step(p, m, v, g, w, 1e-3, 0.9, 0.999, 1e-8, 0.1, step)
    assert _maxerr(p, ref_p.data) < 1e-5, _maxerr(p, ref_p.data)
    assert torch.equal(w, p.to(BF16))


# ------------------------------------------------------------------------------------------ GEMM v3: operand layouts, K tails
@pytest.mark.parametrize("tA,tB", [(False, True), (True, False), (True, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1200, 2560, 1200), (200, 128, 200), (2560, 9728, 1200),
                                   (136, 264, 72), (6144, 2560, 1208)])
def test_gemm_layouts_and_k_tail(ops, tA, tB, M, N, K):
    if K % 64 == 0 and not (tA or tB):
        pytest.skip("plain NT with full K tiles is covered by test_gemm_shapes")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    Bm = torch.randn(N, K, generator=g).to(BF16).cuda()
    ref = A.float() @ Bm.float().t()
    a_op = A.t().contiguous() if tA else A            # k-major: [K, M]
    b_op = Bm.t().contiguous() if tB else Bm
    out = torch.full((M, N), float("nan"), device="cuda", dtype=F32)
    ops.gemm_raw(a_op, b_op, out, M, N, K, a_op.stride(0), b_op.stride(0), N, transA=tA, transB=tB)
    e = _relerr(out, ref)
    assert e < 2e-6 * math.sqrt(K) + 1e-6, f"tA={tA} tB={tB} rel err {e}"


def test_gemm_kmajor_identity_asymmetric(ops):
    """Exact-integer check of the transposed-read fragment maps (k order, row/col) for both k-major operands."""
    K, M, N = 128, 128, 256
    A = torch.eye(K, dtype=F32)[:M].to(BF16).cuda()                      # [M, K] = I
    Bm = (torch.arange(N * K, dtype=F32).reshape(N, K) % 251 - 125).to(BF16).cuda()
    ref = A.float() @ Bm.float().t()
    for tA, tB in ((True, True), (True, False), (False, True)):
        a_op = A.t().contiguous() if tA else A
        b_op = Bm.t().contiguous() if tB else Bm
        out = torch.empty((M, N), device="cuda", dtype=F32)
        ops.gemm_raw(a_op, b_op, out, M, N, K, a_op.stride(0), b_op.stride(0), N, transA=tA, transB=tB)
        assert torch.equal(out, ref), (tA, tB, _maxerr(out, ref))


def test_gemm_kmajor_batched_attention_shapes(ops):
    """dK = sum_g dS^T Q with the (group, query) pair as the contraction and P.V with V as stored (no transposes)."""
    Bz, Hq, Hkv, L, D, Lp = 2, 8, 2, 200, 128, 256
    G = Hq // Hkv
    dS = _rand((Bz, Hq, L, Lp), 0.2, seed=31); dS[..., L:] = 0
    Q = _rand((Bz, Hq, L, D), 1.0, seed=32)
    dK = torch.empty((Bz, Hkv, L, D), device="cuda", dtype=F32)
    # A = dS[b, kv*G:(kv+1)*G] viewed as k-major [G*L, Lp] (M = L keys), B = Q[b, kv*G..] as k-major [G*L, D]
    ops.gemm_raw(dS, Q, dK, L, D, G * L, Lp, D, D, nb1=Bz, nb2=Hkv, sA=(Hq * L * Lp, G * L * Lp),
                 sB=(Hq * L * D, G * L * D), sC=(Hkv * L * D, L * D), transA=True, transB=True)
    ref = torch.einsum("bkgqj,bkgqd->bkjd", dS.float().view(Bz, Hkv, G, L, Lp)[..., :L], Q.float().view(Bz, Hkv, G, L, D))
    assert _relerr(dK, ref) < 1e-5
    P = _rand((Bz, Hq, L, Lp), 0.1, seed=33); P[..., L:] = 0
    V = _rand((Bz, Hkv, L, D), 1.0, seed=34)
    O = torch.zeros((Bz, L, Hq * D), device="cuda", dtype=F32)
    ops.gemm_raw(P, V, O, L, D, L, Lp, D, Hq * D, nb1=Bz, nb2=Hq, b2divB=G, sA=(Hq * L * Lp, L * Lp),
                 sB=(Hkv * L * D, L * D), sC=(L * Hq * D, D), transB=True)
    ref = torch.einsum("bhlm,bhmd->blhd", P.float()[..., :L], V.float().repeat_interleave(G, 1)).reshape(Bz, L, Hq * D)
    assert _relerr(O, ref) < 1e-5


def test_gemm_split_k(ops):
    M, N, K = 16, 2560, 152000
    A = _rand((M, K), 0.1, seed=71); Bk = _rand((K, N), 0.1, seed=72)   # B k-major
    ref = A.float() @ Bk.float()
    out = torch.zeros((M, N), device="cuda", dtype=F32)
    ops.gemm_raw(A, Bk, out, M, N, K, K, N, N, transB=True, ksplit=12, alpha=0.5)
    assert _relerr(out, 0.5 * ref) < 1e-4


@pytest.mark.parametrize("B,Hkv,L,pad,G", [(1, 1, 32, "none", 4), (2, 2, 40, "right", 4), (3, 1, 200, "right", 4),
                                            (2, 2, 104, "left", 4), (1, 2, 264, "none", 4), (2, 1, 72, "right", 2),
                                            (1, 3, 50, "left", 1)])
def test_qwen_flash_attention_fwd_bwd(ops, B, Hkv, L, pad, G):
    """Fused causal GQA attention vs an fp32 autograd reference: padding on either side (left padding makes whole query rows
    maskless: their output and gradients must be exactly zero, never NaN), L not a multiple of the 32-row blocks."""
    torch.manual_seed(L)
    Hq, D = G * Hkv, 128
    dev = "cuda"
    Q = torch.randn(B, Hq, L, D, device=dev).to(BF16)
    K = torch.randn(B, Hkv, L, D, device=dev).to(BF16)
    V = torch.randn(B, Hkv, L, D, device=dev).to(BF16)
    dO = torch.randn(B * L, Hq * D, device=dev).to(BF16)
    mask = torch.ones(B, L, dtype=torch.uint8, device=dev)
    if pad == "right":
        for b in range(B):
            mask[b, L - 3 - 7 * b:] = 0
    elif pad == "left":
        for b in range(B):
            mask[b, : 2 + 5 * b] = 0
    scale = D ** -0.5
    O, lse = ops.qwen_flash_fwd(Q, K, V, mask, B, L, Hq, Hkv, D, scale)
    dQ, dK, dV = ops.qwen_flash_bwd(Q, K, V, mask, O, dO, lse, B, L, Hq, Hkv, D, scale, kv_parts=1)
    # the split form (partial dK / dV slabs, one per group of query blocks) must add up to the same gradients
    for parts in (2, 3):
        dQp, dKp, dVp = ops.qwen_flash_bwd(Q, K, V, mask, O, dO, lse, B, L, Hq, Hkv, D, scale, kv_parts=parts)
        assert dKp.shape == (parts, B, Hkv, L, D) and torch.equal(dQp, dQ)
        for a, b in ((dKp, dK), (dVp, dV)):
            assert ((a.float().sum(0) - b.float()).norm() / (b.float().norm() + 1e-9)).item() < 1e-2
    q, k, v = (t.float().detach().requires_grad_(True) for t in (Q, K, V))
    ke, ve = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    s = (q @ ke.transpose(2, 3)) * scale
    vis = torch.tril(torch.ones(L, L, dtype=torch.bool, device=dev))[None, None] & (mask.bool()[:, None, None, :])
    s = s.masked_fill(~vis, float("-inf"))
    rowvis = vis.any(-1, keepdim=True)
    p = torch.softmax(s.masked_fill(~rowvis, 0.0), dim=-1) * rowvis          # maskless rows -> 0
    ref = (p @ ve).permute(0, 2, 1, 3).reshape(B * L, Hq * D)
    ref.backward(dO.float())
    def rel(a, b):
        return ((a.float() - b).norm() / (b.norm() + 1e-9)).item()
    assert torch.isfinite(O.float()).all() and torch.isfinite(dQ.float()).all() and torch.isfinite(dK.float()).all()
    assert rel(O, ref.detach()) < 1e-2
    assert rel(dQ, q.grad) < 2e-2 and rel(dK, k.grad) < 2e-2 and rel(dV, v.grad) < 2e-2
    dead = ~rowvis.expand(B, Hq, L, 1).reshape(B, Hq, L)
    if dead.any():
        Oh = O.view(B, L, Hq, D).permute(0, 2, 1, 3)
        assert (Oh[dead] == 0).all() and (dQ[dead] == 0).all()
    # reference LSE (log2 domain of the scaled scores)
    lse_ref = torch.logsumexp(s.detach(), dim=-1) * 1.4426950408889634
    ok = rowvis.squeeze(-1).expand(B, Hq, L)
    assert (lse[ok] - lse_ref[ok]).abs().max() < 2e-2


def test_gemm_tune_file_makes_the_choice_repeatable(tmp_path):
    """VQ3_GEMM_TUNE_FILE: the measured kernel choices of one process are appended to the file; a second process reads them and measures
    nothing (same kernels, same summation order from run to run and on every rank that is pointed at the file)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune = tmp_path / "tune.txt"
    code = ("import torch; from vggt_qwen3_amd import ops; x = torch.randn(512, 1024, device='cuda').to(torch.bfloat16); "
            "w = torch.randn(768, 1024, device='cuda').to(torch.bfloat16); y = ops.linear(x, w); torch.cuda.synchronize(); "
            "print('ok', float(y.float().abs().sum()) > 0)")
    env = dict(os.environ, VQ3_GEMM_TUNE_FILE=str(tune), VQ3_GEMM_AUTOTUNE_LOG="1", PYTHONPATH=repo, VQ3_GEMM_TUNE_TABLE="0")
    first = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=repo)
    assert first.returncode == 0 and "ok True" in first.stdout, first.stderr[-2000:]
    assert "[vq3 gemm autotune] M=512 N=768 K=1024" in first.stderr
    lines = [l.split() for l in tune.read_text().splitlines() if l.strip()]
    assert any(l[:4] == ["512", "768", "1024", "1"] and len(l) == 6 for l in lines)
    second = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=repo)
    assert second.returncode == 0 and "ok True" in second.stdout, second.stderr[-2000:]
    assert "[vq3 gemm autotune]" not in second.stderr                      # nothing was measured: the table came from the file
    assert "[vq3 gemm choice] M=512 N=768 K=1024" in second.stderr and "(table)" in second.stderr
    assert tune.read_text().count("512 768 1024 1 ") == 1
    # the same file as a TABLE (what the package ships): read, never appended to; with measuring held (a multi-rank job) a shape the
    # table does not know takes the heuristic, and says so
    code2 = ("import torch; from vggt_qwen3_amd import ops, _lib; ops.gemm_tune_setup(); _lib.load().vq3_gemm_autotune_hold(1); "
             "x = torch.randn(512, 1024, device='cuda').to(torch.bfloat16); w = torch.randn(768, 1024, device='cuda').to(torch.bfloat16); "
             "w2 = torch.randn(640, 1024, device='cuda').to(torch.bfloat16); y = ops.linear(x, w); z = ops.linear(x, w2); "
             "torch.cuda.synchronize(); print('ok', float((y.float().abs().sum() + z.float().abs().sum())) > 0)")
    env3 = dict(os.environ, VQ3_GEMM_TUNE_TABLE=str(tune), VQ3_GEMM_AUTOTUNE_LOG="1", PYTHONPATH=repo)
    env3.pop("VQ3_GEMM_TUNE_FILE", None)
    before = tune.read_text()
    third = subprocess.run([sys.executable, "-c", code2], env=env3, capture_output=True, text=True, timeout=300, cwd=repo)
    assert third.returncode == 0 and "ok True" in third.stdout, third.stderr[-2000:]
    assert "[vq3 gemm autotune]" not in third.stderr and tune.read_text() == before
    assert "M=512 N=768 K=1024" in third.stderr and "(table)" in third.stderr
    assert "M=512 N=640 K=1024" in third.stderr and "(heuristic)" in third.stderr


def test_gemm_tuner_measures_in_the_callers_workspace(ops, monkeypatch):
    """VERDICT r3 item 6 + ADVICE r4: trial output and cache-flush buffer of the kernel-choice measurements come from torch-owned memory
    that the library asks for through a provider callback (ops.gemm_tune_setup: vq3_gemm_workspace_provider) the moment it is about to
    measure - a first-sight shape allocates nothing through the HIP allocator behind torch's back; when the provider has nothing to give
    (VQ3_GEMM_TUNE_WS_MB=0) the shape is not measured and the product is still right; a process whose shapes the shipped table knows, and
    a process with measuring held (a rank of a multi-rank job), never hold the workspace at all."""
    import os
    import subprocess
    import sys
    ops.gemm_tune_setup()
    dev = torch.cuda.current_device()
    x = _rand((640, 1024), 1.0, seed=1); w = _rand((896, 1024), 0.05, seed=2)
    y = ops.linear(x, w)                                   # first sight of (640, 896, 1024): measured in the provider's block
    torch.cuda.synchronize()
    blk = ops._TUNE_WS["tune"].get(dev)
    assert blk is not None and blk.numel() >= (512 << 20)
    assert _relerr(y, x.float() @ w.float().t()) < 4e-3
    free0 = torch.cuda.mem_get_info()[0]
    y1 = ops.linear(_rand((672, 1024), 1.0, seed=4), w)    # another first-sight shape: the same block, nothing new from the driver
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (64 << 20) and y1.shape == (672, 896)
    assert ops._TUNE_WS["tune"][dev] is blk
    # a provider with nothing to give: not measured, still correct
    from vggt_qwen3_amd import _lib
    assert _lib.load().vq3_gemm_tune_workspace(None, 0) == 0
    ops._TUNE_WS["tune"].clear()
    monkeypatch.setenv("VQ3_GEMM_TUNE_WS_MB", "0")
    xs = _rand((704, 1024), 1.0, seed=3)
    y2 = ops.linear(xs, w)
    assert _relerr(y2, xs.float() @ w.float().t()) < 4e-3 and not ops._TUNE_WS["tune"]
    monkeypatch.delenv("VQ3_GEMM_TUNE_WS_MB")
    # fresh processes: (a) only table-known shapes -> no workspace; (b) measuring held -> none either, whatever the shape
    repo = str(Path(__file__).resolve().parents[1])
    code = ("import torch; from vggt_qwen3_amd import ops, _lib; ops.gemm_tune_setup(); HOLD and _lib.load().vq3_gemm_autotune_hold(1); "
            "x = torch.randn(M_, 2560, device='cuda').to(torch.bfloat16); w = torch.randn(6144, 2560, device='cuda').to(torch.bfloat16); "
            "y = ops.linear(x, w); torch.cuda.synchronize(); print('ws', len(ops._TUNE_WS['tune']), float(y.float().abs().sum()) > 0)")
    env = dict(os.environ, PYTHONPATH=repo)
    env.pop("VQ3_GEMM_TUNE_FILE", None)
    a = subprocess.run([sys.executable, "-c", code.replace("HOLD", "False").replace("M_", "12000")], env=env, capture_output=True, text=True, timeout=300, cwd=repo)
    assert a.returncode == 0 and "ws 0 True" in a.stdout, (a.stdout, a.stderr[-1500:])          # (12000, 6144, 2560) is in the shipped table
    b = subprocess.run([sys.executable, "-c", code.replace("HOLD", "True").replace("M_", "1111")], env=env, capture_output=True, text=True, timeout=300, cwd=repo)
    assert b.returncode == 0 and "ws 0 True" in b.stdout, (b.stdout, b.stderr[-1500:])


# ------------------------------------------------------------------------------------------ fused Perceiver cross-attention
def _xattn_ref(q, kv, B, H, N, T, hd):
    """fp32 torch: softmax(q k^T / sqrt(hd)) and its product with v, per (sample, head)."""
    D = H * hd
    qf = q.float().view(B, N, H, hd).permute(0, 2, 1, 3)
    kf = kv.float()[:, :D].reshape(B, T, H, hd).permute(0, 2, 1, 3)
    vf = kv.float()[:, D:].reshape(B, T, H, hd).permute(0, 2, 1, 3)
    P = torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, dim=-1)            # [B, H, N, T]
    return P, vf


@pytest.mark.parametrize("B,H,N,T,hd", [(2, 2, 16, 70, 64), (3, 8, 128, 128, 512), (2, 3, 100, 200, 128), (1, 2, 64, 33, 256),
                                       (2, 1, 130, 1, 64), (1, 4, 8, 257, 128)])
def test_perceiver_xattn_fused_vs_torch(ops, B, H, N, T, hd):
    """vq3_perceiver_xattn_fwd against fp32 torch (projector_perceiver.py:44: MultiheadAttention(latents, context, context)):
    ragged latent blocks (N not a multiple of 64), context lengths that are not multiples of the 32-key chunk (masked tail,
    zero-filled V rows), a single key, every instantiated head size; the kept P has zero pad columns."""
    D = H * hd
    q, kv = _rand((B * N, D), 1.5, seed=1), _rand((B * T, 2 * D), 1.0, seed=2)
    Tp = (T + 63) // 64 * 64
    o, P, Pd = ops.perceiver_xattn(q, kv, B, H, N, T, hd, Tp, keep_p=True)
    assert Pd is P
    Pr, vf = _xattn_ref(q, kv, B, H, N, T, hd)
    assert _maxerr(P.view(B, H, N, Tp)[..., :T], Pr) < 4e-3          # bf16 rounding of probabilities <= 1
    assert P.view(B, H, N, Tp)[..., T:].abs().max().item() == 0 if Tp > T else True
    oref = (Pr @ vf).permute(0, 2, 1, 3).reshape(B * N, D)
    assert _relerr(o, oref) < 1e-2
    o2 = ops.perceiver_xattn(q, kv, B, H, N, T, hd, Tp)               # without the kept P: the same product
    assert torch.equal(o2, o)


@pytest.mark.parametrize("B,H,N,T,hd", [(2, 2, 48, 70, 64), (2, 8, 128, 128, 512)])
def test_perceiver_xattn_fused_dropout_is_vq3_dropouts_mask(ops, B, H, N, T, hd):
    """Attention-weight dropout inside the fused kernel: Pd must be BIT-identical to vq3_dropout applied to the kept P with the same
    (seed, offset) - the mask is indexed by the element's position in the [B*H, N, Tp] tensor - and O must be Pd . V."""
    D = H * hd
    q, kv = _rand((B * N, D), 1.5, seed=3), _rand((B * T, 2 * D), 1.0, seed=4)
    Tp = (T + 63) // 64 * 64
    seed, off, p = 0x1234567, 991, 0.25
    o, P, Pd = ops.perceiver_xattn(q, kv, B, H, N, T, hd, Tp, p, seed, off, keep_p=True)
    want = ops.dropout_(P.clone(), p, seed, off)
    assert torch.equal(Pd, want)
    frac = (Pd.view(B, H, N, Tp)[..., :T] == 0).float().mean().item()
    assert abs(frac - p) < 0.03
    vf = kv.float()[:, D:].reshape(B, T, H, hd).permute(0, 2, 1, 3)
    oref = (Pd.float().view(B, H, N, Tp)[..., :T] @ vf).permute(0, 2, 1, 3).reshape(B * N, D)
    assert _relerr(o, oref) < 4e-3                                   # same bf16 weights, f32 accumulation: only O's rounding differs
    o2 = ops.perceiver_xattn(q, kv, B, H, N, T, hd, Tp, p, seed, off)
    assert torch.equal(o2, o)


@pytest.mark.parametrize("train", [False, True])
def test_perceiver_projector_fused_route_equals_three_launch_route(ops, train, monkeypatch):
    """PerceiverProjector.forward with the fused cross-attention against the batched GEMM / softmax / batched GEMM route it replaces
    (VQ3_PERCEIVER_FUSED=0), eval and train mode (the four dropout sites draw the same masks in both routes)."""
    from vggt_qwen3_amd.perceiver import PerceiverConfig, PerceiverProjector
    torch.manual_seed(0)
    cfg = PerceiverConfig(latent_dim=512, num_latents=40, num_heads=4, num_layers=2, ffn_dim=1024, dropout=0.1)
    m = PerceiverProjector(cfg, in_dim=192, out_dim=256).cuda()
    m.train(train)
    x = _rand((3, 72, 192), 1.0, F32, seed=5)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("VQ3_PERCEIVER_FUSED", flag)
        m._drop_offset = 0
        outs.append(m(x).clone())
    assert outs[0].shape == (3, 40, 256)
    assert _relerr(outs[0], outs[1]) < 5e-3


def test_perceiver_xattn_rejects_bad_args(ops):
    from vggt_qwen3_amd import _lib
    q, kv = _rand((16, 96), seed=1), _rand((16, 192), seed=2)
    with pytest.raises((_lib.Vq3Error, AssertionError)):
        ops.perceiver_xattn(q, kv, 1, 1, 16, 16, 96, 64)             # head size without an instantiation


@pytest.mark.parametrize("dt", [BF16, F32])
def test_dropout_mask_does_not_depend_on_the_access_width(ops, dt):
    """vq3_dropout keeps / zeroes element i by a hash of (seed, offset + i) alone: the 16-byte-per-lane walk over an aligned tensor and the
    element-wise walk a misaligned view takes must produce the same values, tail elements included; the kept fraction is 1 - p."""
    n = 1_000_003
    g = torch.Generator().manual_seed(11)
    base = torch.randn(n + 8, generator=g).to(dt).cuda()
    a = base[:n].clone()                               # 16-byte aligned: vector path + scalar tail
    buf = torch.empty(n + 8, device="cuda", dtype=dt)
    b = buf[1:1 + n]                                   # misaligned by one element: scalar path
    b.copy_(base[:n])
    assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 != 0 and b.is_contiguous()
    ops.dropout_(a, 0.3, 987654321, 12345)
    ops.dropout_(b, 0.3, 987654321, 12345)
    assert torch.equal(a, b)
    kept = (a != 0).float().mean().item()
    assert abs(kept - 0.7) < 5e-3
    ref = (base[:n].float() * (1.0 / 0.7)).to(dt)
    m = a != 0
    assert torch.equal(a[m], ref[m])


def test_gemm_v7_ln_fold_is_repeatable(ops):
    """Round 5: cfg 24 (gemm7.hip, two workgroups per CU) staged the folded LayerNorm's (mu, rstd) pairs in LDS behind a raw s_barrier
    without waiting for its own LDS write - about one launch in three at 6174 x 4096 x 1024 one wave normalised 16 rows x 64 columns of
    one tile with stale pairs (errors the size of the activations themselves; found through tests/test_fulldepth_gpu.py). The kernel's
    arithmetic is the 256 x 256 kernel's (same K order, same rounding points): every launch must equal cfg 20 BIT FOR BIT - fc1 with the
    fold + bias + GELU and the fused q|k|v epilogue behind the fold, 40 launches each on a grid of more than two workgroups per CU."""
    torch.manual_seed(0)
    M, C = 6174, 1024
    x = torch.randn(M, C, device="cuda").to(BF16)
    w1 = (torch.randn(4 * C, C, device="cuda") * 0.03).to(BF16)
    wq = (torch.randn(3 * C, C, device="cuda") * 0.03).to(BF16)
    b4, c4 = torch.randn(4 * C, device="cuda"), torch.randn(4 * C, device="cuda")
    b3, c3 = torch.randn(3 * C, device="cuda"), torch.randn(3 * C, device="cuda")
    st = ops.rowstats128(x)
    ang = torch.rand(34, 16, device="cuda") * 3.0
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(BF16).contiguous(), emb.sin().to(BF16).contiguous()
    qn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)

    def fc1():
        return (ops.linear(x, w1, bias=b4, act=ops.ACT_GELU, ln_fold=ops.ln_fold(stats_in=st, eps=1e-5, colsum=c4)),)

    def qkv():
        return ops.linear_vit_qkv(x, wq, b3, 1029, 16, qn=qn, kn=qn, cos=cos, sin=sin, tokens_per_frame=1029, patch_start=5, Wp=32, eps=1e-5,
                                  ln_fold=ops.ln_fold(stats_in=st, eps=1e-5, colsum=c3))
    try:
        for fn in (fc1, qkv):
            ops.gemm_force_config(20)
            ref = [t.clone() for t in fn()]
            assert all(torch.isfinite(t.float()).all() for t in ref)
            ops.gemm_force_config(24)
            for it in range(40):
                got = fn()
                for a, b in zip(got, ref):
                    assert torch.equal(a, b), (fn.__name__, it, float((a.float() - b.float()).abs().max()))
    finally:
        ops.gemm_force_config(-3)


@pytest.mark.parametrize("cfg", [20, 21, 22, 24, 25, 30, 103, 105, 106, 107, "fp8", "wt"])
def test_qwen_layer_is_repeatable_under_every_gemm_kernel(ops, cfg):
    """The text model's launches under every NT kernel the tuner may pick (as tests/test_vggt_gpu.py does for the tower), every schedule
    of the k-major (weight-gradient / dgrad) kernels (103 / 105: gemm3.hip; 106 / 107: the k-major 8-phase kernel without / with the
    last-round split), the e4m3 projections ("fp8") and the W^T dgrad route ("wt"): two Qwen3-4B-width layers, 1536 token rows (grids
    that over-subscribe two workgroups per CU on the wide outputs), forward + backward four times from the same inputs - hidden states,
    d(inputs_embeds), projection and norm weight gradients must come out bit-identical, the scalar loss (an f32-atomic sum) within 1e-6, and everything within bf16 rounding of the plain cfg 20 run (e4m3: a sanity bound)."""
    from vggt_qwen3_amd.qwen3 import Qwen3Config, Qwen3ForCausalLM
    c = Qwen3Config.qwen3_4b(); c.num_hidden_layers = 2; c.vocab_size = 2048
    tm = Qwen3ForCausalLM(c, device="cuda", seed=4)
    mode = cfg if isinstance(cfg, str) else None
    cfg = -3 if mode else cfg
    g = torch.Generator().manual_seed(8)
    B, L = 8, 192
    emb = (torch.randn(B, L, c.hidden_size, generator=g) * 0.5).to(BF16).cuda()
    mask = torch.ones(B, L, dtype=torch.long); mask[1, 100:] = 0; mask[5, 37:] = 0
    labels = torch.full((B, L), -100, dtype=torch.long)
    labels[:, 20:36] = torch.randint(0, 2048, (B, 16), generator=g)
    mask, labels = mask.cuda(), labels.cuda()

    def run():
        h, saved = tm.forward_hidden(emb, mask, save=True)
        loss, head = tm.loss_head(h, labels, save=True, L=saved["L"])
        dh = tm.backward_loss_head(head, B * saved["L"], 1.0, accumulate=False)
        d_emb = tm.backward_hidden(saved, dh, accumulate=False)
        torch.cuda.synchronize()
        return [h.clone(), d_emb.clone(), tm._g["l0.qkv"].clone(), tm._g["l1.gu"].clone(), tm._g["l1.down"].clone(), tm._g["l0.ln2"].clone(),
                tm._g["l0.qn"].clone(), tm._g["l1.kn"].clone(), tm._g["embed"][:2048].clone(), tm._g["norm"].clone()], float(loss)
    try:
        ops.gemm_force_config(20)
        base, lb = run()
        ops.gemm_force_config(-3)
        ops.gemm_force_config(cfg)
        if mode == "fp8":
            tm.enable_fp8_forward(True)
        elif mode == "wt":
            tm.enable_dgrad_transposes(True)
        first, l1 = run()
        for it in range(3):
            again, l2 = run()
            # forward AND backward: bit for bit. (Until round 5 the lm_head's input-gradient product met its K slices through f32 atomics:
            # a 1e-7 difference there flipped a bf16 rounding now and then, every flipped element moved a whole row of the next GEMM's output
            # by a fraction of an ulp, and two layers further down d(inputs_embeds) differed by 2-3e-3 from run to run - bf16-ulp noise on
            # every gradient. It is a batched product into slabs summed in slice order now: qwen3.py: backward_loss_head.)
            for k, (a, b) in enumerate(zip(again, first)):
                assert torch.equal(a, b), (cfg, mode, it, k, float((a.float() - b.float()).abs().max()))
            assert abs(l2 - l1) <= 1e-6 * abs(l1)
    finally:
        ops.gemm_force_config(-3)
    tol = 0.25 if mode == "fp8" else 2e-2        # (e4m3 against bf16 arithmetic: only a sanity bound here; its parity tests are tests/test_fp8_gpu.py)
    for k, (a, b) in enumerate(zip(first, base)):
        assert torch.isfinite(a.float()).all()
        assert ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item() < tol, (cfg, mode, k)
    assert abs(l1 - lb) <= (5e-2 if mode == "fp8" else 2e-3) * abs(lb)
    assert not ops.gemm_split_gave_up()


def test_attention_kernels_are_repeatable(ops):
    """Same inputs, same bits, launch after launch, on grids larger than the chip: the VGGT flash kernel at 8232 keys (VAR 1: QK of keys
    32-63 under the exponentials of keys 0-31; XCD-aware placement) and at 1029 keys with the ragged-row workgroup, its leading-rows form,
    and the Perceiver's one-launch cross-attention (dropout on: the mask is a pure function of seed / offset / element index)."""
    g = torch.Generator().manual_seed(12)
    for G, NH, N, nq in ((2, 16, 8232, None), (40, 16, 1029, None), (40, 16, 1029, 128)):
        Q = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
        K = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
        V = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
        ref = ops.flash_attn(Q, K, V, q_rows=nq).clone()
        assert torch.isfinite(ref.float()).all()
        for it in range(6):
            assert torch.equal(ops.flash_attn(Q, K, V, q_rows=nq), ref), (G, N, nq, it)
    B, H, N, T, hd = 48, 8, 128, 128, 512
    q = torch.randn(B * N, H * hd, generator=g).to(BF16).cuda()
    kv = torch.randn(B * T, 2 * H * hd, generator=g).to(BF16).cuda()
    ref = ops.perceiver_xattn(q, kv, B, H, N, T, hd, 128, 0.1, 1234, 77).clone()
    for it in range(6):
        assert torch.equal(ops.perceiver_xattn(q, kv, B, H, N, T, hd, 128, 0.1, 1234, 77), ref), it
