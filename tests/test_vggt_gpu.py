"""VGGT aggregator: HIP path vs this repo's CPU restatement (oracle/vggt.py; parity with the real package is
unpinned - the reference does not vendor it). Also the flash-attention kernel alone vs torch SDPA in fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("G,NH,N", [(2, 2, 21), (1, 4, 64), (3, 2, 138), (2, 16, 1029), (1, 2, 2058), (36, 16, 1029), (40, 16, 541)])
def test_flash_attention_vs_sdpa(G, NH, N):
    """(N = 4 x 256 + 5: the ragged rows run as one more workgroup per pair with the keys split over its waves; 40 x 16 pairs with
    N = 2 x 256 + 29: as a one-wave launch on a side stream)"""
    from vggt_qwen3_amd import ops
    g = torch.Generator().manual_seed(N)
    Q = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
    K = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
    V = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
    out = ops.flash_attn(Q, K, V).view(G, N, NH, 64).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float())
    e = relerr(out, ref)
    assert e < 1e-2, f"flash attention rel err {e}"


@pytest.mark.parametrize("G,NH,N", [(32, 16, 1029), (32, 16, 272), (64, 8, 513), (32, 16, 779)])
def test_flash_attention_ragged_tail_rows(G, NH, N):
    """The 1 .. 16 query rows past the last full 256-row block (flash_tail_body: one more workgroup per pair, keys split over its four
    waves, partial (max, sum, O) merged through LDS) checked ON THEIR OWN against SDPA - in the whole-tensor error they are 5 rows of 1029 - with one dominant key per
    wave's share of the key tiles for some of them (every wave's running maximum moves, and the merge weighs four different maxima), a
    tail of 16 (272), of 1 (513) and of 11 (779: a last 32-key tile with 11 keys)."""
    from vggt_qwen3_amd import ops
    g = torch.Generator().manual_seed(N)
    Q = torch.randn(G, NH, N, 64, generator=g)
    K = torch.randn(G, NH, N, 64, generator=g)
    V = torch.randn(G, NH, N, 64, generator=g)
    t0 = N // 256 * 256
    for j, f in ((3, 2.0), (40, 3.0), (70, 4.0), (100, 2.5), (N - 2, 3.5)):       # keys in tiles 0, 1, 2, 3 (waves 0-3) and the last tile
        K[0, 0, j] = Q[0, 0, N - 1] * f
        K[1, 3, j] = Q[1, 3, t0] * f
    Q, K, V = Q.to(BF16).cuda(), K.to(BF16).cuda(), V.to(BF16).cuda()
    out = ops.flash_attn(Q, K, V).view(G, N, NH, 64).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(Q[:, :, t0:].float(), K.float(), V.float())
    assert relerr(out[:, :, t0:], ref) < 1e-2
    assert (out[:, :, t0:].float() - ref).abs().max().item() < 0.05
    refm = torch.nn.functional.scaled_dot_product_attention(Q[:, :, :t0].float(), K.float(), V.float())
    assert relerr(out[:, :, :t0], refm) < 1e-2


@pytest.mark.parametrize("G,NH,N,nq", [(2, 2, 138, 40), (3, 16, 1029, 128), (1, 4, 2058, 128), (2, 4, 1029, 600), (1, 2, 300, 300)])
def test_flash_attention_leading_query_rows(G, NH, N, nq):
    """vq3_flash_attn_fwd_rows: the first nq queries of every group against all N keys -> [G * nq, NH * 64]."""
    from vggt_qwen3_amd import ops
    g = torch.Generator().manual_seed(N + nq)
    Q = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
    K = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
    V = torch.randn(G, NH, N, 64, generator=g).to(BF16).cuda()
    out = ops.flash_attn(Q, K, V, q_rows=nq)
    assert out.shape == (G * nq, NH * 64)
    ref = torch.nn.functional.scaled_dot_product_attention(Q[:, :, :nq].float(), K.float(), V.float())
    assert relerr(out.view(G, nq, NH, 64).transpose(1, 2), ref) < 1e-2
    with pytest.raises(RuntimeError):
        ops.flash_attn(Q, K, V, q_rows=N + 1)


@pytest.mark.parametrize("N", [300, 2200])
def test_flash_attention_spiked_max(N):
    """Forces the running-max rescale: one key per tile dominates for some queries (guide rule 26). N = 300 runs the kernel variant with the
    row sums on the matrix pipe; N = 2200 (>= 2048) the one that multiplies keys 32-63 while keys 0-31 are exponentiated: there the spikes
    sit in both halves of their tiles, one of them (x 10: 2^115 above the stale maximum) is large enough for the overflow guard - the first half's scores are multiplied
    again - and one lies in the key tail's tile."""
    from vggt_qwen3_amd import ops
    g = torch.Generator().manual_seed(3)
    G, NH = 1, 1
    Q = torch.randn(G, NH, N, 64, generator=g)
    K = torch.randn(G, NH, N, 64, generator=g)
    V = torch.randn(G, NH, N, 64, generator=g)
    spikes = [(70, 5, 3.0), (150, 5, 3.0), (290, 40, 3.0), (10, 100, 3.0)]
    if N > 2048:
        spikes += [(64 * 9 + 17, 7, 3.0), (64 * 9 + 49, 7, 4.0), (64 * 20 + 40, 300, 10.0), (64 * 21 + 3, 300, 3.0), (N - 3, 1500, 5.0)]
    for j, q, f in spikes:
        K[0, 0, j] = Q[0, 0, q] * f
    Q, K, V = Q.to(BF16).cuda(), K.to(BF16).cuda(), V.to(BF16).cuda()
    out = ops.flash_attn(Q, K, V).view(G, N, NH, 64).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float())
    assert relerr(out, ref) < 1e-2
    assert (out.float() - ref).abs().max().item() < 0.05


@pytest.mark.parametrize("G,NH,N,nq", [(2, 2, 21, None), (3, 2, 138, None), (2, 16, 1029, None), (1, 2, 2058, None), (36, 16, 1029, None),
                                       (1, 16, 8232, None), (3, 16, 1029, 128), (2, 4, 1029, 600)])
def test_flash_attention_with_a_score_bound_vs_sdpa(G, NH, N, nq):
    """vq3_flash_attn_fwd_bounded (the kernels without a running maximum) against SDPA in fp32 - on unit-normal operands, and on operands
    scaled so that the scores fill the promised range: some rows whose largest score sits near +bound (2^+72 and more as the exponent's argument),
    some near -bound with every other score below (all weights tiny: 2^-65 and less), a dominant key in the tail tile."""
    from vggt_qwen3_amd import ops
    g = torch.Generator().manual_seed(N)
    Q = torch.randn(G, NH, N, 64, generator=g)
    K = torch.randn(G, NH, N, 64, generator=g)
    V = torch.randn(G, NH, N, 64, generator=g)
    bound = 60.0                                                   # natural units; the library's limit is 90 / log2(e) = 62.4
    rows = N if nq is None else nq

    def run(Q, K, V, bound):
        Qd, Kd, Vd = Q.to(BF16).cuda(), K.to(BF16).cuda(), V.to(BF16).cuda()
        out = ops.flash_attn(Qd, Kd, Vd, q_rows=nq, score_bound=bound).view(G, rows, NH, 64).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(Qd[:, :, :rows].float(), Kd.float(), Vd.float())
        assert torch.isfinite(out).all()
        return out, ref
    out, ref = run(Q, K, V, bound)
    assert relerr(out, ref) < 1e-2
    # scores near the edge of the promise: |q| = |k| = sqrt(8 * 50) for the chosen pairs -> q . k / 8 = +-50 (picks that share a head
    # interact through the shifted keys: up to ~ +-58)
    Q2, K2 = Q.clone(), K.clone()
    unit = lambda v: v / v.norm()
    a = (8 * 50.0) ** 0.5
    picks = [(0, 0, 3 % rows, 5 % N, 1.0), (0, 0, 7 % rows, N - 1, 1.0), (G - 1, NH - 1, rows - 1, 2 % N, -1.0), (G - 1, 0, 11 % rows, 9 % N, -1.0)]
    for gi, hi, qi, ki, sign in picks:
        d = unit(torch.randn(64, generator=g))
        Q2[gi, hi, qi] = a * d
        if sign > 0:
            K2[gi, hi, ki] = a * d                                  # one key at +50, the others ~ N(0, 50 / 8)
        else:
            K2[gi, hi] = K2[gi, hi] - (K2[gi, hi] @ d)[:, None] * d[None, :] - 0.9 * a * d[None, :]      # every key at -0.9 * 50 along d ...
            K2[gi, hi, ki] = -a * d                                 # ... one at -50: all scores of this query in [-50, -45]
    q2, k2 = Q2.to(BF16).float(), K2.to(BF16).float()
    assert (torch.einsum("ghqd,ghkd->ghqk", q2[:, :, :rows], k2).abs().max() / 8).item() <= bound
    out2, ref2 = run(Q2, K2, V, bound)
    assert relerr(out2, ref2) < 1e-2
    assert (out2.float().cpu() - ref2.cpu()).abs().max().item() < 0.05
    for gi, hi, qi, ki, sign in picks:                              # the edge rows on their own
        assert relerr(out2[gi, hi, qi], ref2[gi, hi, qi]) < 2e-2, (gi, hi, qi, sign)


def test_flash_attention_score_bound_above_the_limit_takes_the_general_kernels():
    """A promise the no-maximum kernels cannot use (bound > 62.4) must run the general ones: scores of +-200 (2^+-288 as exponent arguments:
    overflow without a running maximum) stay exact. A negative bound means 'no promise'."""
    from vggt_qwen3_amd import ops
    g = torch.Generator().manual_seed(5)
    G, NH, N = 1, 2, 600
    Q = torch.randn(G, NH, N, 64, generator=g)
    K = torch.randn(G, NH, N, 64, generator=g)
    V = torch.randn(G, NH, N, 64, generator=g)
    d = torch.randn(64, generator=g); d = d / d.norm()
    Q[0, 0, 5] = 40.0 * d; K[0, 0, 300] = 40.0 * d; K[0, 0, 301] = -40.0 * d              # q . k / 8 = +-200
    Qd, Kd, Vd = Q.to(BF16).cuda(), K.to(BF16).cuda(), V.to(BF16).cuda()
    ref = torch.nn.functional.scaled_dot_product_attention(Qd.float(), Kd.float(), Vd.float())
    for bound in (250.0, -1.0, float("inf")):
        out = ops.flash_attn(Qd, Kd, Vd, score_bound=bound).view(G, N, NH, 64).transpose(1, 2)
        assert torch.isfinite(out).all() and relerr(out, ref) < 1e-2, bound


def test_vggt_score_bound_from_the_norm_weights_holds():
    """The bound the aggregator promises for its frame / global attentions (vggt.py: from the q / k LayerNorm weights) against the scores
    the fused q|k|v epilogue actually produces, with non-trivial norm weights and biases; the DINOv2 stage (no q / k norm) promises nothing."""
    from vggt_qwen3_amd import ops
    from vggt_qwen3_amd.vggt import VGGT
    model = VGGT(img_size=70, patch_size=14, embed_dim=128, depth=2, dino_depth=1, device="cuda", seed=3)
    agg = model.aggregator
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in agg.named_tensors().items():
            if "q_norm" in n or "k_norm" in n:
                p.copy_((1.5 * torch.randn(p.shape, generator=g) + (1.0 if n.endswith("weight") else 0.0)).to(p.dtype))
    agg.invalidate_compute_copies()
    cc = agg._prepare()
    assert all("score_bound" not in w for w in cc["dino"])
    seen = []
    real = ops.flash_attn

    def spy(Q, K, V, out=None, q_rows=None, score_bound=None):
        if score_bound is not None:
            s = torch.einsum("ghqd,ghkd->ghqk", Q.float(), K.float()).abs().max().item() / 8.0
            seen.append((s, score_bound))
        return real(Q, K, V, out=out, q_rows=q_rows, score_bound=score_bound)
    ops.flash_attn = spy
    try:
        agg(torch.rand(1, 2, 3, 70, 70, generator=g).cuda())
    finally:
        ops.flash_attn = real
    assert len(seen) == 4                                           # 2 frame + 2 global blocks
    for s, b in seen:
        assert s <= b, (s, b)
        assert b < 40 * s                                           # (and not vacuous)


@pytest.mark.parametrize("H,W,S,B,C", [(56, 56, 3, 2, 128), (70, 84, 2, 1, 256), (112, 112, 2, 1, 128)])
def test_aggregator_vs_cpu_restatement(H, W, S, B, C):
    from oracle import vggt as ov
    from vggt_qwen3_amd.vggt import VGGT
    model = VGGT(img_size=70, patch_size=14, embed_dim=C, depth=2, dino_depth=2, device="cuda", seed=11)
    agg = model.aggregator
    # make LayerScale / special tokens non-trivial so every term matters
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, t in agg.named_tensors().items():
            if n.endswith("gamma"):
                t.copy_((0.5 + 0.5 * torch.rand(t.shape, generator=g)).to(BF16))
            elif n.endswith("bias") or n in ("camera_token", "register_token", "patch_embed.cls_token",
                                             "patch_embed.register_tokens"):
                t.copy_((0.1 * torch.randn(t.shape, generator=g)).to(BF16))
            elif "norm" in n and n.endswith("weight"):
                t.copy_((1.0 + 0.2 * torch.randn(t.shape, generator=g)).to(BF16))
            elif t.dim() >= 2:
                t.copy_((0.05 * torch.randn(t.shape, generator=g)).to(BF16))
    agg._cc = None; agg._pos_cache.clear()
    images = torch.rand(B, S, 3, H, W, generator=g)
    outs, ps = agg(images.cuda(), return_all=True)
    assert ps == 5 and len(outs) == 2
    P = 5 + (H // 14) * (W // 14)
    assert outs[-1].shape == (B, S, P, 2 * C)
    sd = {n: t.detach().float().cpu() for n, t in agg.named_tensors().items()}
    ref32 = ov.aggregator(images, sd, num_heads=C // 64, depth=2, dino_depth=2, dtype=torch.float32)
    ref16 = ov.aggregator(images, sd, num_heads=C // 64, depth=2, dino_depth=2, dtype=torch.bfloat16)
    for i in range(2):
        e32 = relerr(outs[i], ref32[i])
        noise = relerr(ref16[i], ref32[i])       # what bf16 evaluation itself costs on the CPU restatement
        assert e32 < max(2e-2, 3 * noise), f"iterate {i}: HIP vs fp32 restatement {e32}, bf16 CPU noise {noise}"


def test_aggregator_row_split_chain_equals_the_unsplit_blocks(monkeypatch):
    """Aggregator._block with the proj -> fc1 -> fc2 chain split by rows (whole rounds of 16 384 rows on the caller's stream, the rows
    behind them as a chain of their own on a second stream, joined before the next block's q|k|v) against the unsplit blocks: the three
    GEMMs are row-wise, so both iterates agree to GEMM-configuration noise. 64 frames x 261 tokens = 16 704 rows: a 320-row tail."""
    from vggt_qwen3_amd.vggt import Aggregator
    C = 128
    agg = Aggregator(img_size=224, patch_size=14, embed_dim=C, depth=2, num_heads=C // 64, dino_depth=2, device="cuda", seed=3)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, t in agg.named_tensors().items():
            if n.endswith("gamma"):
                t.copy_((0.5 + 0.5 * torch.rand(t.shape, generator=g)).to(BF16))
            elif t.dim() >= 2:
                t.copy_((0.05 * torch.randn(t.shape, generator=g)).to(BF16))
    agg._cc = None; agg._pos_cache.clear()
    images = torch.rand(4, 16, 3, 224, 224, generator=g).cuda()
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("VQ3_VGGT_ROW_SPLIT", mode)
        assert (agg._split_rows(64 * 261) == 16384) == (mode == "2")
        o, _ = agg(images, return_all=True)
        torch.cuda.synchronize()
        outs[mode] = [t.clone() for t in o]
    for a, b in zip(outs["2"], outs["0"]):
        assert relerr(a, b) < 2e-3
        assert relerr(a[-1, -1], b[-1, -1]) < 2e-3           # the last frame holds the tail rows


@pytest.mark.parametrize("use_norm,use_rope", [(True, True), (False, False), (True, False), (False, True)])
def test_vit_qkprep_vs_fp32_reference(monkeypatch, use_norm, use_rope):
    """Per-head LayerNorm(64) + 2-D rotate-half RoPE + head-major split, both lane layouts (4 features per lane; the
    2-feature form forced through VQ3_VIT_QKPREP_VEC2=1) against a plain fp32 restatement of the same arithmetic."""
    from vggt_qwen3_amd import ops
    torch.manual_seed(0)
    G, P, NH, Wp, ps = 2, 21, 4, 4, 5                  # 21 tokens per frame: 5 special + 4x4 patches
    N = P                                               # frame attention: one frame per group
    T = G * N
    qkv = torch.randn(T, 3 * NH * 64, device="cuda").to(torch.bfloat16)
    qn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    kn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    ang = torch.rand(Wp + 1, 16, device="cuda") * 3.0
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(torch.bfloat16).contiguous(), emb.sin().to(torch.bfloat16).contiguous()
    kw = dict(qn=qn if use_norm else None, kn=kn if use_norm else None, cos=cos if use_rope else None,
              sin=sin if use_rope else None, tokens_per_frame=P, patch_start=ps, Wp=Wp, eps=1e-5)
    monkeypatch.delenv("VQ3_VIT_QKPREP_VEC2", raising=False)
    Q4, K4, V4 = ops.vit_qkprep(qkv, N, NH, **kw)
    monkeypatch.setenv("VQ3_VIT_QKPREP_VEC2", "1")
    Q2, K2, V2 = ops.vit_qkprep(qkv, N, NH, **kw)
    monkeypatch.delenv("VQ3_VIT_QKPREP_VEC2", raising=False)
    assert torch.equal(V4, V2)
    for a, b in ((Q4, Q2), (K4, K2)):
        assert (a.float() - b.float()).abs().max() <= 2 ** -6 * b.float().abs().max()     # LayerNorm sums associate differently
    # fp32 restatement
    x = qkv.float().view(G, N, 3, NH, 64).permute(2, 0, 3, 1, 4)          # [3, G, NH, N, 64]
    def prep(t, nw):
        if use_norm:
            t = torch.nn.functional.layer_norm(t, (64,), nw[0], nw[1], 1e-5)
        t = t.to(torch.bfloat16).float()
        if use_rope:
            tp = torch.arange(N, device="cuda") % P
            py = torch.where(tp >= ps, (tp - ps) // Wp + 1, torch.zeros_like(tp))
            px = torch.where(tp >= ps, (tp - ps) % Wp + 1, torch.zeros_like(tp))
            out = []
            for blk, pos in ((t[..., :32], py), (t[..., 32:], px)):
                c, s = cos.float()[pos], sin.float()[pos]                   # [N, 32]
                rot = torch.cat([-blk[..., 16:], blk[..., :16]], -1)
                out.append(blk * c + rot * s)
            t = torch.cat(out, -1)
        return t
    for got, ref in ((Q4, prep(x[0], qn)), (K4, prep(x[1], kn)), (V4, x[2])):
        assert ((got.float() - ref).norm() / ref.norm()).item() < 6e-3


def test_dino_backbone_vs_transformers_golden():
    """HIP DINOv2-with-registers stage (im2col + patch GEMM + position table + 2 ViT blocks + final LayerNorm) against the
    output of transformers' Dinov2WithRegistersModel on the same weights (tests/golden/dinov2_tiny.npz): bf16 tolerance."""
    from tests.golden_io import load, meta, weights
    from vggt_qwen3_amd.vggt import VGGT
    z = load("dinov2_tiny.npz")
    m = meta(z)
    model = VGGT(img_size=56, patch_size=m["patch"], embed_dim=m["embed_dim"], depth=1, dino_depth=m["depth"], device="cuda", seed=3)
    agg = model.aggregator
    missing = agg.load_named({k: v for k, v in weights(z).items()})
    assert all(not n.startswith("patch_embed.") for n in missing), [n for n in missing if n.startswith("patch_embed.")][:4]
    for name in ("native", "interp"):
        images = torch.from_numpy(z[f"{name}:images"]).cuda()[:, None]          # [N, 1, 3, H, W]: one view per sample
        got = agg.dino_tokens(images)
        ref = torch.from_numpy(z[f"{name}:tokens"])
        assert got.shape == ref.shape
        assert relerr(got, ref) < 2e-2, (name, relerr(got, ref))


@pytest.mark.parametrize("use_norm,use_rope", [(True, True), (False, False)])
@pytest.mark.parametrize("cfg", [-3, 20, 24, 11, 7])
def test_fused_qkv_epilogue_equals_linear_plus_vit_qkprep(use_norm, use_rope, cfg):
    """vq3_gemm_vit_qkv (head split + q/k LayerNorm + 2-D RoPE in the GEMM epilogue, qkv never materialised) against the two-launch
    form it replaces - vq3_gemm_bf16_nt followed by vq3_vit_qkprep - on the same inputs, for every tile configuration that can be
    chosen, with M not a multiple of any tile height (3 frames of 5 + 6x7 tokens, two groups)."""
    from vggt_qwen3_amd import ops
    torch.manual_seed(1)
    NH, C, Wp, ps = 4, 256, 7, 5
    P = ps + 6 * Wp                                      # 47 tokens per frame
    G, N = 2, 3 * P                                      # global attention over 3 frames
    T = G * N
    x = torch.randn(T, C, device="cuda").to(BF16)
    w = (torch.randn(3 * C, C, device="cuda") * 0.06).to(BF16)
    bias = torch.randn(3 * C, device="cuda") * 0.1
    qn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    kn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    ang = torch.rand(8, 16, device="cuda") * 3.0
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(BF16).contiguous(), emb.sin().to(BF16).contiguous()
    kw = dict(qn=qn if use_norm else None, kn=kn if use_norm else None, cos=cos if use_rope else None,
              sin=sin if use_rope else None, tokens_per_frame=P, patch_start=ps, Wp=Wp, eps=1e-5)
    try:
        ops.gemm_force_config(cfg)
        qkv = ops.linear(x, w, bias=bias)
        ref = ops.vit_qkprep(qkv, N, NH, **kw)
        got = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
    finally:
        ops.gemm_force_config(-3)
    for name, a, b in zip("QKV", got, ref):
        assert a.shape == b.shape == (G, NH, N, 64)
        # same inputs and rounding points; only the LayerNorm sums associate differently (one bf16 ulp at most)
        assert (a.float() - b.float()).abs().max() <= 2 ** -6 * b.float().abs().max(), name
        assert ((a.float() - b.float()).norm() / b.float().norm()).item() < 2e-3, name
    assert torch.equal(got[2], ref[2])                   # V is a pure copy


def test_fused_qkv_epilogue_with_row_tail_launch():
    """cfg 30 (whole rounds of 256 x 256 tiles + a small-tile launch for the remaining rows) under the fused q|k|v epilogue: the tail
    launch starts at token 16 384 of 16 632 (its rows' group / token indices must carry that offset). 7 groups of 2376 tokens
    (= 8 frames of 5 + 292 patch tokens; 65 x 12 = 780 tiles = 3 rounds of 256 + 12)."""
    from vggt_qwen3_amd import ops
    torch.manual_seed(2)
    NH, C, Wp, ps = 16, 128, 4, 5
    P = ps + 73 * Wp                                     # 297 tokens per frame
    G, N = 7, 8 * P
    T = G * N
    assert T == 16632
    x = torch.randn(T, C, device="cuda").to(BF16)
    w = (torch.randn(3 * NH * 64, C, device="cuda") * 0.06).to(BF16)
    bias = torch.randn(3 * NH * 64, device="cuda") * 0.1
    qn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    kn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    ang = torch.rand(75, 16, device="cuda") * 3.0
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(BF16).contiguous(), emb.sin().to(BF16).contiguous()
    kw = dict(qn=qn, kn=kn, cos=cos, sin=sin, tokens_per_frame=P, patch_start=ps, Wp=Wp, eps=1e-5)
    try:
        ops.gemm_force_config(20)
        ref = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
        ops.gemm_force_config(30)
        got = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
        ops.gemm_force_config(24)                        # two workgroups per CU, 256 x 128 tiles: 65 x 12 tiles
        got7 = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
    finally:
        ops.gemm_force_config(-3)
    for name, a, b in zip("QKV", got, ref):
        af, bf = a.float().reshape(-1, NH, N, 64), b.float().reshape(-1, NH, N, 64)
        assert (af - bf).abs().max() <= 2 ** -6 * bf.abs().max(), name
        assert ((af - bf).norm() / bf.norm()).item() < 2e-3, name
        # the tail rows are the last 248 tokens of the last group
        assert ((af[-1, :, -248:] - bf[-1, :, -248:]).norm() / bf[-1, :, -248:].norm()).item() < 2e-3, name
    for name, a, b in zip("QKV", got7, ref):
        assert torch.equal(a, b), name                   # same arithmetic, same rounding points: bit-identical


def test_fused_qkv_epilogue_behind_a_split_last_round():
    """cfg 25 under the fused q|k|v epilogue (round 4: the reducer of a K-split tile runs the head split + LayerNorm + RoPE on the summed
    tile): 22 x 12 = 264 tiles of 256 x 256 = one round of 256 CUs + 8 tiles cut into two K halves (K = 1024). Against the unsplit kernel:
    the same values up to the order of the f32 sum."""
    from vggt_qwen3_amd import ops
    torch.manual_seed(3)
    NH, C, Wp, ps = 16, 1024, 7, 5
    P = ps + 48 * Wp                                     # 341 tokens per frame
    G, N = 2, 8 * P
    T = G * N
    assert T == 5456 and ops.gemm_split_plan(T, 3 * NH * 64, C) == (256, 8, 2)
    x = torch.randn(T, C, device="cuda").to(BF16)
    w = (torch.randn(3 * NH * 64, C, device="cuda") * 0.03).to(BF16)
    bias = torch.randn(3 * NH * 64, device="cuda") * 0.1
    qn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    kn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
    ang = torch.rand(50, 16, device="cuda") * 3.0
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(BF16).contiguous(), emb.sin().to(BF16).contiguous()
    kw = dict(qn=qn, kn=kn, cos=cos, sin=sin, tokens_per_frame=P, patch_start=ps, Wp=Wp, eps=1e-5)
    try:
        ops.gemm_force_config(20)
        ref = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
        ops.gemm_force_config(25)
        got = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
        got2 = ops.linear_vit_qkv(x, w, bias, N, NH, **kw)
    finally:
        ops.gemm_force_config(-3)
    assert not ops.gemm_split_gave_up()
    for name, a, b, c in zip("QKV", got, ref, got2):
        af, bf = a.float(), b.float()
        assert (af - bf).abs().max() <= 2 ** -6 * bf.abs().max(), name
        assert ((af - bf).norm() / bf.norm()).item() < 2e-3, name
        assert torch.equal(a, c), name                   # the partial sums meet in a fixed order: repeatable
    # the split tiles are the last 8 of the launch's order; every row block was written
    assert all(torch.isfinite(t.float()).all() for t in got)


def test_aggregator_is_batch_invariant():
    """precompute_vision() runs the tower once over several micro-batches' images: every sample must come out as in its own
    micro-batch (frame attention per (sample, view), global attention per sample, GEMM rows independent of M - the tile configuration
    may change with M, the k order of a row's dot products does not)."""
    from vggt_qwen3_amd.vggt import VGGT
    model = VGGT(img_size=70, patch_size=14, embed_dim=128, depth=2, dino_depth=2, device="cuda", seed=3)
    agg = model.aggregator
    g = torch.Generator().manual_seed(9)
    a = torch.rand(2, 3, 3, 70, 84, generator=g).cuda()
    b = torch.rand(1, 3, 3, 70, 84, generator=g).cuda()
    ya, yb = agg(a)[0][-1], agg(b)[0][-1]
    yab = agg(torch.cat([a, b], dim=0))[0][-1]
    assert yab.shape[0] == 3
    assert relerr(yab[:2], ya) < 1e-3 and relerr(yab[2:], yb) < 1e-3
    assert (yab[:2].float() - ya.float()).abs().max() <= 2 ** -6 * ya.float().abs().max()


@pytest.mark.parametrize("n", [7, 40, 200])
def test_aggregator_forward_head_equals_the_slice_the_reference_takes(n, monkeypatch):
    """Aggregator.forward_head(images, n) = forward(images)[0][-1].reshape(B, S*P, 2C)[:, :n] (vggt_qwen3_vlm.py:144-156); the last global
    block runs attention / proj / MLP on the kept rows only. n = 40 crosses from view 0 into view 1 (P = 35), 200 > S*P = 105 keeps all."""
    from vggt_qwen3_amd.vggt import VGGT
    for fold in ("1", "0"):
        monkeypatch.setenv("VQ3_VGGT_LN_FOLD", fold)
        model = VGGT(img_size=70, patch_size=14, embed_dim=128, depth=2, dino_depth=2, device="cuda", seed=3)
        agg = model.aggregator
        g = torch.Generator().manual_seed(19)
        x = torch.rand(2, 3, 3, 70, 84, generator=g).cuda()
        full = agg(x)[0][-1]
        B, S, P, C2 = full.shape
        want = full.reshape(B, S * P, C2)[:, :n]
        got = agg.forward_head(x, n)
        assert got.shape == want.shape
        assert relerr(got, want) < 1e-3
        assert (got.float() - want.float()).abs().max() <= 2 ** -6 * want.float().abs().max()


@pytest.mark.parametrize("cfg", [20, 21, 22, 24, 25, 30, -3])
def test_tower_forward_is_repeatable_under_every_gemm_kernel(cfg):
    """Round 5 (the gemm7 race): every GEMM kernel the tuner may pick for the tower's launches - 256 x 256 (20), its two-phase persistent
    forms (21 / 22), two workgroups per CU (24), the K-split last round (25), whole rounds + row tail (30), and the tuner's own choice (-3) -
    must give the SAME BITS for the same input, launch after launch (none of them uses atomics; every one sums K in a fixed order), on a
    grid that over-subscribes the chip (7 frames x 1029 tokens = 7203 rows: 29 x 32 tiles of 256 x 128), and all of them agree within
    bf16 rounding. Full width (C = 1024, 16 heads), one DINOv2 + one frame + one global block, LayerNorm folds on."""
    from vggt_qwen3_amd import ops
    from vggt_qwen3_amd.vggt import VGGT
    model = VGGT(img_size=518, patch_size=14, embed_dim=1024, depth=1, dino_depth=1, device="cuda", seed=5)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(7, 1, 3, 448, 448, generator=g).cuda()
    agg = model.aggregator
    ops.gemm_force_config(20)
    try:
        base = agg(img)[0][-1].clone()
        ops.gemm_force_config(cfg)
        first = agg(img)[0][-1].clone()
        for it in range(10):
            again = agg(img)[0][-1]
            assert torch.equal(again, first), (cfg, it, float((again.float() - first.float()).abs().max()))
    finally:
        ops.gemm_force_config(-3)
    assert torch.isfinite(first.float()).all()
    assert ((first.float() - base.float()).norm() / base.float().norm()).item() < 5e-3, cfg
    assert not ops.gemm_split_gave_up()


@pytest.mark.parametrize("N,nq", [(1029, None), (541, None), (8232, None), (1029, 128), (21, None)])
def test_flash_attention_does_not_read_past_the_last_row(N, nq):
    """K / V rows of the last, partly filled 64-key tile lie past N: for every (sample, head) pair but the last they are the NEXT pair's
    rows, for the last pair they are whatever follows the tensors in memory. They must never reach the result (the buffer descriptor's
    range check returns zeros for them, and their scores are masked): Q, K and V here are views at the front of larger allocations whose
    remainder is NaN, and the output must be finite and equal to the run on exact-size tensors."""
    from vggt_qwen3_amd import ops
    G, NH = 2, 16
    g = torch.Generator().manual_seed(N)
    n = G * NH * N * 64
    base = [torch.randn(n, generator=g).to(BF16).cuda() for _ in range(3)]
    want = ops.flash_attn(*[t.view(G, NH, N, 64) for t in base], q_rows=nq).clone()
    padded = []
    for t in base:
        big = torch.full((n + 64 * 64 * 4,), float("nan"), dtype=BF16, device="cuda")
        big[:n].copy_(t)
        padded.append(big[:n].view(G, NH, N, 64))
    got = ops.flash_attn(*padded, q_rows=nq)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want)
