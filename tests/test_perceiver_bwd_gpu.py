"""The "corrected" mode of SURVEY.md 3.1 (VisionLanguageConfig.train_projector): the Perceiver projector's hand-written backward
against torch autograd through the CPU oracle (oracle/perceiver.py restates src/models/projector_perceiver.py:30-82 and is pinned
by the reference's own output in tests/test_oracle_golden.py). Tolerance for gradients: 4e-2 relative L2 per tensor (bf16 GEMM
operands, f32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def test_layernorm_bwd_gelu_kernels_vs_autograd():
    from vggt_qwen3_amd import ops
    torch.manual_seed(0)
    for rows, cols, with_res in ((300, 4096, True), (77, 256, False), (1029, 1024, True)):
        x = torch.randn(rows, cols, device="cuda") * 2 + 0.3
        res = torch.randn(rows, cols, device="cuda") if with_res else None
        w = torch.rand(cols, device="cuda") + 0.5
        b = torch.randn(cols, device="cuda")
        dy = torch.randn(rows, cols, device="cuda")
        xr = ((x + res) if with_res else x).double().requires_grad_(True)
        wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
        F.layer_norm(xr, (cols,), wr, br, 1e-5).backward(dy.double())
        dw0 = torch.full((cols,), 2.0, device="cuda")
        dx, dw, db = ops.layernorm_bwd(dy, x, w, 1e-5, res=res, dw_out=dw0)
        assert relerr(dx, xr.grad) < 1e-4 and relerr(db, br.grad) < 1e-4
        assert relerr(dw, wr.grad + 2.0) < 1e-4 and dw.data_ptr() == dw0.data_ptr()        # accumulated into the given buffer
    z = (torch.randn(4096, 256, device="cuda") * 2).to(BF16)
    dh = torch.randn(4096, 256, device="cuda").to(BF16)
    zr = z.double().requires_grad_(True)
    hr = F.gelu(zr)
    hr.backward(dh.double())
    assert relerr(ops.gelu_bwd(dh, z), zr.grad) < 4e-3
    assert relerr(ops.gelu_fwd(z), hr) < 4e-3
    # the separate activation pass equals the GEMM epilogue's (same rounding points)
    a = (torch.randn(256, 128, device="cuda") * 0.5).to(BF16)
    wt = (torch.randn(256, 128, device="cuda") * 0.3).to(BF16)
    assert torch.equal(ops.gelu_fwd(ops.linear(a, wt)), ops.linear(a, wt, act=ops.ACT_GELU))
    m = torch.randn(513, 1000, device="cuda")
    acc = torch.ones(1000, device="cuda")
    assert relerr(ops.colsum_f32(m), m.double().sum(0)) < 1e-5
    assert relerr(ops.colsum_f32(m, out=acc, accumulate=True), m.double().sum(0) + 1.0) < 1e-5


def _build(cfg_kw, in_dim, out_dim, seed):
    from vggt_qwen3_amd.perceiver import PerceiverConfig, PerceiverProjector
    torch.manual_seed(seed)
    m = PerceiverProjector(PerceiverConfig(**cfg_kw), in_dim, out_dim).cuda()
    with torch.no_grad():      # non-trivial biases / norms: the reference's init leaves them at 0 / 1
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "norm" in n and n.endswith("weight"):
                p.uniform_(0.7, 1.3)
    return m


def _oracle_grads(m, tokens, d_out, heads, layers):
    from oracle import perceiver as operc
    sd = {k: v.detach().float().cpu().requires_grad_(True) for k, v in m.state_dict().items()}
    out = operc.projector(tokens.float().cpu(), sd, heads, layers)
    out.backward(d_out.float().cpu())
    return out.detach(), {k: v.grad for k, v in sd.items()}


@pytest.mark.parametrize("which", ["tiny", "fullwidth"])
def test_perceiver_backward_vs_oracle_autograd(which):
    """Every projector gradient (latents, in_proj, packed in_proj_weight / bias, out_proj, both MLP linears, both LayerNorms of every
    layer, out_proj) against torch autograd through the oracle: the tiny two-layer configuration and ONE layer at the production width
    (latent 4096, 8 heads x 512, FFN 16384, 128 latents x 128 context tokens)."""
    if which == "tiny":
        kw, in_dim, out_dim, B, T = dict(latent_dim=128, num_latents=16, num_heads=2, num_layers=2, ffn_dim=256, dropout=0.1), 64, 64, 3, 16
    else:
        kw, in_dim, out_dim, B, T = dict(latent_dim=4096, num_latents=128, num_heads=8, num_layers=1, ffn_dim=16384, dropout=0.1), 2048, 2560, 2, 128
    m = _build(kw, in_dim, out_dim, 3).eval()                 # eval: dropout off (the masked case is the next test)
    g = torch.Generator().manual_seed(5)
    tokens = torch.randn(B, T, in_dim, generator=g).cuda()
    d_out = torch.randn(B, kw["num_latents"], out_dim, generator=g).cuda() * 0.1
    out, ctx = m.forward_train(tokens)
    assert relerr(out, m(tokens)) < 1e-6                      # the saving forward IS the forward
    m.backward(ctx, d_out)
    ref_out, ref = _oracle_grads(m, tokens, d_out, kw["num_heads"], kw["num_layers"])
    assert relerr(out, ref_out) < 1e-2
    bad = {n: relerr(p.grad, ref[n]) for n, p in m.named_parameters() if relerr(p.grad, ref[n]) >= 4e-2}
    assert not bad, bad
    # gradients accumulate: a second backward of the same state doubles them
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    out2, ctx2 = m.forward_train(tokens)
    m.backward(ctx2, d_out)
    assert all(relerr(p.grad, 2 * g1[n]) < 1e-3 for n, p in m.named_parameters())


def test_perceiver_backward_with_dropout_masks():
    """Train mode: the four dropout sites per layer (projector_perceiver.py:33,37,42,46-49) are active in the saving forward and their
    masks are regenerated in the backward. Reference: the same layer arithmetic in torch with the masks the HIP kernel draws
    (read back by dropping a tensor of ones at the recorded offsets)."""
    from vggt_qwen3_amd import ops
    kw = dict(latent_dim=128, num_latents=16, num_heads=2, num_layers=2, ffn_dim=256, dropout=0.25)
    in_dim, out_dim, B, T = 64, 64, 2, 16
    m = _build(kw, in_dim, out_dim, 4).train()
    g = torch.Generator().manual_seed(6)
    tokens = torch.randn(B, T, in_dim, generator=g).cuda()
    d_out = torch.randn(B, 16, out_dim, generator=g).cuda() * 0.1
    out, ctx = m.forward_train(tokens)
    m.backward(ctx, d_out)
    D, N, Hh, hd, Tp = 128, 16, 2, 64, ctx["Tp"]

    def mask(shape, off):
        return ops.dropout_(torch.ones(shape, device="cuda"), 0.25, ctx["seed"], off).cpu()
    sd = {k: v.detach().float().cpu().requires_grad_(True) for k, v in m.state_dict().items()}
    tk = tokens.float().cpu()
    cx = F.linear(tk, sd["in_proj.weight"], sd["in_proj.bias"])
    lat = sd["latents"].unsqueeze(0).expand(B, -1, -1)
    for li in range(2):
        pre, sv = f"layers.{li}.", ctx["saved"][li]
        W, b = sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"]
        q = F.linear(lat, W[:D], b[:D]).view(B, N, Hh, hd).transpose(1, 2) * hd ** -0.5
        k = F.linear(cx, W[D:2 * D], b[D:2 * D]).view(B, T, Hh, hd).transpose(1, 2)
        v = F.linear(cx, W[2 * D:], b[2 * D:]).view(B, T, Hh, hd).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-2, -1), dim=-1) * mask((B * Hh, N, Tp), sv["off_P"]).view(B, Hh, N, Tp)[..., :T]
        o = (w @ v).transpose(1, 2).reshape(B, N, D)
        a = F.linear(o, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"]) * mask((B * N, D), sv["off_a"]).view(B, N, D)
        x = F.layer_norm(lat + a, (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
        h = F.gelu(F.linear(x, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])) * mask((B * N, 256), sv["off_h"]).view(B, N, 256)
        mo = F.linear(h, sd[pre + "mlp.3.weight"], sd[pre + "mlp.3.bias"]) * mask((B * N, D), sv["off_mo"]).view(B, N, D)
        lat = F.layer_norm(x + mo, (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)
    ref_out = F.linear(lat, sd["out_proj.weight"], sd["out_proj.bias"])
    ref_out.backward(d_out.float().cpu())
    assert relerr(out, ref_out) < 2e-2
    bad = {n: relerr(p.grad, sd[n].grad) for n, p in m.named_parameters() if relerr(p.grad, sd[n].grad) >= 5e-2}
    assert not bad, bad


def test_train_projector_flag_end_to_end():
    """VisionLanguageConfig.train_projector through the VLM and the trainer on the tiny golden configuration: off (the default, the
    reference's behaviour) the projector never changes and has no gradient; on, d(loss)/d(visual rows) reaches every projector parameter
    (autograd route and native trainer agree), the optimiser step moves the projector with proj_lr, and the loss is what it was."""
    from tests.golden_io import load, meta
    from tests.test_parity_gpu import _build_vlm
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z = load("vlm_tiny.npz")
    mt = meta(z)
    geom = {k: torch.from_numpy(z["geom:" + k]).cuda() for k in ("R", "t", "K", "depth_hist")}
    batch = {"pixel_values": torch.from_numpy(z["pixel_values"].astype(np.float32)).cuda(), "geom_token": geom,
             "input_ids": torch.from_numpy(z["input_ids"]).cuda(), "attention_mask": torch.from_numpy(z["attention_mask"]).cuda(),
             "labels": torch.from_numpy(z["labels"]).cuda()}
    kw = dict(images=batch["pixel_values"], geom_token=geom, input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
              labels=batch["labels"])
    off = _build_vlm(z, mt).train()
    off.projector.eval()
    loss_off = off(**kw)
    loss_off.backward()
    assert all(p.grad is None for p in off.projector.parameters())
    on = _build_vlm(z, mt).train()
    on.projector.eval()                                      # (dropout off so that both models see the same visual tokens)
    on.train_projector = True
    loss_on = on(**kw)
    assert abs(loss_on.item() - loss_off.item()) < 1e-4 * abs(loss_off.item())
    loss_on.backward()
    g_auto = {n: p.grad.clone() for n, p in on.projector.named_parameters()}
    assert all(v.abs().max().item() > 0 for v in g_auto.values())
    # native trainer: same gradients in the flat buffer, and the step moves the projector
    tr_model = _build_vlm(z, mt).train()
    tr_model.projector.eval()
    tr_model.train_projector = True
    tr = Stage1Trainer(tr_model, lr=1e-3, proj_lr=1e-2, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=2, max_grad_norm=None)
    w0 = {n: p.detach().clone() for n, p in tr_model.projector.named_parameters()}
    tr.micro_step(batch)
    for n, p in tr_model.projector.named_parameters():
        assert relerr(p.grad * 2.0, g_auto[n]) < 1e-3, n                 # (loss / grad_accum)
    assert all(torch.equal(p.detach(), w0[n]) for n, p in tr_model.projector.named_parameters())     # no step yet
    tr.micro_step(batch)
    assert tr.opt_step == 1
    moved = [n for n, p in tr_model.projector.named_parameters() if not torch.equal(p.detach(), w0[n])]
    assert len(moved) == len(w0)
    l2 = tr_model(**kw)                                      # compute copies follow the updated fp32 parameters
    assert torch.isfinite(l2) and abs(l2.item() - loss_off.item()) > 1e-6


def test_prefetch_in_train_projector_mode_caches_the_frozen_tokens_only():
    """with train_projector on, prefetch_images() for a future batch must not run the (trainable, train-mode-dropout)
    projector ahead of time: it caches the frozen tower's tokens, the training forward consumes exactly that cache (nothing stays parked),
    the dropout counter advances as without the prefetch, and loss and projector gradients are those of the un-prefetched pass."""
    from tests.golden_io import load, meta
    from tests.test_parity_gpu import _build_vlm
    z = load("vlm_tiny.npz")
    mt = meta(z)
    geom = {k: torch.from_numpy(z["geom:" + k]).cuda() for k in ("R", "t", "K", "depth_hist")}
    kw = dict(images=torch.from_numpy(z["pixel_values"].astype(np.float32)).cuda(), geom_token=geom,
              input_ids=torch.from_numpy(z["input_ids"]).cuda(), attention_mask=torch.from_numpy(z["attention_mask"]).cuda(),
              labels=torch.from_numpy(z["labels"]).cuda())
    runs = []
    for prefetch in (False, True):
        m = _build_vlm(z, mt).train()                        # projector in train mode: its dropout sites are live
        m.train_projector = True
        m.projector._drop_seed, m.projector._drop_offset = 1234, 0
        if prefetch:
            m.prefetch_images(kw["images"])
            assert m._prefetched is not None and m._prefetched[2] == "tokens"
            assert m.projector._drop_offset == 0             # the projector did not run
        loss = m(**kw)
        assert m._prefetched is None                         # consumed, not parked
        loss.backward()
        runs.append((loss.item(), m.projector._drop_offset, {n: p.grad.clone() for n, p in m.projector.named_parameters()}))
    (l0, o0, g0), (l1, o1, g1) = runs
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and o0 == o1          # (the CE row sums meet by f32 atomics; same dropout counter: the prefetch drew nothing)
    assert all(torch.equal(g0[n], g1[n]) for n in g0)
    # reference mode (train_projector off): the whole encode_images result is prefetched, as before
    m = _build_vlm(z, mt).train()
    m.prefetch_images(kw["images"])
    assert m._prefetched[2] == "encoded"
    assert torch.isfinite(m(**kw)) and m._prefetched is None
