"""Stage1Trainer semantics the reference's default launch has (train_sft.py:138-163,208-220 under Accelerate + DeepSpeed):
global-norm clipping at 1.0 (configs/deepspeed_zero3.json:15), empty-label micro-batches, derived weight copies after a
resume / checkpoint load, gradients of the native trainer == gradients of the autograd route == the reference's goldens."""
import numpy as np
import pytest
import torch

from tests.golden_io import bf16, load, meta

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _vlm():
    from tests.test_parity_gpu import _build_vlm
    z = load("vlm_tiny.npz")
    m = meta(z)
    geom = {k: torch.from_numpy(z["geom:" + k]).cuda() for k in ("R", "t", "K", "depth_hist")}
    batch = {"pixel_values": torch.from_numpy(z["pixel_values"].astype(np.float32)).cuda(), "geom_token": geom,
             "input_ids": torch.from_numpy(z["input_ids"]).cuda(), "attention_mask": torch.from_numpy(z["attention_mask"]).cuda(),
             "labels": torch.from_numpy(z["labels"]).cuda()}
    return z, m, (lambda: _build_vlm(z, m).train()), batch


def test_sumsq_and_clipped_adamw_vs_torch():
    """vq3_sumsq + the clip coefficient inside vq3_adamw_step against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on
    fp32 parameters fed the same (bf16-valued) gradients, three steps, clipping active (norm >> max_norm) and inactive."""
    from vggt_qwen3_amd import ops
    torch.manual_seed(0)
    n = 3 * 1000 * 1000 + 13
    for max_norm, gstd in ((1.0, 0.01), (1.0, 1e-6)):
        p0 = torch.randn(n, device="cuda")
        master, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        w = torch.empty(n, device="cuda", dtype=BF16)
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
        part, acc = torch.zeros(1024, device="cuda"), torch.zeros(1, device="cuda")
        for step in range(1, 4):
            g = (torch.randn(n, device="cuda") * gstd).to(BF16)
            acc.zero_()
            ops.sumsq(g, part, acc)
            want = g.double().pow(2).sum()
            assert abs(acc.item() - want.item()) < 1e-5 * want.item()
            gscale = 0.5                                           # 1 / world
            ops.adamw_step(master, m, v, g, w, 1e-3, 0.9, 0.999, 1e-8, 0.1, step, gscale, clip=(acc, max_norm))
            ref.grad = g.float() * gscale
            total = torch.nn.utils.clip_grad_norm_([ref], max_norm)
            assert (total.item() > max_norm) == (gstd > 1e-4)      # first case clips, second does not
            opt.step()
            assert relerr(master, ref.detach()) < 1e-6
            assert torch.equal(w, master.to(BF16))
    # f32 input form + accumulation into the same accumulator
    x = torch.randn(5001, device="cuda")
    acc.zero_()
    ops.sumsq(x, part, acc)
    ops.sumsq(x, part, acc)
    assert abs(acc.item() - 2 * x.double().pow(2).sum().item()) < 1e-5 * acc.item()


def test_native_trainer_gradients_vs_autograd_and_reference_golden():
    """Stage1Trainer.micro_step leaves in flat_g exactly what the autograd route (`loss.backward()`) produces - per tensor,
    rel err <= 2e-2 (they are the same kernels; f32 atomics in two reductions are the only difference) - and both agree with
    the gradients the REFERENCE produced on the same inputs and weights (tests/golden/vlm_tiny.npz, tools/make_golden.py)."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    a = build()
    tr = Stage1Trainer(a, lr=0.0, proj_lr=0.0, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=1)
    loss_a = tr.micro_step(batch)
    b = build()
    loss_b = b(images=batch["pixel_values"], geom_token=batch["geom_token"], input_ids=batch["input_ids"],
               attention_mask=batch["attention_mask"], labels=batch["labels"])
    loss_b.backward()
    assert abs(loss_a.item() - loss_b.item()) < 1e-5
    bad = {}
    for name, gb in b.text_model.grad_views.items():
        ga = a.text_model.grad_views[name]
        e = relerr(ga, gb)
        if e > 2e-2:
            bad[name] = e
        ref = bf16(z["g:text_model." + name]) if ("g:text_model." + name) in z else None
        if ref is not None:
            e = relerr(ga, ref)
            if e > 5e-2:
                bad["golden:" + name] = e
    assert not bad, bad
    # the autograd route exposes the flat buffer itself: .grad are views, nothing was cloned
    p0 = next(iter(b.text_model.parameters()))
    assert p0.grad.data_ptr() == b.text_model.grad_views["model.embed_tokens.weight"].data_ptr()
    # and accumulates across backward() calls until zero_grad(), like autograd does
    g1 = b.text_model.flat_g.float().clone()
    loss_b2 = b(images=batch["pixel_values"], geom_token=batch["geom_token"], input_ids=batch["input_ids"],
                attention_mask=batch["attention_mask"], labels=batch["labels"])
    loss_b2.backward()
    assert relerr(b.text_model.flat_g, 2 * g1) < 2e-2
    for p in b.parameters():
        p.grad = None
    loss_b3 = b(images=batch["pixel_values"], geom_token=batch["geom_token"], input_ids=batch["input_ids"],
                attention_mask=batch["attention_mask"], labels=batch["labels"])
    loss_b3.backward()
    assert relerr(b.text_model.flat_g, g1) < 2e-2


def test_trainer_global_norm_clipping():
    """With max_grad_norm far below the gradient norm the update equals AdamW on gradients scaled by max_norm / norm, the
    norm taken over text model AND geom_head gradients (one global norm, as DeepSpeed / clip_grad_norm_ take it)."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    a = build()
    # eps is made LARGE on purpose: Adam's step is scale-invariant for |g| >> eps, so only with |g| << eps does the update
    # become linear in the clip coefficient and the test can see a wrong one
    tr = Stage1Trainer(a, lr=1e-3, proj_lr=1e-3, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=1,
                       max_grad_norm=0.05, eps=1e-2)
    w0 = tr.master.clone()
    tr.micro_step(batch)
    g = a.text_model.flat_g.float()
    gg = tr.geom_grad[: tr._gn]
    norm = torch.sqrt(g.double().pow(2).sum() + gg.double().pow(2).sum()).item()
    assert norm > 0.1                                           # clipping is active (coefficient < 0.5)
    assert abs(tr.last_grad_norm.item() - norm) < 1e-4 * norm
    coef = 0.05 / (norm + 1e-6)
    gc = g * coef
    mm, vv = 0.1 * gc, 0.001 * gc * gc                           # first Adam step from zero moments
    want = w0 - 1e-3 * (mm / 0.1) / ((vv / 0.001).sqrt() + 1e-2)
    assert relerr(tr.master - w0, want - w0) < 1e-4
    unclipped = w0 - 1e-3 * g / (g.abs() + 1e-2)
    assert relerr(tr.master - w0, unclipped - w0) > 0.3         # and it is NOT what the unclipped step would have been
    # unclipped trainer moves differently where |g| is tiny relative to eps only; sanity: clip off => last_grad_norm None
    b = build()
    tb = Stage1Trainer(b, lr=1e-3, proj_lr=1e-3, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=1,
                       max_grad_norm=None)
    tb.micro_step(batch)
    assert tb.last_grad_norm is None


def test_empty_label_micro_batch_contributes_zero_gradient():
    """A micro-batch whose answer was truncated away (no label left) must not leave the previous window's gradients in
    flat_g, wherever it falls in the accumulation window: windows [empty, x] and [x, empty] give the same update as each
    other, and a window made only of it moves nothing but weight decay."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    empty = dict(batch, labels=torch.full_like(batch["labels"], -100))
    kw = dict(lr=1e-3, proj_lr=1e-3, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=2)
    res = []
    for order in ((empty, batch), (batch, empty)):
        a = build()
        tr = Stage1Trainer(a, **kw)
        tr.micro_step(batch); tr.micro_step(batch)              # a first window leaves gradients behind in flat_g
        w1 = tr.master.clone()
        l0 = tr.micro_step(order[0]); l1 = tr.micro_step(order[1])
        assert torch.isnan(l0 if order[0] is empty else l1)     # the reference's mean over zero targets is NaN too
        assert tr.opt_step == 2 and torch.isfinite(tr.master).all()
        res.append((tr.master - w1).clone())
        g_after = a.text_model.flat_g.float().clone()
        assert torch.isfinite(g_after).all()
    assert relerr(res[0], res[1]) < 2e-2
    a = build()
    tr = Stage1Trainer(a, **kw)
    tr.micro_step(batch); tr.micro_step(batch)
    w1 = tr.master.clone()
    gm1 = tr.geom_master.clone()
    tr.micro_step(empty); tr.micro_step(empty)
    assert float(a.text_model.flat_g.float().abs().max()) == 0.0
    # zero gradient: Adam's momentum still moves the weights a little (m from window 1), but far less than a real step
    assert (tr.master - w1).abs().max() <= 1.1e-3
    assert torch.equal(tr.geom_master, gm1)                     # no geom gradient on any rank -> geom_head untouched


def test_resume_refreshes_weight_derived_copies(tmp_path):
    """after load_trainer_state / load_checkpoint_if_available the e4m3 copies and the W^T dgrad copies
    must follow the loaded weights. Resume with grad_accum = 4 (W^T on) and the e4m3 forward, from a model built with
    DIFFERENT initial weights: loss and gradients of the next micro-batch equal the uninterrupted run's."""
    from vggt_qwen3_amd.checkpoint import load_checkpoint_if_available, load_trainer_state, save_model, save_trainer_state
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    kw = dict(lr=2e-3, proj_lr=2e-3, weight_decay=0.1, warmup_ratio=0.0, max_steps=40, grad_accum=4)
    a = build()
    a.text_model.enable_fp8_forward(True)
    ta = Stage1Trainer(a, **kw)
    assert a.text_model._wt is not None
    for _ in range(4):
        ta.micro_step(batch)
    save_trainer_state(ta, tmp_path)
    save_model(a, tmp_path)
    c = build()
    with torch.no_grad():
        c.text_model.flat_w.mul_(0.5)                           # different initial weights
    c.text_model.enable_fp8_forward(True)
    tc = Stage1Trainer(c, **kw)
    load_trainer_state(tc, tmp_path)
    assert torch.equal(c.text_model.flat_w, a.text_model.flat_w)
    for name, wt in c.text_model._wt.items():
        assert torch.equal(wt, c.text_model._w[name].t().contiguous()), name
    for name, (wq, ws) in c.text_model._fp8.items():
        assert torch.equal(wq, a.text_model._fp8[name][0]) and torch.equal(ws, a.text_model._fp8[name][1]), name
    la = ta.micro_step(batch)
    lc = tc.micro_step(batch)
    assert abs(la.item() - lc.item()) < 1e-5 * abs(la.item())
    a.text_model.flush_deferred(); c.text_model.flush_deferred()   # mid-window: the projections' weight-gradient GEMMs are still pending
    assert relerr(c.text_model.flat_g, a.text_model.flat_g) < 2e-2
    # model checkpoint loaded AFTER a trainer exists: derived copies and the fp32 master follow
    d = build()
    with torch.no_grad():
        d.text_model.flat_w.mul_(0.25)
    d.text_model.enable_fp8_forward(True)
    td = Stage1Trainer(d, **kw)
    assert load_checkpoint_if_available(d, str(tmp_path), verbose=False) is not None
    w_saved = torch.load(next((tmp_path / "pytorch_model_fp32").glob("*.bin")), map_location="cpu")
    assert torch.equal(td.master, d.text_model.flat_w.float())
    for name, wt in d.text_model._wt.items():
        assert torch.equal(wt, d.text_model._w[name].t().contiguous()), name
    k = "text_model.model.layers.0.mlp.down_proj.weight"
    if k in w_saved:
        assert torch.equal(d.text_model._w["l0.down"].float().cpu(), w_saved[k].to(BF16).float())


def test_perceiver_train_mode_dropout():
    """The reference's projector keeps its nn.Dropout(p=0.1) sites active under model.train() even though encode_images is
    no_grad (projector_perceiver.py:33,37,42,46-49; vggt_qwen3_vlm.py:128): train mode -> stochastic output with the right
    statistics, eval mode -> deterministic and equal to the dropout-free path; the kernel keeps the mean and zeroes ~p."""
    from vggt_qwen3_amd import ops
    from vggt_qwen3_amd.perceiver import PerceiverConfig, PerceiverProjector
    x = torch.ones(1 << 20, device="cuda", dtype=F32)
    ops.dropout_(x, 0.1, 1234, 0)
    zeros = float((x == 0).float().mean())
    assert abs(zeros - 0.1) < 3e-3 and abs(float(x.mean()) - 1.0) < 5e-3 and abs(float(x.max()) - 1 / 0.9) < 1e-6
    y = torch.ones(1 << 20, device="cuda", dtype=BF16)
    ops.dropout_(y, 0.1, 1234, 0)
    assert torch.equal(y == 0, x == 0)                                     # same (seed, offset) -> same mask in both dtypes
    z = torch.ones(1 << 20, device="cuda", dtype=F32)
    ops.dropout_(z, 0.1, 1234, 1 << 20)
    assert not torch.equal(z == 0, x == 0)
    torch.manual_seed(0)
    proj = PerceiverProjector(PerceiverConfig(latent_dim=128, num_latents=16, num_heads=2, num_layers=2, ffn_dim=256, dropout=0.1),
                              64, 96).cuda()
    tok = torch.randn(2, 24, 64, device="cuda")
    proj.eval()
    e1, e2 = proj(tok), proj(tok)
    assert torch.equal(e1, e2)
    proj.train()
    t1, t2 = proj(tok), proj(tok)
    assert not torch.equal(t1, t2)                                         # a fresh mask per call
    d = relerr(t1, e1)
    assert 1e-3 < d < 0.6, d                                               # perturbed, not destroyed
    proj.cfg.dropout = 0.0
    assert torch.equal(proj(tok), e1)                                      # p = 0 in train mode == eval path


def test_deferred_weight_gradients_match_per_microbatch_gemms():
    """Stage1Trainer(wgrad_defer=n): the projections' weight-gradient GEMMs run once per n micro-batches over the concatenated
    token rows (qwen3.py "deferred weight gradients"). Over one window of 4 DIFFERENT micro-batches - labels moved around, one
    micro-batch without any label, window boundary in the middle of a group - flat_g must equal what per-micro-batch GEMMs
    accumulate (up to their bf16 read-modify-write rounding), and must be no further from an f32 accumulation of the per-micro-batch
    gradients than the per-micro-batch path is. The pairing guard must refuse two forwards in a row."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()

    def variants():
        out = []
        for k in range(4):
            b = dict(batch)
            ids = batch["input_ids"].clone()
            lab = batch["labels"].clone()
            keep = lab != -100
            if k == 2:
                lab[:] = -100                                   # a micro-batch whose answer was truncated away
            elif k:
                g = torch.Generator().manual_seed(100 + k)
                noise = torch.randint(5, 200, ids.shape, generator=g).to(ids.device)
                ids = torch.where(keep, noise, ids)
                lab = torch.where(keep, noise, lab)
            b["input_ids"], b["labels"] = ids, lab
            out.append(b)
        return out

    mbs = variants()
    grads = {}
    for depth in (1, 2, 3, 4):
        model = build()
        tr = Stage1Trainer(model, lr=0.0, proj_lr=0.0, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=4,
                           wgrad_defer=depth, text_group=1, max_grad_norm=None)
        assert model.text_model._wd_depth == depth
        for b in mbs:
            tr.micro_step(b)
        assert not any(model.text_model._wd_rows)               # nothing left pending after the boundary
        grads[depth] = model.text_model.flat_g.float().clone()
        del tr, model
    # f32 reference: per-micro-batch gradients (grad_accum=1 windows, scale 1/4 applied afterwards) summed in f32
    model = build()
    tr = Stage1Trainer(model, lr=0.0, proj_lr=0.0, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=1, max_grad_norm=None)
    ref = torch.zeros_like(grads[1])
    for k, b in enumerate(mbs):
        tr.micro_step(b)
        if k != 2:
            ref += model.text_model.flat_g.float() * 0.25
    e1 = relerr(grads[1], ref)
    for depth in (2, 3, 4):
        assert relerr(grads[depth], grads[1]) < 1e-2, (depth, relerr(grads[depth], grads[1]))
        assert relerr(grads[depth], ref) <= e1 * 1.5 + 1e-4, (depth, relerr(grads[depth], ref), e1)
    # pairing guard
    model = build()
    tm = model.text_model
    tm.enable_wgrad_deferral(2)
    st1 = model.forward_state(batch["pixel_values"], batch["geom_token"], batch["input_ids"], batch["attention_mask"], batch["labels"],
                              need_grad=True)
    st2 = model.forward_state(batch["pixel_values"], batch["geom_token"], batch["input_ids"], batch["attention_mask"], batch["labels"],
                              need_grad=True)
    with pytest.raises(RuntimeError, match="alternate"):
        model._backward_text(st1, 1.0, False)
    model._backward_text(st2, 1.0, False)                       # the latest forward's operands are intact


class _Tower(torch.nn.Module):
    """A stand-in tower whose output depends on its input, sample by sample (the goldens' stub returns a constant)."""
    def __init__(self, base):
        super().__init__()
        self.base, self.embed_dim = base, base.shape[-1]

    def aggregator(self, images):
        scale = 1.0 + images.float().mean(dim=(1, 2, 3, 4))                       # [B']
        out = self.base[:1].float() * scale.view(-1, *([1] * (self.base.dim() - 1)))
        return [out.to(self.base.dtype)], 5


def test_vision_group_matches_per_microbatch_tower():
    """Stage1Trainer(vision_group=n) + micro_step(upcoming=...): the frozen aggregator runs once over n micro-batches' concatenated
    images; every micro-batch must get exactly the visual tokens (and so loss and gradients) its own pass would have produced,
    results are handed out to the right batch (different images per micro-batch), and nothing is left waiting afterwards."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    mbs = []
    for k in range(3):
        b = dict(batch)
        g = torch.Generator().manual_seed(7 + k)
        b["pixel_values"] = (batch["pixel_values"] * 0.5 + 0.5 * torch.rand(batch["pixel_values"].shape, generator=g).cuda()).contiguous()
        mbs.append(b)
    out = {}
    for vg in (1, 3):
        model = build()
        model.vision_model = _Tower(model.vision_model.agg)
        tr = Stage1Trainer(model, lr=0.0, proj_lr=0.0, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=3, wgrad_defer=1,
                           vision_group=vg, text_group=1, max_grad_norm=None)
        losses = [float(tr.micro_step(b, upcoming=mbs[i + 1:]).item()) for i, b in enumerate(mbs)]
        assert not model._vis_group
        out[vg] = (losses, model.text_model.flat_g.float().clone())
        # the aggregator output itself, grouped vs alone (bitwise: every row of every GEMM sees the same operands in the same order)
        if vg == 3:
            model.precompute_vision([b["pixel_values"] for b in mbs])
            grouped = [model._take_grouped(b["pixel_values"]) for b in mbs]
            for b, ga in zip(mbs, grouped):
                alone = model.vision_model.aggregator(b["pixel_values"])[0][-1]
                assert relerr(ga, alone) < 2e-3
    assert len(set(out[1][0])) == 3                                  # the three micro-batches really differ
    for a, b in zip(out[1][0], out[3][0]):
        assert abs(a - b) < 2e-3 * abs(a), (out[1][0], out[3][0])
    assert relerr(out[3][1], out[1][1]) < 2e-2


def test_merged_micro_batches_match_one_by_one():
    """Stage1Trainer(text_group=n) + micro_step(upcoming=...): n micro-batches of a window run as ONE forward/backward over their
    concatenated samples. Each micro-batch's loss must be the mean over ITS OWN labelled rows, the gradient that of their sum (what
    the micro-batches give one by one) - with different images, ids and label counts per micro-batch, one micro-batch without labels,
    a window (5) that is not a multiple of the group (3), and geom_head gradients included."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    mbs = []
    for k in range(5):
        b = dict(batch)
        g = torch.Generator().manual_seed(50 + k)
        lab = batch["labels"].clone()
        ids = batch["input_ids"].clone()
        keep = lab != -100
        if k == 1:
            lab[:] = -100                                        # no labelled row at all
        elif k:
            noise = torch.randint(5, 200, ids.shape, generator=g).to(ids.device)
            drop = (torch.rand(ids.shape, generator=g) < 0.15 * k).to(ids.device) & keep      # different label counts per micro-batch
            ids = torch.where(keep, noise, ids)
            lab = torch.where(keep & ~drop, noise, torch.full_like(lab, -100))
        b["input_ids"], b["labels"] = ids, lab
        b["pixel_values"] = (batch["pixel_values"] * 0.5 + 0.5 * torch.rand(batch["pixel_values"].shape, generator=g).cuda()).contiguous()
        mbs.append(b)
    res = {}
    for tg in (1, 3):
        model = build()
        model.vision_model = _Tower(model.vision_model.agg)
        tr = Stage1Trainer(model, lr=0.0, proj_lr=0.0, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=5, wgrad_defer=1,
                           vision_group=1, text_group=tg, max_grad_norm=None)
        losses = [float(tr.micro_step(b, upcoming=mbs[i + 1:]).item()) for i, b in enumerate(mbs)]
        assert not tr._merged_pending and tr.micro == 5 and tr.opt_step == 1
        res[tg] = (losses, model.text_model.flat_g.float().clone(), tr.geom_grad.clone())
    l1, l3 = res[1][0], res[3][0]
    assert l1[1] != l1[1] and l3[1] != l3[1]                     # NaN for the unlabelled micro-batch, as the reference's mean over nothing
    for k in (0, 2, 3, 4):
        assert abs(l1[k] - l3[k]) < 2e-3 * abs(l1[k]), (k, l1, l3)
    assert relerr(res[3][1], res[1][1]) < 2e-2, relerr(res[3][1], res[1][1])
    assert relerr(res[3][2][:-1], res[1][2][:-1]) < 2e-2
    # out-of-order hand-back is refused
    model = build()
    model.vision_model = _Tower(model.vision_model.agg)
    tr = Stage1Trainer(model, lr=0.0, proj_lr=0.0, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=4, text_group=2,
                       max_grad_norm=None)
    tr.micro_step(mbs[0], upcoming=[mbs[2]])
    with pytest.raises(RuntimeError, match="same objects"):
        tr.micro_step(mbs[3])
    # ... and the interrupted group can be accounted for: the gradient of mbs[2] is already in flat_g
    rest = tr.flush_pending()
    assert len(rest) == 1 and tr.micro == 2 and not tr._merged_pending and tr.opt_step == 0
    tr.micro_step(mbs[3])                                        # a different batch is accepted again
    assert tr.micro == 3


def test_merged_group_keeps_counters_and_weights_in_phase():
    """with merged micro-batches the window's AdamW step must run with the micro_step() call that returns the window's LAST
    loss - for every call before it `micro`, `opt_step`, lrs() and the weights are those of a loop that runs the micro-batches
    one by one (a step_N save inside the group must not contain the window's update)."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    mbs = [dict(batch) for _ in range(6)]
    model = build()
    model.vision_model = _Tower(model.vision_model.agg)
    tr = Stage1Trainer(model, lr=1e-3, proj_lr=1e-3, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=3, text_group=3)
    w0 = model.text_model.flat_w.clone()
    for i in range(3):
        tr.micro_step(mbs[i], upcoming=mbs[i + 1:3])
        assert tr.micro == i + 1
        if i < 2:
            assert tr.opt_step == 0 and torch.equal(model.text_model.flat_w, w0), i     # the update waits for the last loss
    assert tr.opt_step == 1 and not torch.equal(model.text_model.flat_w, w0) and not tr._opt_due
    # an interrupted group: flush_pending() runs the deferred step exactly once
    w1 = model.text_model.flat_w.clone()
    tr.micro_step(mbs[3], upcoming=mbs[4:6])
    assert tr.opt_step == 1 and torch.equal(model.text_model.flat_w, w1) and tr._opt_due
    tr.flush_pending()
    assert tr.micro == 6 and tr.opt_step == 2 and not tr._opt_due and not torch.equal(model.text_model.flat_w, w1)
    assert tr.flush_pending() == [] and tr.opt_step == 2


def test_set_schedule_never_rewinds_the_micro_batch_counter():
    """set_schedule() between two windows keeps `micro` (max_steps, step_N names, the saved state count from it); a new
    grad_accum moves it UP to the next window boundary of the new schedule, so boundaries and the counter stay in phase."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    model = build()
    model.vision_model = _Tower(model.vision_model.agg)
    tr = Stage1Trainer(model, lr=1e-3, proj_lr=1e-3, weight_decay=0.0, warmup_ratio=0.0, max_steps=100, grad_accum=2, text_group=1)
    for _ in range(4):
        tr.micro_step(dict(batch))
    assert tr.micro == 4 and tr.opt_step == 2
    tr.set_schedule(text_group=2)                       # same window: the counter is untouched
    assert tr.micro == 4
    tr.set_schedule(grad_accum=3)                       # 4 -> 6: the next multiple of the new window, never back to 0 or 3
    assert tr.micro == 6 and tr.micro % tr.grad_accum == 0
    for i in range(3):
        tr.micro_step(dict(batch))
        assert tr.opt_step == (3 if i == 2 else 2)      # the optimiser step closes the new window of 3
    assert tr.micro == 9
    tr.micro_step(dict(batch))
    with pytest.raises(RuntimeError):
        tr.set_schedule(grad_accum=2)                   # mid-window: refused


def test_optimizer_step_beside_the_next_pass_equals_the_serial_schedule():
    """Stage1Trainer.overlap_optimizer (on inside fit() and bench.py): the window's optimiser step runs on a side stream while the next
    window's pass starts with the frozen vision tower; every reader of an updated tensor waits on model._weights_gate. Same arithmetic in
    the same order per tensor, so three windows (merged passes, geometry tokens, clipping on) must leave bit-identical weights, optimiser
    state and losses - and a direct look at the weights after sync_optimizer() sees the finished step."""
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z, m, build, batch = _vlm()
    mbs = [dict(batch) for _ in range(6)]
    for i, b in enumerate(mbs):                               # different labels per micro-batch
        lab = b["labels"].clone()
        lab[i % lab.shape[0]] = -100
        b["labels"] = lab
    out = []
    for overlap in (False, True):
        model = build()
        model.vision_model = _Tower(model.vision_model.agg)
        tr = Stage1Trainer(model, lr=1e-3, proj_lr=1e-3, weight_decay=0.1, warmup_ratio=0.0, max_steps=100, grad_accum=2, text_group=2)
        tr.overlap_optimizer = overlap
        losses = []
        for w in range(3):
            pair = mbs[2 * w: 2 * w + 2]
            losses.append(float(tr.micro_step(pair[0], upcoming=pair[1:]).item()))
            losses.append(float(tr.micro_step(pair[1]).item()))
            if overlap:
                assert model._weights_gate is not None        # the step was left running for the next pass to wait on
        tr.sync_optimizer()
        assert model._weights_gate is None and tr.opt_step == 3
        torch.cuda.synchronize()
        out.append((losses, model.text_model.flat_w.clone(), tr.master.clone(), tr.m.clone(), tr.geom_master.clone()))
    (l0, w0, ma0, m0, g0), (l1, w1, ma1, m1, g1) = out
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(l0, l1))       # (the loss sums its rows by f32 atomics)
    assert torch.equal(w0, w1) and torch.equal(ma0, ma1) and torch.equal(m0, m1) and torch.equal(g0, g1)


def test_training_is_bit_reproducible_at_full_width():
    """Round 5: two training runs from the same seeds leave the SAME BITS in the weights, the fp32 master copy and Adam's moments - real
    tower kernels (a 2-block VGGT at C = 1024), the Perceiver with train-mode dropout (counter-based masks), two Qwen3-4B-width layers,
    merged passes, deferred weight gradients, clipping, the optimiser step beside the next pass. Nothing on the path sums in an order the
    scheduler chooses any more (the lm_head's input-gradient split and the q/k-norm weight-gradient partials did until this round); what
    remains nondeterministic is only the reported loss scalar (an f32-atomic sum over rows)."""
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.trainer import Stage1Trainer
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig

    def build():
        q = Qwen3Config.qwen3_4b(); q.num_hidden_layers = 2; q.vocab_size = 4096
        p = PerceiverConfig(latent_dim=512, num_latents=32, num_heads=2, num_layers=2, ffn_dim=1024, dropout=0.1)
        cfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=32, geom_tokens=4, projector_cfg=p,
                                   text_config=q, vision_config=dict(depth=1, dino_depth=1), device="cuda", seed=11)
        return VGGTQwen3VLM(cfg).train()

    def batches():
        g = torch.Generator().manual_seed(21)
        out = []
        for i in range(4):
            B, V, L = 3, 1, 96
            ids = torch.randint(10, 4000, (B, L), generator=g)
            ids[:, 70:] = 0
            labels = torch.full((B, L), -100)
            for r in range(B):
                ids[r, 8 + r] = 4096                           # <image>: the 36-row span [4 geom | 32 visual] lies under attended text
                labels[r, 60:70] = ids[r, 60:70]
            geom = {"R": torch.randn(B, V, 9, generator=g), "t": torch.randn(B, V, 3, generator=g), "K": torch.randn(B, V, 9, generator=g),
                    "depth_hist": torch.rand(B, V, 16, generator=g)}
            out.append({"pixel_values": torch.rand(B, V, 3, 224, 224, generator=g).cuda(), "geom_token": {k: v.cuda() for k, v in geom.items()},
                        "input_ids": ids.cuda(), "attention_mask": (ids != 0).long().cuda(), "labels": labels.cuda()})
        return out

    res = []
    for run in range(2):
        model = build()
        assert model.image_id == 4096
        tr = Stage1Trainer(model, lr=1e-3, proj_lr=1e-3, weight_decay=0.1, warmup_ratio=0.0, max_steps=100, grad_accum=2, text_group=2)
        tr.overlap_optimizer = True
        bs = batches()
        losses = []
        for w in range(2):
            pair = bs[2 * w: 2 * w + 2]
            losses.append(float(tr.micro_step(pair[0], upcoming=pair[1:]).item()))
            losses.append(float(tr.micro_step(pair[1]).item()))
        tr.sync_optimizer()
        torch.cuda.synchronize()
        tr.check_kernels()
        res.append((losses, model.text_model.flat_w.clone(), tr.master.clone(), tr.m.clone(), tr.v.clone(), tr.geom_master.clone()))
        del tr, model
    (l0, *t0), (l1, *t1) = res
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(l0, l1)) and all(x == x for x in l0)
    for k, (a, b) in enumerate(zip(t0, t1)):
        assert torch.equal(a, b), (k, float((a.float() - b.float()).abs().max()))
    assert (t0[0].float() - build().text_model.flat_w.float()).abs().max() > 0          # (the weights did move)
