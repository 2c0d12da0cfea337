"""CPU tests of the host-side logic around the HIP path: integer splice map and label-row selection against the
oracle, LR schedule against transformers, bucket planning, and the N>1 gradient exchange on gloo (world_size 2)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import qwen3 as oq
from oracle import vlm as ovlm


def test_srcmap_matches_oracle_bit_exact():
    from vggt_qwen3_amd.vlm import build_srcmap
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 50, (5, 200), generator=g)
    ids[ids == 7] = 8
    for b, p in enumerate([3, 17, 40, 63, 0]):
        ids[b, p] = 7
    ids[4, 0] = 9                      # row without <image>
    ids[2, 45] = 7                     # repeated <image>: the later span overwrites the earlier one
    a = build_srcmap(ids, 7, 136)
    b = ovlm.splice_srcmap(ids, 136, 7)
    assert a.dtype == torch.int32 and torch.equal(a, b)
    assert (a[4] == -1).all()
    assert a[2, 45] == 0 and a[2, 44] == 4
    with pytest.raises(RuntimeError):
        build_srcmap(ids[:, :100], 7, 136)


def test_label_rows_match_shifted_cross_entropy():
    from vggt_qwen3_amd.qwen3 import Qwen3ForCausalLM
    g = torch.Generator().manual_seed(1)
    B, L, V = 3, 20, 11
    labels = torch.full((B, L), -100)
    labels[0, 5:9] = torch.randint(0, V, (4,), generator=g)
    labels[1, 19] = 3
    labels[2, 0] = 4                   # a label at position 0 is never predicted (shift)
    idx, tgt = Qwen3ForCausalLM.label_rows(labels)
    logits = torch.randn(B, L, V, generator=g)
    ref = oq.causal_lm_loss(logits, labels)
    flat = logits.reshape(-1, V)[idx.long()]
    mine = torch.nn.functional.cross_entropy(flat, tgt.long(), reduction="mean")
    assert torch.allclose(ref, mine, atol=1e-6)
    assert idx.tolist() == [4, 5, 6, 7, 20 + 18]


def test_cosine_schedule_matches_transformers():
    from transformers import get_cosine_schedule_with_warmup
    from vggt_qwen3_amd.trainer import cosine_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1.0)
    total, warm = 1000, 30
    sch = get_cosine_schedule_with_warmup(opt, warm, total)
    for step in range(0, total, 37):
        while sch.last_epoch < step:
            opt.step(); sch.step()
        assert math.isclose(sch.get_last_lr()[0], cosine_with_warmup(step, warm, total), rel_tol=1e-9, abs_tol=1e-12)


def _fake_table(L=6, H=64, I=128, D=128, nqkv=256, V=50):
    ent = [("embed", (V, H))]
    for i in range(L):
        ent += [(f"l{i}.qkv", (nqkv, H)), (f"l{i}.o", (H, 128)), (f"l{i}.gu", (2 * I, H)), (f"l{i}.down", (H, I)),
                (f"l{i}.ln1", (H,)), (f"l{i}.ln2", (H,)), (f"l{i}.qn", (D,)), (f"l{i}.kn", (D,))]
    ent.append(("norm", (H,)))
    off, table = 0, {}
    for n, s in ent:
        table[n] = (off, s)
        off += (math.prod(s) + 63) // 64 * 64
    return table, off


def test_bucket_plan_covers_buffer_exactly():
    from vggt_qwen3_amd import dp
    table, total = _fake_table()
    for bl in (1, 2, 4, 6, 7):
        buckets, emb = dp.plan_buckets(table, 6, bl)
        dp.check_cover(buckets, emb, total)
        assert len(buckets) == math.ceil(6 / bl)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vggt_qwen3_amd import dp
        table, total = _fake_table()
        buckets, emb = dp.plan_buckets(table, 6, 4)
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(total, generator=g)
        mine = flat.clone()
        dp.allreduce_in_backward_order(flat, buckets, emb)
        others = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        ref = sum(others)
        ret[rank] = (torch.allclose(flat, ref, atol=1e-6), float((flat - mine).abs().sum()) > 0)
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r][0] and ret[r][1] for r in range(world)), dict(ret)


def _dp_shard_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vggt_qwen3_amd import dp
        table, total = _fake_table()
        buckets, emb = dp.plan_buckets(table, 6, 4)
        g = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        ref = sum(g)
        flat = g[rank].clone()
        ok = True
        spans = list(buckets.values()) + [emb, (7, 7 + 3 * world * dp.SHARD_ALIGN + 5)]      # (+ a span with a ragged tail; overlaps: reset below)
        for lo, hi in spans[:-1]:
            dp.reduce_scatter_span(flat, lo, hi, rank, world)
            s, tail = dp.shard_layout(lo, hi, world)
            assert s % dp.SHARD_ALIGN == 0 and lo + world * s == tail <= hi and hi - tail < world * dp.SHARD_ALIGN + world
            mine = slice(lo + rank * s, lo + (rank + 1) * s)
            ok = ok and torch.allclose(flat[mine], ref[mine], atol=1e-6) and torch.allclose(flat[tail:hi], ref[tail:hi], atol=1e-6)
            flat[lo:tail] = float("nan")                     # only this rank's shard may be relied on ...
            flat[mine] = ref[mine]
            dp.all_gather_span(flat, lo, hi, rank, world)    # ... and the gather restores every shard on every rank
            ok = ok and torch.allclose(flat[lo:hi], ref[lo:hi], atol=1e-6)
        lo, hi = spans[-1]
        flat = g[rank].clone()
        dp.reduce_scatter_span(flat, lo, hi, rank, world)
        s, tail = dp.shard_layout(lo, hi, world)
        ok = ok and s == 3 * dp.SHARD_ALIGN and hi - tail == 5 and torch.allclose(flat[tail:hi], ref[tail:hi], atol=1e-6)
        ok = ok and torch.allclose(flat[lo + rank * s: lo + (rank + 1) * s], ref[lo + rank * s: lo + (rank + 1) * s], atol=1e-6)
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_reduce_scatter_all_gather_world2_gloo():
    """dp.reduce_scatter_span + dp.all_gather_span (the opt-in exchange of Stage1Trainer(dp_mode="sharded")) over every bucket of the
    plan: each rank ends up with the all-reduce's sum in its shard and in the replicated tail, and the gather makes the span whole."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_shard_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_checkpoint_layout_and_search_order(tmp_path):
    """File layout / index / search order of vggt_qwen3_amd.checkpoint (reference: qa_inference.py:51-105), on a small
    stand-in module (no GPU): sharded dir wins over flat files, legacy dir name accepted, first flat file otherwise."""
    import json
    from vggt_qwen3_amd import checkpoint as ck

    def make(seed):
        torch.manual_seed(seed)
        m = torch.nn.Module()
        m.text_model = torch.nn.Module()
        m.text_model.lm_head = torch.nn.Linear(8, 16, bias=False)
        m.text_model.model = torch.nn.Module()
        m.text_model.model.embed_tokens = torch.nn.Embedding(16, 8)
        m.text_model.lm_head.weight = m.text_model.model.embed_tokens.weight
        m.projector = torch.nn.Linear(8, 8)
        m.vision_model = torch.nn.Linear(4, 4)
        return m

    a = make(1)
    wm = ck.save_model(a, tmp_path / "run", max_shard_bytes=300)
    root = tmp_path / "run" / ck.MERGED_DIR
    idx = json.loads((root / ck.INDEX_NAME).read_text())
    assert idx["weight_map"] == wm
    assert sorted(wm) == ["projector.bias", "projector.weight", "text_model.model.embed_tokens.weight"]
    assert len(set(wm.values())) >= 2 and all((root / f).exists() for f in wm.values())
    assert idx["metadata"]["total_size"] == 4 * (8 + 64 + 128)
    # a decoy flat file next to the sharded dir must be ignored
    torch.save({"projector.bias": torch.full((8,), 9.0)}, tmp_path / "run" / "zzz.bin")
    b = make(2)
    rep = ck.load_checkpoint_if_available(b, str(tmp_path / "run"), verbose=False)
    assert sorted(rep["matched"]) == sorted(wm)
    assert [k for k in rep["missing"] if not k.startswith("vision_model")] == []
    for k in wm:
        assert torch.equal(a.state_dict()[k], b.state_dict()[k])
    assert torch.equal(b.text_model.lm_head.weight, a.text_model.model.embed_tokens.weight)
    # legacy directory name
    (tmp_path / "legacy").mkdir()
    root.rename(tmp_path / "legacy" / ck.LEGACY_MERGED_DIR)
    c = make(3)
    assert ck.load_checkpoint_if_available(c, str(tmp_path / "legacy"), verbose=False) is not None
    assert torch.equal(c.projector.weight, a.projector.weight)
    # flat file: only the first one is read; unknown keys are reported, not fatal
    (tmp_path / "flat").mkdir()
    torch.save({"projector.bias": torch.full((8,), 3.0), "nope": torch.zeros(1)}, tmp_path / "flat" / "a.bin")
    d = make(4)
    rep = ck.load_checkpoint_if_available(d, str(tmp_path / "flat"), verbose=False)
    assert rep["unexpected"] == ["nope"] and torch.all(d.projector.bias == 3.0)
    # shape mismatch raises, missing directory / empty directory keep the base weights
    torch.save({"projector.bias": torch.zeros(3)}, tmp_path / "flat" / "a.bin")
    with pytest.raises(RuntimeError, match="size mismatch"):
        ck.load_checkpoint_if_available(make(5), str(tmp_path / "flat"), verbose=False)
    assert ck.load_checkpoint_if_available(make(6), str(tmp_path / "absent"), verbose=False) is None
    (tmp_path / "empty").mkdir()
    assert ck.load_checkpoint_if_available(make(7), str(tmp_path / "empty"), verbose=False) is None
    assert ck.load_checkpoint_if_available(make(8), None) is None


def test_generate_host_logic_cpu():
    """Host-side pieces of the inference path that need no GPU: the insertion splice (qa_inference.py:119-145, integer
    layout bit-exact) and the attended-span detection that decides how prompts are compacted."""
    from vggt_qwen3_amd.generate import _row_spans, insert_vision_tokens
    ids = torch.tensor([[5, 6, 9, 7, 8], [1, 9, 2, 3, 9]])
    mask = torch.tensor([[1, 1, 1, 1, 0], [0, 1, 1, 1, 1]])
    emb = torch.arange(2 * 5 * 4, dtype=torch.float32).view(2, 5, 4)
    vis = -torch.ones(2, 3, 4)
    e2, m2 = insert_vision_tokens(ids, mask, emb, vis, 9)
    assert e2.shape == (2, 7, 4) and m2.tolist() == [[1, 1, 1, 1, 1, 1, 0], [0, 1, 1, 1, 1, 1, 1]]
    assert torch.equal(e2[:, :2], emb[:, :2]) and torch.equal(e2[:, 2:5], vis) and torch.equal(e2[:, 5:], emb[:, 3:])
    e3, m3 = insert_vision_tokens(ids, mask, emb, vis, 77)          # no <image>: untouched, same objects
    assert e3 is emb and m3 is mask
    assert _row_spans(torch.tensor([[1, 1, 1], [0, 1, 1], [0, 1, 0]])) == [(0, 3), (1, 2), (1, 1)]
    with pytest.raises(ValueError, match="contiguous"):
        _row_spans(torch.tensor([[1, 0, 1]]))
    with pytest.raises(ValueError, match="attends to nothing"):
        _row_spans(torch.tensor([[0, 0, 0]]))


def test_collate_size_rules_and_plan_cpu():
    """torchvision's Resize / CenterCrop integer rules as the batch builder applies them, against the oracle's
    restatement, and the plan cache of the library's host-side resampling routine."""
    from oracle import preprocess as opre
    from vggt_qwen3_amd import collate
    for h, w, S in [(480, 640, 448), (640, 480, 448), (448, 448, 448), (449, 448, 448), (100, 1000, 56), (37, 53, 16),
                    (3, 1000, 8)]:
        assert collate.resized_size(h, w, S) == opre.resized_size(h, w, S)
        nh, nw = collate.resized_size(h, w, S)
        assert min(nh, nw) == S or (h, w) == (nh, nw)
        assert collate.crop_offset(nw, S) == int(round((nw - S) / 2.0)) and collate.crop_offset(nh, S) == int(round((nh - S) / 2.0))
    assert collate.crop_offset(451, 448) == 2 and collate.crop_offset(453, 448) == 2      # 1.5 -> 2, 2.5 -> 2 (half to even)
    ks, b, c = collate.axis_plan(640, 448)
    k2, b2, c2 = opre.precompute_coeffs(640, 448)
    assert ks == k2 and (b == b2).all() and (c == c2).all()
    assert collate.axis_plan(640, 448)[1] is b                                              # cached
    rows = collate._max_src_rows(b, collate.crop_offset(448, 448), 448, 16)
    assert rows == int(max(b[min(y + 15, 447), 0] + b[min(y + 15, 447), 1] - b[y, 0] for y in range(0, 448, 16)))


def test_bench_synthetic_batch_follows_survey_layout():
    """bench.py's synthetic Stage-1 batch: [question tokens, "\\n", <image>, "\\n", answer tokens, pad...] to
    L = 200, labels only on the answer, attention_mask = ids != pad, pixel values in [0, 1), one <image> per row, the 128-token
    visual span fits, and the FLOP model reproduces the survey's per-sample figures."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("vq3_bench_cpu", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, V, L, img, vocab, image_id, pad, nl = 6, 2, 200, 28, 151936, 151936, 151643, 198
    b = bench.synthetic_batch(B, V, L, img, vocab, image_id, pad, nl, 1234, torch.device("cpu"), True)
    ids, lab, mask = b["input_ids"], b["labels"], b["attention_mask"]
    assert ids.shape == (B, L) and ids.dtype == torch.int64 and b["pixel_values"].shape == (B, V, 3, img, img)
    assert 0.0 <= float(b["pixel_values"].min()) and float(b["pixel_values"].max()) < 1.0
    assert torch.equal(mask, (ids != pad).long())
    for r in range(B):
        pos = (ids[r] == image_id).nonzero().flatten()
        assert pos.numel() == 1 and ids[r, pos - 1] == nl and ids[r, pos + 1] == nl and 6 <= int(pos) - 1 <= 40
        assert int(pos) + 136 <= L                                           # visual + geometry span fits (no overrun)
        n = int(mask[r].sum())
        ans = (lab[r] != -100).nonzero().flatten()
        assert 1 <= ans.numel() <= 4 and int(ans[0]) == int(pos) + 2 and int(ans[-1]) == n - 1
        assert torch.equal(lab[r, ans], ids[r, ans]) and (ids[r, n:] == pad).all()
    g = b["geom_token"]
    assert g["R"].shape == (B, V, 9) and g["t"].shape == (B, V, 3) and g["K"].shape == (B, V, 9) and g["depth_hist"].shape == (B, V, 16)
    assert torch.allclose(g["depth_hist"].sum(-1), torch.ones(B, V), atol=1e-5) and (g["depth_hist"] >= 0).all()
    # 2.18 TF VGGT fwd (V=1, 448 px), 0.316 TF Perceiver fwd, 1.621 TF Qwen3 fwd at L=200 -> 7.36 TF train
    assert abs(bench.flops_vggt(1, 448) / 1e12 - 2.18) < 0.03
    assert abs(bench.flops_perceiver() / 1e12 - 0.316) < 0.005
    assert abs(bench.flops_qwen(200) / 1e12 - 1.621) < 0.01
    assert abs((bench.flops_vggt(1, 448) + bench.flops_perceiver() + 3 * bench.flops_qwen(200)) / 1e12 - 7.36) < 0.05
    assert abs(bench.flops_vggt(8, 448) / 1e12 - 23.25) < 0.3


def test_scheduler_rule_matches_accelerated_scheduler(monkeypatch):
    """The LR each optimiser step uses, against the reference's actual stack: transformers' cosine LambdaLR wrapped in
    accelerate's AcceleratedScheduler with gradient accumulation, at 1 and 4 processes (train_sft.py:158-163,217-220;
    accelerate/scheduler.py:54-82). Stage1Trainer.lr_mult(opt_step) = schedule_multiplier(opt_step, world, warmup, total)."""
    from types import SimpleNamespace
    import accelerate.scheduler as asch
    from transformers import get_cosine_schedule_with_warmup
    from vggt_qwen3_amd.trainer import schedule_multiplier
    total, warm, accum = 400, 12, 4
    for world in (1, 4):
        monkeypatch.setattr(asch, "AcceleratorState", lambda: SimpleNamespace(num_processes=world))
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=1.0)
        opt.step_was_skipped = False
        sch = asch.AcceleratedScheduler(get_cosine_schedule_with_warmup(opt, warm, total), [opt], step_with_optimizer=True,
                                        split_batches=False)
        opt_step = 0
        for micro in range(160):
            boundary = micro % accum == accum - 1
            sch.gradient_state._set_sync_gradients(boundary)
            if boundary:
                opt_step += 1
                used = opt.param_groups[0]["lr"]                 # the LR optimizer.step() applies now
                assert math.isclose(used, schedule_multiplier(opt_step, world, warm, total), rel_tol=1e-9, abs_tol=1e-12), \
                    (world, opt_step, used)
                p.grad = torch.zeros(1)
                opt.step()
            sch.step()
        sch.gradient_state._set_sync_gradients(True)


def test_hostplan_remembers_per_tensor_object_and_version():
    """hostplan.PLAN: a host-side fact is reused only for the very same, unmodified tensor object."""
    import gc
    from vggt_qwen3_amd.hostplan import HostPlan
    plan, calls = HostPlan(capacity=4), []

    def fact(t):
        return plan.get("nnz", (t,), lambda: (calls.append(1), int((t != 0).sum()))[1])

    a = torch.tensor([1, 0, 2, 0])
    assert fact(a) == 2 and fact(a) == 2 and len(calls) == 1          # second look is free
    a[1] = 5                                                          # in-place write bumps _version: recomputed
    assert fact(a) == 3 and len(calls) == 2
    b = a.clone()                                                     # equal contents, another object: its own entry
    assert fact(b) == 3 and len(calls) == 3
    ida = id(a)
    del a
    gc.collect()
    c = torch.tensor([0, 0, 0, 7])
    assert fact(c) == 1                                               # (even if the allocator hands out the same id, the dead weakref misses)
    for _ in range(10):                                               # capacity: the table is cleared, never grows without bound
        fact(torch.zeros(3))
    assert len(plan._d) <= 4
    assert ida is not None


def test_trainer_batch_merge_helpers():
    """Stage1Trainer._mergeable / _merge on CPU tensors: what may share a pass, what the merged dict looks like, and the reuse of the
    merged tensors for the same group of batch OBJECTS."""
    from types import SimpleNamespace
    from vggt_qwen3_amd.trainer import Stage1Trainer

    def mk(B, L=6, geom=True, views=1):
        d = {"pixel_values": torch.rand(B, views, 3, 8, 8), "input_ids": torch.randint(0, 9, (B, L)),
             "attention_mask": torch.ones(B, L, dtype=torch.long), "labels": torch.full((B, L), -100)}
        if geom:
            d["geom_token"] = {"R": torch.rand(B, views, 9), "t": torch.rand(B, views, 3)}
        return d

    a, b, c = mk(2), mk(3), mk(2)
    assert Stage1Trainer._mergeable([a, b, c])                        # different sample counts are fine
    assert not Stage1Trainer._mergeable([a, mk(2, L=7)])              # different padded length
    assert not Stage1Trainer._mergeable([a, mk(2, views=2)])          # different view count
    assert not Stage1Trainer._mergeable([a, mk(2, geom=False)])       # geometry tokens on one side only
    holder = SimpleNamespace(_merge_cache=None)
    merged, sizes = Stage1Trainer._merge(holder, [a, b, c])
    assert sizes == [2, 3, 2] and merged["input_ids"].shape == (7, 6) and merged["pixel_values"].shape[0] == 7
    assert torch.equal(merged["labels"][2:5], b["labels"]) and torch.equal(merged["geom_token"]["R"][5:], c["geom_token"]["R"])
    again, _ = Stage1Trainer._merge(holder, [a, b, c])
    assert again is merged                                            # same batch objects: same merged tensors (memoised host facts hit)
    other, _ = Stage1Trainer._merge(holder, [a, c, b])
    assert other is not merged and torch.equal(other["input_ids"][2:4], c["input_ids"])


def test_trainer_pass_sizes_cut_a_window_into_equal_passes():
    """Stage1Trainer.pass_size: what is left of an accumulation window goes into the fewest passes of at most `text_group` micro-batches,
    equal up to one - never a short last pass (a pass of 2 micro-batches is the one-micro-batch regime again), never across the window."""
    from types import SimpleNamespace
    from vggt_qwen3_amd.trainer import Stage1Trainer

    def passes(window, cap):
        tr, out, left = SimpleNamespace(text_group=cap), [], window
        while left:
            g = Stage1Trainer.pass_size(tr, left)
            assert 1 <= g <= min(cap, left)
            out.append(g)
            left -= g
        return out

    assert passes(32, 10) == [8, 8, 8, 8]
    assert passes(20, 10) == [10, 10]
    assert passes(18, 10) == [9, 9]
    assert passes(21, 10) == [7, 7, 7]
    assert passes(8, 10) == [8] and passes(1, 10) == [1] and passes(5, 1) == [1] * 5
    assert passes(5, 3) == [3, 2]                       # (tests/test_trainer_gpu.py::test_merged_micro_batches_match_one_by_one runs this one)
    for w in range(1, 70):
        for cap in (1, 3, 8, 10, 16):
            p = passes(w, cap)
            assert sum(p) == w and max(p) - min(p) <= 1 and len(p) == -(-w // cap)


def test_lm_head_dgrad_slice_rule():
    """qwen3.py: _lm_dgrad_slices - K slices of the lm_head's input-gradient product are whole 64-element K tiles (a divisor of Vp / 64),
    near one workgroup per CU; no such divisor (or VQ3_LMHEAD_DGRAD_ATOMIC=1) -> 1 = the f32-atomic split takes over."""
    import os
    from vggt_qwen3_amd.qwen3 import Qwen3ForCausalLM as Q
    for Vp, ntile in ((152000, 40), (152000, 20), (2048, 20), (320, 2), (151936, 40), (64 * 97, 4)):
        S = Q._lm_dgrad_slices(Vp, ntile)
        assert S >= 1 and (S == 1 or (Vp % (64 * S) == 0 and S <= 128)), (Vp, ntile, S)
    assert Q._lm_dgrad_slices(152000, 40) == 5 and Q._lm_dgrad_slices(2048, 20) == 8
    assert Q._lm_dgrad_slices(64 * 97, 4) in (1, 97)          # a prime number of K tiles: one slice per tile, or none
    assert Q._lm_dgrad_slices(100, 4) == 1                     # not a multiple of 64
    os.environ["VQ3_LMHEAD_DGRAD_ATOMIC"] = "1"
    try:
        assert Q._lm_dgrad_slices(152000, 40) == 1
    finally:
        del os.environ["VQ3_LMHEAD_DGRAD_ATOMIC"]


def test_bench_self_launch_command_line(monkeypatch):
    """bench.py: self_launch - `python bench.py --gpus N` without WORLD_SIZE becomes `python -m torch.distributed.run --nnodes=1
    --nproc-per-node=N --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` as a CHILD process (never an exec)."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("vq3_bench_cpu", root / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "8"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [str(root / "bench.py"), "--gpus", "4", "--steps", "8"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
