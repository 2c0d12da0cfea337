"""The REAL data-parallel path with more than one rank: two processes drive Stage1Trainer.micro_step itself (forward, HIP
backward with the bucket hooks, all-reduce from inside the backward, clipping, fused AdamW) on the one GPU of the box,
exchanging gradients over gloo (RCCL refuses two ranks on one device; dp.allreduce_tensor stages through the host for gloo).
Reference behaviour being matched: src/train/train_sft.py:208-220 under DDP - every rank runs its own micro-batches, the
gradients are averaged over ranks once per accumulation window, every rank applies the same update.

Checked: both ranks end with bit-identical weights; they equal a single-process run over the concatenated micro-batch
sequence (grad_accum x world) to bf16-accumulation tolerance; the bucket hook fires exactly once per bucket and only on the
boundary micro-batch; geom_head gradients are reduced; a rank whose micro-batch has no labelled token issues the same
collectives as the others (no hang) and contributes zero."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.golden_io import load, meta

pytestmark = pytest.mark.gpu
KW = dict(lr=2e-3, proj_lr=2e-3, weight_decay=0.1, warmup_ratio=0.0, max_steps=50, bucket_layers=1, max_grad_norm=1.0,
          accelerate_scheduler_rule=False, eps=1e-3)   # eps >> bf16 noise of the gradients: updates comparable element-wise


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(z, rank):
    """Four micro-batches per rank; they differ between ranks (labels), and each rank has one WITHOUT any label: rank 1 on
    the boundary micro-batch of window 1, rank 0 on the first micro-batch of window 2."""
    geom = {k: torch.from_numpy(z["geom:" + k]).cuda() for k in ("R", "t", "K", "depth_hist")}
    base = {"pixel_values": torch.from_numpy(z["pixel_values"].astype(np.float32)).cuda(), "geom_token": geom,
            "input_ids": torch.from_numpy(z["input_ids"]).cuda(), "attention_mask": torch.from_numpy(z["attention_mask"]).cuda(),
            "labels": torch.from_numpy(z["labels"]).cuda()}
    out = []
    for mi in range(4):
        lab = base["labels"].clone()
        rows = lab.shape[0]
        lab[(rank + mi) % rows] = -100                          # drop one row's labels: different gradients per rank / step
        if (rank, mi) in ((1, 1), (0, 2)):
            lab[:] = -100
        out.append(dict(base, labels=lab))
    return out


def _worker(rank, world, port, outdir, merge=False, mode="allreduce"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VQ3_DP_MODE=mode)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_parity_gpu import _build_vlm
        from vggt_qwen3_amd import dp
        from vggt_qwen3_amd.trainer import Stage1Trainer
        z = load("vlm_tiny.npz")
        model = _build_vlm(z, meta(z)).train()
        if merge:
            from tests.test_trainer_gpu import _Tower
            model.vision_model = _Tower(model.vision_model.agg)          # a tower that follows its input's batch axis
        tr = Stage1Trainer(model, grad_accum=2, **KW)
        assert tr.dist_on and tr.world == 2 and len(tr.buckets) == model.text_model.config.num_hidden_layers and tr.dp_mode == mode
        fired, spans = [], []
        orig_done, orig_ar = tr._layer_done, dp.allreduce_tensor
        tr._layer_done = lambda i: (fired.append((tr.micro, i)), orig_done(i))[1]
        dp.allreduce_tensor = lambda t, group=None: (spans.append((tr.micro, t.numel())), orig_ar(t, group=group))[1]
        losses = []
        bs = _batches(z, rank)
        for i, b in enumerate(bs):
            # merge: the two micro-batches of a window run as one pass (micro_step(upcoming=...)); the hooks then fire in that pass
            losses.append(float(tr.micro_step(b, upcoming=bs[i + 1:] if merge else None).item()))
        if mode == "sharded":               #  the public save refuses a state whose other shards are stale, on every rank
            from vggt_qwen3_amd import checkpoint
            assert not tr._shards_gathered
            with pytest.raises(RuntimeError):
                checkpoint.save_trainer_state(tr, os.path.join(outdir, f"stale{rank}"))
        tr.gather_sharded_state()           # (sharded mode: the fp32 state is current inside each rank's shards only)
        if mode == "sharded":
            assert tr._shards_gathered
            checkpoint.save_trainer_state(tr, os.path.join(outdir, f"state{rank}"))
        torch.cuda.synchronize()
        torch.save({"master": tr.master.cpu(), "geom_master": tr.geom_master.cpu(), "flat_w": model.text_model.flat_w.cpu(),
                    "fired": fired, "spans": spans, "losses": losses, "opt_step": tr.opt_step,
                    "nbuckets": len(tr.buckets), "geom_n": tr.geom_grad.numel()}, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("merge,mode", [(False, "allreduce"), (True, "allreduce"), (False, "sharded"), (True, "sharded")])
def test_stage1_trainer_two_ranks_equal_single_rank(tmp_path, merge, mode):
    """mode "sharded" (dp.py: reduce-scatter of every bucket, clipping + AdamW on each rank's half, all-gather of the updated weights)
    must leave both ranks with the same weights as the all-reduce mode's replicated step - and so as the single-process run."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), merge, mode), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt", weights_only=False) for i in range(world)]
    # identical replicas after two optimiser steps
    assert r[0]["opt_step"] == r[1]["opt_step"] == 2
    assert torch.equal(r[0]["master"], r[1]["master"]) and torch.equal(r[0]["flat_w"], r[1]["flat_w"])
    assert torch.equal(r[0]["geom_master"], r[1]["geom_master"])
    for i in range(world):
        nb = r[i]["nbuckets"]
        # hooks: only on the boundary micro-batches (micro index 1 and 3), once per layer each (bucket_layers = 1)
        # (merged: the pass that starts at micro index 0 / 2 contains the boundary micro-batch)
        bm = (0, 2) if merge else (1, 3)
        assert sorted(r[i]["fired"]) == sorted([(m, l) for m in bm for l in range(nb)]), (i, r[i]["fired"])
        # collectives per optimiser step: one per layer bucket + embedding + geom_head (+count slot), same on both ranks
        # (merged: the bucket all-reduces fire in the pass, at micro index 0; the embedding / geom_head ones with the optimiser step,
        # which runs with the call that returns the window's LAST loss, micro index 1)
        per_step = [s for s in r[i]["spans"] if s[0] in ((bm[0], bm[0] + 1) if merge else (bm[0],))]
        if mode == "allreduce":
            assert len(per_step) == nb + 2 and any(n == r[i]["geom_n"] for _, n in per_step), per_step
        else:                     # sharded: only the replicated tails, the geom_head gradient and the norm scalar are all-reduced
            assert any(n == r[i]["geom_n"] for _, n in per_step) and any(n == 1 for _, n in per_step), per_step
    assert [s[1] for s in r[0]["spans"]] == [s[1] for s in r[1]["spans"]]
    assert np.isnan(r[1]["losses"][1]) and np.isnan(r[0]["losses"][2]) and np.isfinite(r[0]["losses"][0])

    # single process, same micro-batches in rank-interleaved order, grad_accum x world
    from tests.test_parity_gpu import _build_vlm
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z = load("vlm_tiny.npz")
    model = _build_vlm(z, meta(z)).train()
    if merge:
        from tests.test_trainer_gpu import _Tower
        model.vision_model = _Tower(model.vision_model.agg)
    tr = Stage1Trainer(model, grad_accum=4, **KW)
    w0, g0 = tr.master.clone(), tr.geom_master.clone()
    b0, b1 = _batches(z, 0), _batches(z, 1)
    for mi in range(4):
        tr.micro_step(b0[mi])
        tr.micro_step(b1[mi])
    assert tr.opt_step == 2
    d_single = (tr.master - w0).cpu()
    d_dp = r[0]["master"] - w0.cpu()
    assert d_dp.abs().max() > 0
    e = ((d_dp - d_single).norm() / d_single.norm()).item()
    assert e < 5e-2, f"two-rank update differs from the single-rank update: rel err {e}"
    dg_single = (tr.geom_master - g0).cpu()
    dg_dp = r[0]["geom_master"] - g0.cpu()
    assert dg_dp.abs().max() > 0                                # geom_head gradients were reduced and applied
    assert ((dg_dp - dg_single).norm() / dg_single.norm()).item() < 5e-2


def test_bench_two_ranks_gloo_on_one_gpu():
    """bench.py's N > 1 path end to end without an 8-GPU node: two fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    the environment, as torch.distributed.run sets them; started before anything in them touches the GPU) share the box's one card and
    exchange gradients over gloo (VQ3_DIST_BACKEND=gloo; RCCL refuses two ranks on one device). Rank 0 must print exactly one JSON line
    with n_gpus = 2, a whole-job value, the schedule variants and the all-reduce fields measured with HIP events on the communication
    stream; rank 1 prints none; both exit 0 (matched collectives: a mismatch would hang until the timeout). 2 Qwen3 layers: plumbing,
    not a measurement (the line says valid: false)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    port = _free_port()
    base = dict(os.environ, VQ3_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
                HSA_ENABLE_IPC_MODE_LEGACY="0", VQ3_GEMM_AUTOTUNE_LOG="1", VQ3_GEMM_TUNE_WS_MB="0")
    base.pop("VQ3_GEMM_TUNE_FILE", None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--layers", "2",
           "--no-trim-variant", "--no-cpu-baseline"]
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              cwd=str(root), text=True) for r in range(2)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=900))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, (r, outs[r][1][-2000:])
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not lines1
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 12 and d["value"] > 0 and d["config"]["valid"] is False
    c = d["comm"]
    assert c and c["allreduce_ms_per_opt_step"] > 0 and c["allreduce_bytes_per_opt_step"] > 0 and c["bus_gb_per_s"] > 0
    assert d["accum1_variant"]["comm"]["collectives_per_opt_step"] >= 2 and d["text_group_1_variant"]["value"] > 0
    # a multi-rank job measures no GEMM candidate - no device synchronisation under in-flight collectives, and both
    # ranks make the SAME kernel choices (shipped table, then the heuristic): compared key by key from the two logs
    choices = []
    for r in range(2):
        err = outs[r][1]
        assert "[vq3 gemm autotune]" not in err, err[-1500:]
        choices.append(dict(l.split("] ", 1)[1].split(" -> ") for l in err.splitlines() if l.startswith("[vq3 gemm choice]")))
    common = set(choices[0]) & set(choices[1])      # (the ranks' batches differ: a few label-row-count buckets are seen by one rank only)
    assert len(common) >= 10 and all(choices[0][k] == choices[1][k] for k in common)


def test_bench_one_rank_over_rccl():
    """The same bench.py path over the backend the driver's multi-GPU launch uses: VQ3_FORCE_DIST=1 makes a single rank create an
    RCCL communicator (`init_process_group("nccl", device_id=...)`), issue every gradient all-reduce on the communication stream with
    HIP events around it, and take the barriers / MAX reductions of the timing contract through RCCL. One rank moves no data between
    GPUs, so this checks plumbing (stream ordering, event timing, matched collectives, clean shutdown), not bandwidth."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, VQ3_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VQ3_DIST_BACKEND", None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--layers", "2",
           "--no-trim-variant", "--no-cpu-baseline"]
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(root), text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    c = d["comm"]
    assert c and c["allreduce_ms_per_opt_step"] > 0 and c["allreduce_bytes_per_opt_step"] > 0 and c["collectives_per_opt_step"] >= 2
    assert d["accum1_variant"]["comm"]["collectives_per_opt_step"] >= 2


def test_bench_one_rank_over_rccl_sharded_mode():
    """dp_mode "sharded" through RCCL itself. With VQ3_FORCE_DIST=1 a single rank keeps the sharded mode (it used to
    fall back to "allreduce" silently at world 1), so RCCL's reduce_scatter_tensor / all_gather_into_tensor run in the in-place forms
    dp.py uses (output = a slice of the input / input = a slice of the output) on a GPU, on the communication stream, with HIP events
    around them; the job must finish with a finite loss, matched collectives and the comm fields filled. One rank moves no bytes between
    GPUs: plumbing, not bandwidth - no multi-GPU number exists."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, VQ3_FORCE_DIST="1", VQ3_DP_MODE="sharded", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1",
               RANK="0", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VQ3_DIST_BACKEND", None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--layers", "2",
           "--no-trim-variant", "--no-cpu-baseline", "--no-variants"]
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(root), text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["dp_mode"] == "sharded" and d["loss"] == d["loss"]
    c = d["comm"]
    assert c and c["allreduce_ms_per_opt_step"] > 0 and c["allreduce_bytes_per_opt_step"] > 0 and c["collectives_per_opt_step"] >= 2


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's OWN multi-GPU route (train_sft.py:119-133,165-170,217): model wrapped by accelerator.prepare in
# DistributedDataParallel(find_unused_parameters=True), loss.backward() per micro-batch. The text parameters never pass through
# autograd here, so DDP cannot reduce them: VGGTQwen3VLM reduces the flat gradient itself (autograd_dp = "allreduce", the default)
# or raises ("raise") - never silently divergent replicas.
def _ddp_route_worker(rank, world, port, outdir, wrap):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_parity_gpu import _build_vlm
        z = load("vlm_tiny.npz")
        model = _build_vlm(z, meta(z)).train()
        tm = model.text_model
        bs = _batches(z, rank)[:2]                            # (rank 1's second micro-batch has no labelled token at all)
        kw = lambda b: dict(images=b["pixel_values"], geom_token=b["geom_token"], input_ids=b["input_ids"],
                            attention_mask=b["attention_mask"], labels=b["labels"])
        # this rank's own gradient of the two micro-batches, no exchange: the reference point
        model.autograd_dp = "local"
        model.zero_grad(set_to_none=True)
        for b in bs:
            model(**kw(b)).backward()          # (a micro-batch without labels: NaN loss like the reference, ZERO gradient - vlm._backward_text)
        local = tm.flat_g.float().cpu().clone()
        geom_local = [p.grad.float().cpu().clone() for p in model.geom_head.parameters()]
        # the guarded mode
        model.autograd_dp = "raise"
        model.zero_grad(set_to_none=True)
        with pytest.raises(RuntimeError, match="Stage1Trainer"):
            model(**kw(bs[0])).backward()
        # the reducing mode, plain or under the reference's DDP wrapper
        model.autograd_dp = "allreduce"
        model.zero_grad(set_to_none=True)
        net = model
        if wrap:
            from torch.nn.parallel import DistributedDataParallel as DDP
            net = DDP(model, find_unused_parameters=True)
        for b in bs:
            net(**kw(b)).backward()
        torch.cuda.synchronize()
        torch.save({"local": local, "reduced": tm.flat_g.float().cpu(), "geom_local": geom_local,
                    "geom_reduced": [p.grad.float().cpu() for p in model.geom_head.parameters()],
                    "grad_is_view": next(iter(tm.parameters())).grad.data_ptr() == tm.grad_views["model.embed_tokens.weight"].data_ptr()},
                   os.path.join(outdir, f"ddp{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wrap", [False, True])
def test_autograd_route_reduces_text_gradients_across_ranks(tmp_path, wrap):
    port = _free_port()
    mp.start_processes(_ddp_route_worker, args=(2, port, str(tmp_path), wrap), nprocs=2, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"ddp{k}.pt") for k in range(2)]
    want = 0.5 * (r[0]["local"] + r[1]["local"])              # DDP's average of the ranks' window gradients
    assert want.abs().max() > 0
    for k in range(2):
        assert r[k]["grad_is_view"]
        e = ((r[k]["reduced"] - want).norm() / want.norm()).item()
        assert e < 2e-2, f"rank {k}: reduced text gradient differs from the average of the ranks' gradients: {e}"
    assert torch.equal(r[0]["reduced"], r[1]["reduced"])      # replicas hold the same gradient, bit for bit
    assert (r[0]["local"] - r[1]["local"]).abs().max() > 0    # (and the ranks' own gradients did differ)
    if wrap:                                                   # geom_head: ordinary autograd leaves - DDP averaged them
        for a0, a1, g0, g1 in zip(r[0]["geom_local"], r[1]["geom_local"], r[0]["geom_reduced"], r[1]["geom_reduced"]):
            assert torch.allclose(g0, g1, rtol=1e-5, atol=1e-7)
            assert torch.allclose(g0, 0.5 * (a0 + a1), rtol=2e-2, atol=1e-4 * float((a0.abs().max() + a1.abs().max())))


def test_bench_self_launch_two_ranks_gloo_on_one_gpu():
    """plain `python3 bench.py --gpus 2` - no WORLD_SIZE / RANK in the environment, the way the driver calls
    `--gpus 1` - starts its two ranks itself (torch.distributed.run as a child process, before anything touches the GPU) and rank 0's
    one JSON line comes through with n_gpus = 2 and the communication fields filled. gloo rehearsal backend: both ranks share the
    box's one card. 2 Qwen3 layers: plumbing, the line says valid: false."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, VQ3_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", VQ3_GEMM_TUNE_WS_MB="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQ3_GEMM_TUNE_FILE", "VQ3_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--layers", "2",
           "--no-trim-variant", "--no-cpu-baseline", "--no-variants"]
    pr = subprocess.run(cmd, env=env, cwd=str(root), capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-3000:]
    assert "launching 2 ranks" in pr.stderr
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 12 and d["value"] > 0
    assert d["comm"] and d["comm"]["allreduce_ms_per_opt_step"] > 0 and d["comm"]["bus_gb_per_s"] > 0
    # a WORLD_SIZE that contradicts --gpus is a clear error, not an AssertionError deep in the run
    bad = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         cwd=str(root), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr and "Traceback" not in bad.stderr
