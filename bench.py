#!/usr/bin/env python3
"""Stage-1 training throughput of the VGGT -> Perceiver -> Qwen3-4B path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one micro-batch of B=6 synthetic samples per GPU through the whole hot path: VGGT aggregator forward
(frozen), Perceiver forward (no grad, as in the reference), Qwen3-4B forward + backward, gradient all-reduce over
RCCL, fused AdamW with fp32 master weights, refresh of the transposed weight copies. The timed window is whole
accumulation cycles: grad_accum defaults to min(32, --steps) (the reference's stage-1 schedule is 32), so every
window contains exactly steps/grad_accum optimiser steps + all-reduces and never zero; --grad-accum 1 puts them in
every step. The trainer runs consecutive micro-batches of a window as ONE forward/backward pass (Stage1Trainer.pass_size():
10 micro-batches = 60 samples per pass at the default text_group, so the driver's 20-step window is 2 passes and the reference's
window of 32 is cut 8 + 8 + 8 + 8; each micro-batch's loss is normalised by its own labelled rows - the reference's arithmetic,
tests/test_trainer_gpu.py); steps and ms_per_step still count micro-batches. The line's config.micro_batches_per_pass says what ran.
`python bench.py --gpus N` without WORLD_SIZE in the environment starts its N ranks itself (self_launch). Weights are random-init at the exact
Qwen3-4B / VGGT-1B / Perceiver shapes, inputs synthetic and already resident in HBM. Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch
import torch.distributed as dist

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X bf16 dense MFMA peak (MI355X_MICROARCH.md)

# algorithmic FLOPs per sample
def vit_block(tok, ctx, d=1024, mlp=4): return 2 * tok * (4 * d * d + 2 * mlp * d * d), 4 * tok * ctx * d
def flops_vggt(S, img=448, p=14, depth=24):
    P = (img // p) ** 2 + 5; pe = 2 * (img // p) ** 2 * 3 * p * p * 1024
    dl, da = vit_block(P, P); fl, fa = vit_block(P, P); gl, ga = vit_block(P, S * P)
    return S * (depth * (dl + da) + pe + depth * (fl + fa) + depth * (gl + ga))
def flops_perceiver(T=128, N=128, i=2048, D=4096, F=16384, L=6, o=2560):
    return 2 * T * i * D + L * (2 * N * D * D + 4 * T * D * D + 4 * N * T * D + 2 * N * D * D + 4 * N * D * F) + 2 * N * D * o
def flops_qwen(L, H=2560, nq=32, nkv=8, hd=128, I=9728, n=36, V=151937, head_rows=None):
    """head_rows: rows the lm_head actually runs on (None = all L positions, as the reference materialises them)."""
    lin = 2 * (H * nq * hd + 2 * H * nkv * hd + nq * hd * H + 3 * H * I)
    return n * (L * lin + 4 * L * L * nq * hd / 2) + 2 * (L if head_rows is None else head_rows) * H * V


def synthetic_batch(B, V, L, img, vocab, image_id, pad_id, nl_id, seed, device, geom: bool):
    """layout [q tokens, "\\n", <image>, "\\n", a tokens, pad...], labels only on the answer."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.rand(B, V, 3, img, img, generator=g)
    ids = torch.full((B, L), pad_id, dtype=torch.long)
    labels = torch.full((B, L), -100, dtype=torch.long)
    for b in range(B):
        q = int(torch.clamp(torch.normal(14.0, 5.0, (1,), generator=g), 6, 40).item())
        a = int(torch.randint(1, 5, (1,), generator=g).item())
        def draw(n):
            t = torch.randint(0, vocab, (n,), generator=g)
            t[(t == pad_id) | (t == image_id) | (t == nl_id)] = 11
            return t
        seq = torch.cat([draw(q), torch.tensor([nl_id, image_id, nl_id]), draw(a)])
        ids[b, : len(seq)] = seq
        labels[b, q + 3: q + 3 + a] = seq[q + 3:]
    batch = {"pixel_values": pix.to(device), "input_ids": ids.to(device),
             "attention_mask": (ids != pad_id).long().to(device), "labels": labels.to(device), "geom_token": None}
    if geom:
        dh = torch.rand(B, V, 16, generator=g)
        batch["geom_token"] = {"R": torch.randn(B, V, 9, generator=g).to(device), "t": torch.randn(B, V, 3, generator=g).to(device),
                               "K": torch.randn(B, V, 9, generator=g).to(device), "depth_hist": (dh / dh.sum(-1, keepdim=True)).to(device),
                               "mask": torch.ones(B, dtype=torch.bool, device=device)}
    return batch


class BatchStream:
    """What a DataLoader hands the trainer: a stream of DISTINCT micro-batches, each a dict of fresh tensor objects. A pool of
    `npool` synthetic batches (seeds seed0 .. seed0 + npool - 1: different images, ids and label counts) is rotated; the integer
    tensors of every draw are new objects (clones), so nothing memoised per tensor object (hostplan.PLAN: <image> positions,
    labelled rows, live key tiles; Stage1Trainer._merge: the concatenated pass) can hit - each pass pays its device -> host
    reads and its torch.cat, as a real data stream makes it. Inputs stay resident in HBM (the contract's timed region)."""

    def __init__(self, npool, make):
        self.pool = [make(i) for i in range(npool)]
        self.queue = []
        self.n = 0

    def _draw(self):
        b = self.pool[self.n % len(self.pool)]
        self.n += 1
        d = dict(b)
        for k in ("input_ids", "attention_mask", "labels"):
            d[k] = b[k].clone()
        return d

    def peek(self, n):
        while len(self.queue) < n:
            self.queue.append(self._draw())
        return self.queue[:n]

    def pop(self):
        self.peek(1)
        return self.queue.pop(0)

    def drop(self):
        self.queue = []


def cpu_baseline(model, batch, L):
    """The CPU oracle (PyTorch-CPU restatement of the reference's path, oracle/) timed on this host's cores on ONE WHOLE training sample of
    the workload, nothing extrapolated: the 72-block tower forward (bf16, no_grad), the 6-layer Perceiver forward (fp32, no_grad: the
    reference never back-propagates into it), the splice, all 36 Qwen3-4B layers + full-vocabulary lm_head + shifted CE forward AND
    backward (bf16 parameters, torch autograd) - what VGGTQwen3VLM.forward + loss.backward() cost the reference on CPU (config C1's
    arithmetic: vggt_qwen3_vlm.py:128-201, train_sft.py:217). No optimiser step (it would add 16 B / parameter of host traffic, not
    arithmetic of the path). Reported baseline only; about 20-30 s on a 128-core host."""
    from oracle import perceiver as operc, qwen3 as oq, vggt as ov, vlm as ovlm
    nthreads = torch.get_num_threads()
    t_all = {}
    tm = model.text_model
    agg = model.vision_model.aggregator
    img = batch["pixel_values"][:1].cpu()
    ids, mask, labels = batch["input_ids"][:1].cpu(), batch["attention_mask"][:1].cpu(), batch["labels"][:1].cpu()
    vsd = {n: t.detach().cpu() for n, t in agg.named_tensors().items()}
    psd = {k: v.detach().float().cpu() for k, v in model.projector.state_dict().items()}
    tsd = {n: p.detach().cpu().requires_grad_(True) for n, p in tm.named_parameters() if n != "lm_head.weight"}
    cfg = oq.Qwen3Cfg(num_hidden_layers=tm.config.num_hidden_layers, vocab_size=tm.vocab)
    t_start = time.perf_counter()
    with torch.no_grad():
        t0 = time.perf_counter()
        last = ov.aggregator(img, vsd, num_heads=agg.num_heads, depth=agg.depth, dino_depth=agg.dino_depth)[-1]
        t_all["vggt_fwd"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        vis = operc.projector(ovlm.select_tokens(last, model.num_vis_tokens).float(), psd, model.projector.cfg.num_heads, model.projector.cfg.num_layers)
        t_all["perceiver_fwd"] = time.perf_counter() - t0
    del vsd, psd
    t0 = time.perf_counter()
    emb = torch.nn.functional.embedding(ids, tsd["model.embed_tokens.weight"])
    feats = vis
    if batch.get("geom_token") and model.geom_tokens:
        gsd = {"geom_head." + k: v.detach().float().cpu() for k, v in model.geom_head.state_dict().items()}
        g = ovlm.encode_geom({k: v[:1].float().cpu() for k, v in batch["geom_token"].items() if k != "mask"}, gsd, model.geom_tokens)
        feats = torch.cat([g, vis], dim=1)
    emb = ovlm.splice(emb, ids, feats.to(emb.dtype), model.image_id)
    loss, _ = oq.causal_lm(emb, mask, labels, tsd, cfg)
    t_all["qwen_fwd"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    loss.backward()
    t_all["qwen_bwd"] = time.perf_counter() - t0
    per_sample = time.perf_counter() - t_start
    V = batch["pixel_values"].shape[1]
    return {"value": 1.0 / per_sample, "unit": "samples/s", "cores": nthreads, "kind": "port",
            "sample": ("1 whole sample, measured end to end (no extrapolation): VGGT %d+%d+%d blocks fwd (bf16, %d view(s)) + Perceiver fwd (fp32) + "
                       "Qwen3-4B %d layers + full-vocabulary lm_head + CE fwd+bwd (bf16, L=%d), torch autograd on the oracle; loss %.4f; parts(s)=%s"
                       % (agg.dino_depth, agg.depth, agg.depth, V, cfg.num_hidden_layers, L, float(loss.detach()), {k: round(v, 2) for k, v in t_all.items()}))}


def self_launch(n: int) -> int:
    """Runs this very command line under torch.distributed.run with n local ranks (rendezvous on 127.0.0.1, a free port) as a child
    process and returns its exit code. Replaces what the reference gets from `accelerate launch --config_file accelerate_8gpu.yaml`
    (/root/reference/src/train/train_sft.py:119-133, configs/accelerate_8gpu.yaml)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE in the environment: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)   # rounded up to whole accumulation windows
    ap.add_argument("--grad-accum", type=int, default=0,
                    help="micro-batches per optimiser step; 0 = min(32, --steps): the timed window is one accumulation "
                         "cycle of the reference's stage-1 schedule (grad_accum 32) with exactly one all-reduce + AdamW")
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--seq-len", type=int, default=200)
    ap.add_argument("--image-size", type=int, default=448)
    ap.add_argument("--geom", action="store_true", help="geometry tokens on (config C4)")
    ap.add_argument("--fp8", action="store_true",
                    help="config C5: Qwen3 forward projections in e4m3 (block-scaled MFMA), backward bf16; NOT the default")
    ap.add_argument("--train-projector", action="store_true",
                    help="the 'corrected' mode (VisionLanguageConfig.train_projector): the Perceiver gets a gradient and an AdamW "
                         "group; NOT the default - the reference runs it under no_grad (vggt_qwen3_vlm.py:128,162)")
    ap.add_argument("--trim-pad", action="store_true",
                    help="drop the all-padding tail of the batch (exact; NOT the default: fewer FLOPs are executed)")
    ap.add_argument("--vision-prefetch", action="store_true",
                    help="run the (frozen) vision tower one micro-batch ahead on a second stream (measured: no gain "
                         "on a saturated GPU; off by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trim-variant", action="store_true", help="skip the second timed window (exact padding shortcut)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra result objects (forward_only, c4_variant, c5_variant); they run at --gpus 1 only")
    ap.add_argument("--variant-steps", type=int, default=8, help="timed steps of the 8-view C4 / C5 variant windows (one accumulation cycle "
                                                                     "= one merged pass at the default text_group)")
    ap.add_argument("--layers", type=int, default=36, help="debug only; anything but 36 marks the line invalid")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, the way the driver's launcher does
        # (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`).
        # This process has not touched the GPU (nothing before this line makes a HIP call) and never does: it waits for the
        # launcher - a CHILD process, never an exec - and leaves with its exit code. Rank 0's JSON line goes through on stdout.
        return self_launch(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VQ3_DIST_BACKEND", "nccl") != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = bool(os.environ.get("VQ3_FORCE_DIST"))   # exercise the RCCL path even with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("VQ3_DIST_BACKEND", "nccl")   # "gloo" only to rehearse N ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment (launch with --nproc-per-node {args.gpus}, "
                         "or unset WORLD_SIZE and let bench.py start its ranks itself)")

    from vggt_qwen3_amd import ops
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.trainer import Stage1Trainer
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
    import yaml

    t_build = time.perf_counter()
    qcfg = Qwen3Config.qwen3_4b()
    qcfg.num_hidden_layers = args.layers
    pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
    vcfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128,
                                geom_tokens=8 if args.geom else 0, projector_cfg=pcfg, text_config=qcfg,
                                device=str(dev), seed=0, trim_padding=args.trim_pad, fp8_text_forward=args.fp8,
                                train_projector=args.train_projector)
    model = VGGTQwen3VLM(vcfg)
    model.train()
    accum = args.grad_accum if args.grad_accum > 0 else max(1, min(32, args.steps))
    # the timed window is cut into whole accumulation cycles (the last one shorter if --steps is not a multiple): every
    # timed micro-batch belongs to a cycle that ends with its all-reduce + optimiser step inside the window
    cycles = [accum] * (args.steps // accum) + ([args.steps % accum] if args.steps % accum else [])
    trainer = Stage1Trainer(model, grad_accum=accum, max_steps=30000)
    # as Stage1Trainer.fit() runs it: a window's optimiser step is enqueued on a side stream and the next window's first pass starts with
    # the frozen vision tower beside it (nothing to overlap with in a timed region of ONE window, e.g. the driver's --steps 20: the
    # barrier + synchronize in front of the clock waits for the warm-up window's step, the one behind it for the timed window's)
    trainer.overlap_optimizer = os.environ.get("VQ3_OPT_OVERLAP", "1") != "0"
    B, V, L = args.batch, args.views, args.seq_len
    # >= 2 x text_group distinct micro-batches per rank (seeds 1234 + 1000 * rank + i), rotated as a data stream
    npool = max(2, 2 * trainer.text_group)
    stream = BatchStream(npool, lambda i: synthetic_batch(B, V, L, args.image_size, 151936, model.image_id, 151643, 198,
                                                          1234 + 1000 * rank + i, dev, args.geom))
    batch = stream.pool[0]
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    # labelled rows per sample: the lm_head + CE run on these rows only (identical loss and gradients),
    # so utilisation figures count 2*rows*H*V for the head, not the reference's 2*L*H*V (mean over the pool's batches)
    head_rows = sum(float((b["labels"][:, 1:] != -100).sum().item()) for b in stream.pool) / (B * npool)

    use_dist = world > 1 or force_dist

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        """MAX of a host-side duration over the ranks (the contract's timing rule); the gloo rehearsal backend reduces a host tensor."""
        if not use_dist:
            return seconds
        on_dev = dist.get_backend() != "gloo"
        t = torch.tensor([seconds], device=dev if on_dev else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    loss = None
    # warm-up runs whole accumulation cycles too (so the timed region starts on a cycle boundary); the first cycle
    # also pays the one-time costs (LDS attribute setup, allocator growth, RCCL channel setup)
    # (and, with the weight-gradient GEMMs of `wgrad_defer` micro-batches fused into one, the autotuner's first sight of every
    # contraction length the window's flush pattern produces: the warm-up is rounded UP to whole windows of the timed schedule)
    nwarm = (max(args.warmup, 1) + accum - 1) // accum * accum
    trainer.grad_accum = accum

    def run_window(clen, src=None):
        """One accumulation window of `clen` micro-batches drawn from the stream: forward/backward passes (merged per text_group),
        the gradient all-reduce and the optimiser step. The look-ahead never crosses the window's end."""
        src = stream if src is None else src
        vg = max(1, trainer.vision_group, trainer.text_group)
        trainer.grad_accum, trainer.micro = clen, 0
        out = None
        for i in range(clen):
            b = src.pop()
            ahead = src.peek(min(vg - 1, clen - 1 - i))
            nxt = ahead[0] if (args.vision_prefetch and ahead) else None   # next micro-batch's (frozen) vision forward overlaps this one
            out = trainer.micro_step(b, nxt, upcoming=ahead)
        assert not trainer._merged_pending and not trainer._opt_due
        return out

    for _ in range(nwarm // accum):
        loss = run_window(accum)
    sync()
    assert not model._vis_group, "no precomputed vision result may cross into the timed region"
    stream.drop()
    t0 = time.perf_counter()
    for clen in cycles:
        loss = run_window(clen)       # K vision forwards (in groups) + K text fwd/bwd + len(cycles) all-reduces / AdamW steps in the window
    sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # ---- the same K steps once more with the exact padding shortcut ("if the implementation skips padded
    # rows ... recompute the figure for the tokens actually executed and report both"). `value` above is the dense run;
    # this is reported beside it, never instead of it. Loss and gradients are identical (tests/test_fullsize_gpu.py).
    trimmed = None
    if not args.trim_pad and not args.no_trim_variant:
        model.trim_padding = True
        loss_t = run_window(cycles[0])        # warm the trimmed shapes over one whole window
        sync()
        t0 = time.perf_counter()
        for clen in cycles:
            loss_t = run_window(clen)
        sync()
        dtt = max_over_ranks(time.perf_counter() - t0)
        L_eff = int(model._last_L) if hasattr(model, "_last_L") else None
        trimmed = {"value": round(world * B * args.steps / dtt, 3), "unit": "samples/s", "ms_per_step": round(dtt / args.steps * 1e3, 2),
                   "executed_seq_len": L_eff, "loss": round(float(loss_t.item()), 4),
                   "executed_tflop_per_sample": None if L_eff is None else round(
                       (flops_vggt(V, args.image_size) + flops_perceiver() + 3 * flops_qwen(L_eff, head_rows=head_rows)) / 1e12, 3),
                   "note": "exact: columns that are padding for every row of the batch are not computed; same loss and gradients"}
        model.trim_padding = False

    def timed_windows(clens, warm):
        """warm-up windows (new shapes meet the GEMM tuner there), then the timed ones; MAX over ranks."""
        for c in warm:
            run_window(c)
        sync()
        stream.drop()
        t0 = time.perf_counter()
        last = None
        for c in clens:
            last = run_window(c)
        sync()
        return max_over_ranks(time.perf_counter() - t0), last

    def comm_summary(prof, nopt):
        """HIP-event time of the gradient all-reduces (events on the communication stream) per optimiser step; bus bandwidth with the
        ring factor 2 (N - 1) / N, as RCCL's own tests quote it."""
        if not prof:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in prof) / nopt
        by = sum(b for _, _, b in prof) / nopt
        return {"allreduce_ms_per_opt_step": round(ms, 3), "allreduce_bytes_per_opt_step": int(by), "collectives_per_opt_step": round(len(prof) / nopt, 1),
                "bus_gb_per_s": round(2.0 * (world - 1) / world * by / (ms * 1e-3) / 1e9, 1) if ms > 0 and world > 1 else None}

    # ---- the same job under the schedules SURVEY 8(d) / the reference name: one micro-batch per forward/backward pass (what the
    # reference's loop does, train_sft.py:208-220) and one optimiser step + gradient exchange per micro-batch (C3 "accum = 1").
    # Reported beside `value`, never instead of it. Every rank runs them (the collectives stay matched).
    sched_variants = {}
    comm = None
    if use_dist:                                       # one more window of the headline schedule with events on the comm stream
        trainer.comm_profile = []
        run_window(cycles[0])
        torch.cuda.synchronize()
        comm = comm_summary(trainer.comm_profile, 1)
        trainer.comm_profile = None
    if not args.no_variants and not args.trim_pad:
        tg0 = trainer.text_group
        if tg0 > 1:
            trainer.set_schedule(text_group=1, grad_accum=accum)
            d1, l1 = timed_windows(cycles, [cycles[0]])
            sched_variants["text_group_1_variant"] = {
                "value": round(world * B * args.steps / d1, 3), "unit": "samples/s", "ms_per_step": round(d1 / args.steps * 1e3, 2),
                "micro_batches_per_pass": 1, "grad_accum": accum, "loss": round(float(l1.item()), 4),
                "note": "one micro-batch of %d samples per forward/backward pass, as the reference's loop runs them; the vision tower still "
                        "shares one pass per %d micro-batches" % (B, trainer.vision_group)}
        if accum != 32:
            # the reference's own window (configs/stage1_3d.yaml:31: grad_accum 32) under the trainer's default pass rule, whatever --steps
            # made of the headline's window: one warm-up window, one timed window = one optimiser step
            trainer.set_schedule(text_group=tg0, grad_accum=32)
            d32, l32 = timed_windows([32], [32])
            sched_variants["accum32_variant"] = {
                "value": round(world * B * 32 / d32, 3), "unit": "samples/s", "ms_per_step": round(d32 / 32 * 1e3, 2), "steps": 32, "grad_accum": 32,
                "micro_batches_per_pass": int(trainer.pass_size(32)), "loss": round(float(l32.item()), 4),
                "note": "one accumulation window of the reference's schedule (32 micro-batches of %d, one all-reduce + AdamW), cut into "
                        "passes as Stage1Trainer.pass_size() does" % B}
        n1 = min(args.steps, 16)
        trainer.set_schedule(text_group=1, grad_accum=1)
        if use_dist:
            trainer.comm_profile = []
        d1, l1 = timed_windows([1] * n1, [1, 1])
        sched_variants["accum1_variant"] = {
            "value": round(world * B * n1 / d1, 3), "unit": "samples/s", "ms_per_step": round(d1 / n1 * 1e3, 2), "steps": n1, "grad_accum": 1,
            "loss": round(float(l1.item()), 4),
            "note": "gradient all-reduce + clipping + AdamW after EVERY micro-batch (SURVEY 8(d) C3: accum = 1 beside the schedule's 32)"}
        if use_dist:
            torch.cuda.synchronize()
            sched_variants["accum1_variant"]["comm"] = comm_summary(trainer.comm_profile[2 * len(trainer.comm_profile) // (n1 + 2):], n1)
            trainer.comm_profile = None
        trainer.set_schedule(text_group=tg0, grad_accum=accum)

    def timed(fn, n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        sync()
        return (time.perf_counter() - t0) / n, out

    tf_exec_train = (flops_vggt(V, args.image_size) + flops_perceiver() + 3 * flops_qwen(L, head_rows=head_rows)) / 1e12
    tf_exec_fwd = (flops_vggt(V, args.image_size) + flops_perceiver() + flops_qwen(L, head_rows=head_rows)) / 1e12

    # ---- forward-only window (the north-star's ">= 40 % bf16 MFMA utilisation on the fused VGGT+Perceiver+
    # Qwen3-4B forward"): the reference's unit is VGGTQwen3VLM.forward under no_grad / eval (vggt_qwen3_vlm.py:179-201)
    forward_only = None
    variants = {}
    if world == 1 and not args.no_variants:
        def fwd():
            b = stream.pop()                     # a fresh micro-batch per forward (distinct inputs, new tensor objects)
            with torch.no_grad():
                return model(images=b["pixel_values"], geom_token=b["geom_token"], input_ids=b["input_ids"],
                             attention_mask=b["attention_mask"], labels=b["labels"])
        model.eval()
        fwd()
        dtf, lf = timed(fwd, max(4, min(args.steps, 16)))
        model.train()
        fps = B / dtf
        forward_only = {"forward_samples_per_s": round(fps, 2), "ms_per_forward": round(dtf * 1e3, 2), "loss": round(float(lf.item()), 4),
                        "executed_tflop_per_sample": round(tf_exec_fwd, 3),
                        "reference_algorithmic_tflop_per_sample": round((flops_vggt(V, args.image_size) + flops_perceiver() + flops_qwen(L)) / 1e12, 3),
                        "forward_mfma_frac": round(fps * tf_exec_fwd / BF16_DENSE_PEAK_TFLOPS, 4),
                        "note": "eval mode, no_grad, dense L (no padding trimmed); MFMA fraction counts executed FLOPs only "
                                "(lm_head on the %.1f labelled rows per sample)" % head_rows}

        # the same forward over 8 micro-batches at once (48 samples): what the tower / text GEMMs reach when the batch is not the limit
        if trainer.text_group > 1:
            nb = 8                                   # (a fixed 48 samples: comparable from round to round whatever the training pass size is)

            def fwd_big():
                big, _sizes = trainer._merge([stream.pop() for _ in range(nb)])      # (the torch.cat is inside the timed call)
                with torch.no_grad():
                    return model(images=big["pixel_values"], geom_token=big.get("geom_token"), input_ids=big["input_ids"],
                                 attention_mask=big["attention_mask"], labels=big["labels"])
            model.eval()
            fwd_big()
            dtb, _ = timed(fwd_big, 4)
            model.train()
            forward_only["batched_%d_micro_batches" % nb] = {
                "forward_samples_per_s": round(nb * B / dtb, 2), "ms_per_forward": round(dtb * 1e3, 2),
                "forward_mfma_frac": round(nb * B / dtb * tf_exec_fwd / BF16_DENSE_PEAK_TFLOPS, 4),
                "note": "one forward over %d samples (%d micro-batches concatenated), eval mode, no_grad" % (nb * B, nb)}

        # ---- configs C4 (8 views + geometry tokens) and C5 (C4 with the e4m3 forward) on this one GPU, same B and L
        if not args.geom and V == 1:
            stream8 = BatchStream(npool, lambda i: synthetic_batch(B, 8, L, args.image_size, 151936, model.image_id, 151643, 198,
                                                                   4321 + i, dev, True))
            hr8 = sum(float((b["labels"][:, 1:] != -100).sum().item()) for b in stream8.pool) / (B * npool)
            tf8 = (flops_vggt(8, args.image_size) + flops_perceiver() + 3 * flops_qwen(L, head_rows=hr8)) / 1e12
            model.geom_tokens, trainer.geom_on = 8, True
            for key, fp8 in (("c4_variant", False), ("c5_variant", True)):
                model.text_model.enable_fp8_forward(fp8)
                n8 = max(1, args.variant_steps)
                run_window(n8, stream8)                                  # warm the 8-view shapes over the same cycle
                assert not model._vis_group
                sync()
                t0 = time.perf_counter()
                l8 = run_window(n8, stream8)                             # one accumulation cycle: n8 micro-batches + one AdamW
                sync()
                dt8 = (time.perf_counter() - t0) / n8
                assert not model._vis_group
                variants[key] = {"value": round(B / dt8, 3), "unit": "samples/s", "ms_per_step": round(dt8 * 1e3, 2), "steps": n8,
                                 "views": 8, "geom_tokens": 8, "fp8_text_forward": fp8, "fp8_text_dgrad": bool(fp8 and model.text_model._fp8T is not None), "batch_per_gpu": B, "loss": round(float(l8.item()), 4),
                                 "executed_tflop_per_sample": round(tf8, 3),
                                 "executed_flops_utilisation_vs_bf16_peak": round(B / dt8 * tf8 / BF16_DENSE_PEAK_TFLOPS, 4),
                                 "workload": ("BASELINE config %s on 1 GPU: VGGT aggregator @%dpx x 8 views (8 232-token global attention) "
                                              "+ 8 geometry tokens + Perceiver + Qwen3-4B fwd+bwd + AdamW%s"
                                              % ("C5" if fp8 else "C4", args.image_size, "; Qwen3 forward projections in e4m3" if fp8 else ""))}
            del stream8
            model.text_model.enable_fp8_forward(args.fp8)
            model.geom_tokens, trainer.geom_on = (8 if args.geom else 0), bool(args.geom)
            trainer.geom_grad.zero_()
            if not args.fp8:
                # the e4m3 text model at ONE view (the headline's workload with C5's arithmetic): here the text model is most of the
                # step, so this is where the e4m3 kernels' worth shows
                model.text_model.enable_fp8_forward(True)
                d5, l5 = timed_windows(cycles, [cycles[0]])
                variants["c5_c2_variant"] = {"value": round(world * B * args.steps / d5, 3), "unit": "samples/s", "ms_per_step": round(d5 / args.steps * 1e3, 2),
                                             "views": V, "fp8_text_forward": True, "fp8_text_dgrad": model.text_model._fp8T is not None,
                                             "grad_accum": accum, "loss": round(float(l5.item()), 4),
                                             "workload": "the headline's workload (config C2) with the Qwen3 projections' forward and input-gradient GEMMs in e4m3"}
                model.text_model.enable_fp8_forward(False)
            # the inference row: greedy decoding of one prompt as the reference's callers run it - ms per new token
            # and the fraction of the HBM peak the weight stream reaches (every weight once per token)
            try:
                tmq = model.text_model
                demb = (torch.randn(1, 200, tmq.config.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
                dmask = torch.ones(1, 200, dtype=torch.long, device=dev)
                dkw = dict(inputs_embeds=demb, attention_mask=dmask, repetition_penalty=1.1, no_repeat_ngram_size=4)
                tmq.generate(max_new_tokens=4, **dkw)
                torch.cuda.synchronize()
                # (two runs that differ only in the number of replayed steps: prefill, the eager first step and the graph capture cancel)
                t0 = time.perf_counter(); tmq.generate(max_new_tokens=34, **dkw); torch.cuda.synchronize(); t_s = time.perf_counter() - t0
                t0 = time.perf_counter(); _, dst = tmq.generate(max_new_tokens=66, return_stats=True, **dkw); torch.cuda.synchronize()
                t_l = time.perf_counter() - t0
                per_tok = (t_l - t_s) / 32
                wbytes = 2 * sum(p.numel() for n, p in tmq.named_parameters() if n != "lm_head.weight")
                variants["decode_variant"] = {"value": round(1.0 / per_tok, 1), "unit": "tokens/s", "ms_per_token": round(per_tok * 1e3, 3),
                                              "hbm_frac": round(wbytes / per_tok / 8e12, 3), "weight_GB_per_token": round(wbytes / 1e9, 2),
                                              "persistent_layers_kernel": bool(dst["persistent"]), "graph": bool(dst["graph"]),
                                              "workload": "Qwen3-4B greedy decode, B = 1, prompt 200, tokens 35-66 timed (text_model.generate as "
                                                          "src/inference/qa_inference.py:207-216 calls it); bound: HBM, 8 TB/s"}
                del demb, dmask
            except Exception as e:  # reported, never fatal for the training line
                variants["decode_variant"] = {"error": repr(e)[:200]}

    # ---- live roofline of the dominant kernel (gemm_nt_kernel): one extra instrumented step, HIP events per launch
    roof = None
    if rank == 0:
        model.text_model._wgrad_stream = None   # serial launches: per-launch event times are not inflated by overlap
        row_split_env = os.environ.get("VQ3_VGGT_ROW_SPLIT")
        os.environ["VQ3_VGGT_ROW_SPLIT"] = "0"  # (likewise the tower's second stream: one chain per block while the events are in)
        # one group of the deferred weight-gradient schedule (every projection's weight-gradient GEMM runs over nroof micro-batches'
        # rows): per-step figures below are the group's totals / nroof
        nroof = max(1, min(int(getattr(model.text_model, "_wd_depth", 1)) * max(1, trainer.pass_size(accum)), accum))
        run_window(nroof)       # (the serial schedule's shapes are tuned before the events go in)
        torch.cuda.synchronize()
        ops.GEMM_PROFILE = []
        run_window(nroof)
        torch.cuda.synchronize()
        fl = sum(g[0] for g in ops.GEMM_PROFILE) / nroof
        by = sum(g[1] for g in ops.GEMM_PROFILE) / nroof
        ms = sum(g[2].elapsed_time(g[3]) for g in ops.GEMM_PROFILE) / nroof
        nlaunch = len(ops.GEMM_PROFILE) / nroof
        if os.environ.get("VQ3_GEMM_TABLE"):
            import collections
            tab = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for g in ops.GEMM_PROFILE:
                t = tab[g[4]]; t[0] += 1.0 / nroof; t[1] += g[2].elapsed_time(g[3]) / nroof; t[2] += g[0] / nroof
            print("GEMM shapes (M,N,K,batch): calls, ms/step, TF/s", file=sys.stderr)
            for k, (c, m_, f_) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:40]:
                print(f"  {k}: {c:6.1f} {m_:8.3f} ms {f_ / m_ / 1e9:8.1f} TF/s", file=sys.stderr)
        ops.GEMM_PROFILE = None
        if row_split_env is None:
            os.environ.pop("VQ3_VGGT_ROW_SPLIT", None)
        else:
            os.environ["VQ3_VGGT_ROW_SPLIT"] = row_split_env
        ach = fl / (ms * 1e-3) / 1e12
        # PMC counters cannot be read from inside the run: `traffic` is the rocprofv3 --pmc result of THIS command at the
        # commit named beside it (tools/summarize_pmc.py writes the file); null when no such profile has been committed
        traffic, traffic_src = None, None
        for tname in ("r5_pmc_traffic.json", "r4_pmc_traffic.json"):     # the newest committed PMC pass of this command
            tj = ROOT / "profiles" / tname
            if tj.exists():
                tjd = json.loads(tj.read_text())
                traffic = round(tjd["traffic_bytes_per_launch"])
                traffic_src = {"file": "profiles/" + tname, "measured_at_commit": tjd.get("commit"), "command": tjd.get("command")}
                break
        roof = {"bound": "mfma", "kernel": "gemm kernels behind vq3_gemm_bf16_nt (gemm_v2 / gemm_v3 / gemm_v6)", "achieved": round(ach, 1),
                "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_unit": "bytes/launch (L2<->fabric, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)",
                "algorithmic_bytes_per_launch": round(by / nlaunch), "launches_per_step": round(nlaunch, 1), "avg_launch_us": round(ms * 1e3 / nlaunch, 2),
                "gemm_ms_per_step": round(ms, 2), "gemm_tflop_per_step": round(fl / 1e12, 3)}
    elif world > 1:
        nroof = max(1, min(int(getattr(model.text_model, "_wd_depth", 1)) * max(1, trainer.pass_size(accum)), accum))
        run_window(nroof)       # keep collectives matched across ranks
        run_window(nroof)
    if use_dist:
        dist.barrier()

    if rank == 0:
        tf_train = (flops_vggt(V, args.image_size) + flops_perceiver() + 3 * flops_qwen(L)) / 1e12
        out = {
            "metric": "Stage-1 train samples/sec (VGGT+Qwen3-4B bf16)", "value": round(value, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_executed": nwarm, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8-e4m3 fwd / bf16 bwd" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": ("Stage-1 ScanQA bf16: VGGT-1B aggregator @%dpx x %d view(s) + 128-latent/6-layer "
                                    "Perceiver + Qwen3-4B fwd+bwd + RCCL all-reduce + AdamW(fp32 master); random-init "
                                    "weights; a step = one micro-batch of %d samples, run %d micro-batches per forward/backward pass "
                                    "with each micro-batch's loss normalised by its own labelled rows" % (args.image_size, V, B, trainer.pass_size(accum))),
                       "global_batch": world * B, "batch_per_gpu": B, "seq_len": L, "views": V,
                       "grad_accum": accum, "optimizer_steps_timed": len(cycles), "micro_batches_per_pass": int(trainer.pass_size(accum)),
                       "parallelism": f"dp{world}", "dp_mode": trainer.dp_mode, "geom_tokens": 8 if args.geom else 0,
                       "trim_padding": bool(args.trim_pad), "fp8_text_forward": bool(args.fp8),
                       "train_projector": bool(args.train_projector), "optimizer_overlap": bool(trainer.overlap_optimizer), "vision_prefetch": bool(args.vision_prefetch), "qwen_layers": args.layers, "valid": args.layers == 36},
            "loss": round(float(loss.item()), 4),
            # utilisation counts EXECUTED FLOPs: the lm_head + CE run on the labelled rows only (same loss and gradients);
            # the reference's figure (logits for all L positions) is kept beside it, never used for a utilisation number
            "executed_tflop_per_sample": round(tf_exec_train, 3), "peak_hbm_gb_torch_allocator": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "reference_algorithmic_tflop_per_sample": round(tf_train, 3),
            "model_flops_utilisation": round(value * tf_exec_train / (world * BF16_DENSE_PEAK_TFLOPS), 4),
            "build_s": round(t_build, 1),
            "roofline": roof,
            "forward_only": forward_only,
            "trimmed_padding_variant": trimmed,
            "comm": comm,
        }
        out.update(sched_variants)
        out.update(variants)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model, batch, L)
            except Exception as e:  # the baseline is reported-only; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
