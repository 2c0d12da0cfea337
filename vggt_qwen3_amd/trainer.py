"""Data-parallel Stage-1 step around VGGTQwen3VLM: the MI355X-native replacement for the reference's
Accelerate/DeepSpeed loop body (src/train/train_sft.py:138-163 optimiser + schedule, :208-220 step).

One process per GPU. Every rank holds the full model (288 GB HBM3E: bf16 weights + bf16 grads +
fp32 master/Adam state of the 4 B trainable parameters is ~64 GB, so nothing is sharded - no ZeRO). The only exchange
is a SUM all-reduce of the flat bf16 gradient buffer over RCCL/xGMI, issued bucket by bucket from inside the
hand-written backward (layers 35 -> 0) on a side stream so it overlaps the remaining backward; the 1/world factor is
folded into the fused AdamW kernel. Semantics kept from the reference: two LR groups selected by parameter NAME
("projector"/"geom_head" -> proj_lr, else lr), AdamW(weight_decay) on every parameter, cosine schedule with warm-up
stepped once per optimiser step, loss/grad_accum scaling, and `max_steps` counting micro-batches.

Two things the reference's default launch has and a plain AdamW loop does not (both on by default here):
  * global-norm gradient clipping at 1.0 (configs/deepspeed_zero3.json:15 "gradient_clipping": 1.0): the squared norm of the
    reduced gradients is summed on the device (vq3_sumsq) and the clip coefficient is applied inside the fused AdamW kernel;
  * Accelerate's scheduler rule (accelerate/scheduler.py:73-82, reached from train_sft.py:219): without `split_batches` the
    LR schedule advances `num_processes` times per optimiser step, so at N GPUs the cosine runs N times faster."""
from __future__ import annotations

import math
import os
import weakref
from typing import List, Optional

import torch
import torch.distributed as dist

from . import dp, ops
from .ops import BF16, F32
from .vlm import VGGTQwen3VLM


def cosine_with_warmup(step: int, warmup: int, total: int) -> float:
    """transformers.get_cosine_schedule_with_warmup multiplier (num_cycles = 0.5)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def schedule_multiplier(opt_step: int, sched_ticks: int, warmup: int, total: int) -> float:
    """LR multiplier applied BY optimiser step `opt_step` (1-based). Before it the schedule has been ticked
    (opt_step - 1) * sched_ticks times: sched_ticks = 1 is a plain `optimizer.step(); scheduler.step()` loop, sched_ticks =
    num_processes is what the reference gets from Accelerate (accelerate/scheduler.py:73-82 steps the wrapped LambdaLR once
    per process after every real optimiser step; skipped micro-batches only bump `_step_count`, which LambdaLR ignores)."""
    return cosine_with_warmup((opt_step - 1) * sched_ticks, warmup, total)


class Stage1Trainer:
    def __init__(self, model: VGGTQwen3VLM, *, lr=5e-6, proj_lr=1e-4, weight_decay=0.1, warmup_ratio=0.03,
                 max_steps=30000, grad_accum=32, betas=(0.9, 0.999), eps=1e-8, bucket_layers: int = 4,
                 process_group=None, max_grad_norm: Optional[float] = 1.0, accelerate_scheduler_rule: bool = True,
                 wgrad_defer: Optional[int] = None, vision_group: Optional[int] = None, text_group: Optional[int] = None,
                 dp_mode: Optional[str] = None):
        self.model = model
        self.tm = model.text_model
        self.lr, self.proj_lr, self.wd = lr, proj_lr, weight_decay
        self.betas, self.eps = betas, eps
        self.grad_accum = grad_accum
        self.max_steps = max_steps
        self.warmup = int(warmup_ratio * max_steps)
        self.pg = process_group
        self.dist_on = dist.is_available() and dist.is_initialized()
        ops._TUNE_GROUP = process_group       # "multi-rank" is decided on the trainer's own group
        ops.gemm_tune_setup(force=True)      # multi-rank: no kernel-choice measurements (table, then heuristic: the same on every rank)
        self.world = dist.get_world_size(process_group) if self.dist_on else 1
        self.rank = dist.get_rank(process_group) if self.dist_on else 0
        # "allreduce" (default): SUM all-reduce of every bucket, replicated AdamW. "sharded" (opt-in, dp.py): reduce-scatter of every
        # bucket from inside the backward, clipping + AdamW on this rank's 1/world of each bucket, all-gather of the updated weights
        self.dp_mode = dp_mode or os.environ.get("VQ3_DP_MODE", "allreduce")
        if self.dp_mode not in ("allreduce", "sharded"):
            raise ValueError(f"dp_mode must be 'allreduce' or 'sharded', got {self.dp_mode!r}")
        # (one rank: nothing to shard - unless VQ3_FORCE_DIST asks for the collectives themselves, the one-rank RCCL rehearsal of
        # reduce_scatter_tensor / all_gather_into_tensor in their in-place forms: tests/test_dp_gpu.py)
        if not self.dist_on or (self.world == 1 and not os.environ.get("VQ3_FORCE_DIST")):
            self.dp_mode = "allreduce"
        self.max_grad_norm = max_grad_norm
        # scheduler ticks per optimiser step: `world` under Accelerate's rule (the reference), 1 for a plain loop
        self.sched_ticks = self.world if accelerate_scheduler_rule else 1
        # W^T copies for three dgrad GEMMs cost one 2.7 ms refresh per optimiser step: on when that is amortised
        if grad_accum >= 4 and os.environ.get("VQ3_DGRAD_NT", "1") != "0":
            self.tm.enable_dgrad_transposes(True)
        # weight-gradient GEMMs once per `wgrad_defer` micro-batches of a window over their concatenated token rows (qwen3.py,
        # "deferred weight gradients"): +32 % on those GEMMs at depth 4, +35 % at 8, for 4.2 GB of operand slabs per micro-batch held
        # `text_group` micro-batches of a window run as ONE forward/backward over their concatenated samples when micro_step() is given
        # the upcoming batches (fit() looks ahead): every GEMM of the text model sees 8 x 1200 token rows (256 x 256 tiles instead of
        # 200 under-filled 128 x 128 ones), the tower 8 x 6174; each micro-batch's loss stays the mean over ITS labelled rows and the
        # gradient is that of their sum - what the micro-batches produce one by one (Qwen3ForCausalLM.loss_head(groups=...)).
        if text_group is None:
            # measured (round 2, window of 32): 1: 96.6, 2: 106.5, 4: 115.1, 8: 119.1, 16: 115.3 samples/s. Round 3: a window that is a multiple of
            # 10 runs in passes of 10 - [12000, 2560] outputs are 470 tiles of 256 x 256 = 1.84 rounds of 256 CUs (92 % full) where 9600 rows
            # give 380 tiles = 1.48 rounds (74 %), and a window of 20 is two whole passes instead of 8 + 8 + 4 (same box, alternating runs:
            # 137.0 / 137.7 -> 148.5 / 147.6 samples/s at grad_accum 20; 124 GB peak instead of 114)
            # text_group is the CAP; pass_size() cuts a window into equal passes under it (32 -> 4 x 8, 20 -> 2 x 10, 18 -> 2 x 9)
            text_group = int(os.environ.get("VQ3_TEXT_GROUP", "10"))
        self.text_group = max(1, text_group)
        self._merged_pending: List = []     # [(batch dict, loss)] of a merged pass, handed out by the following micro_step() calls
        self._opt_due = False               # the merged pass held the window's boundary: AdamW runs when its last loss is handed out
        self._merge_cache = None
        if wgrad_defer is None:
            wgrad_defer = int(os.environ.get("VQ3_WGRAD_DEFER", "8"))
        self._wgrad_defer = wgrad_defer
        # (depth counts backward passes: with merged passes the same number of token rows per weight-gradient product)
        self.tm.enable_wgrad_deferral(min(max(1, wgrad_defer // self.text_group), max(1, grad_accum)))
        # the frozen vision tower runs once per `vision_group` micro-batches on their concatenated images (vlm.py: precompute_vision)
        # when micro_step() is given the upcoming batches; fit() looks ahead by itself
        if vision_group is None:
            vision_group = int(os.environ.get("VQ3_VISION_GROUP", "4"))      # measured: 1: 91.9, 2: 94.8, 4: 96.0, 8 / 16: 95.8 samples/s
        self.vision_group = max(1, vision_group)
        self.micro = 0       # micro-batches seen (the reference's `step`)
        self.opt_step = 0    # optimiser / scheduler steps
        dev = self.tm.flat_w.device
        n = self.tm.flat_w.numel()
        # fp32 master weights + Adam moments for the text model (flat) and geom_head (small, flat)
        self.master = ops.cast(self.tm.flat_w, F32)
        self.m = torch.zeros(n, device=dev, dtype=F32)
        self.v = torch.zeros(n, device=dev, dtype=F32)
        self.geom_params = [p for _, p in model.geom_head.named_parameters()]
        gn = sum(p.numel() for p in self.geom_params)
        self.geom_master = torch.cat([p.detach().reshape(-1).float() for p in self.geom_params]).contiguous()
        self.geom_m = torch.zeros(gn, device=dev, dtype=F32)
        self.geom_v = torch.zeros(gn, device=dev, dtype=F32)
        # geom_grad carries one extra slot: the number of micro-batches of this window that produced a geom_head gradient
        # (summed over ranks by the same all-reduce, so every rank takes the same "step geom_head or not" decision)
        self.geom_on = model.geom_tokens > 0
        self.geom_grad = torch.zeros(gn + 1, device=dev, dtype=F32)
        self.geom_w16 = torch.zeros(gn, device=dev, dtype=BF16)
        self._gn = gn
        self.norm_sq = torch.zeros(1, device=dev, dtype=F32)
        self.norm_part = torch.zeros(1024, device=dev, dtype=F32)
        self.last_grad_norm = None      # device scalar of the last optimiser step (pre-clip), for logging / tests
        # all-reduce buckets over flat_g: groups of `bucket_layers` layers (contiguous), norm with the last group,
        # the tied embedding last (its gradient is completed by the embedding backward at the very end)
        self.buckets, self.embed_span = dp.plan_buckets(self.tm.table, self.tm.config.num_hidden_layers, bucket_layers)
        dp.check_cover(self.buckets, self.embed_span, n)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.dist_on else None
        # "corrected" mode (VisionLanguageConfig.train_projector): the Perceiver's fp32 parameters, gradients and Adam moments as flat
        # buffers (the parameters and their .grad become views), one gradient all-reduce bucket of their own, the proj_lr group
        self.proj_on = bool(getattr(model, "train_projector", False))
        if self.proj_on:
            pp = [p_ for _, p_ in model.projector.named_parameters()]
            pn = sum(p_.numel() for p_ in pp)
            self.proj_w32 = torch.empty(pn, device=dev, dtype=F32)
            self.proj_g32 = torch.zeros(pn, device=dev, dtype=F32)
            self.proj_m = torch.zeros(pn, device=dev, dtype=F32)
            self.proj_v = torch.zeros(pn, device=dev, dtype=F32)
            self.proj_w16 = torch.empty(pn, device=dev, dtype=BF16)      # (AdamW writes the bf16 image; the compute copies are refreshed from fp32)
            off = 0
            with torch.no_grad():
                for p_ in pp:
                    k = p_.numel()
                    self.proj_w32[off:off + k].copy_(p_.detach().reshape(-1))
                    p_.data = self.proj_w32[off:off + k].view_as(p_)
                    p_.grad = self.proj_g32[off:off + k].view_as(p_)
                    off += k
            model.projector._cc = None
        # dp_mode "sharded": fp32 master / moments are current in this rank's shards only until gather_sharded_state() ran after the
        # last optimiser step (checkpoint.save_trainer_state refuses a state that is not gathered)
        self._shards_gathered = True
        # the optimiser step beside the next pass's vision tower (see _optimizer_step): opt-in, on inside fit() and bench.py
        self.overlap_optimizer = False
        self._opt_stream = None
        self._works: List = []
        self.comm_profile: Optional[List] = None    # set to [] to collect (start event, end event, bytes) per gradient all-reduce
        self._fired: List[int] = []     # buckets all-reduced from inside the backward of the current boundary micro-batch
        if not hasattr(model, "_trainers"):
            model._trainers = []
        model._trainers.append(weakref.ref(self))

    def pass_size(self, remaining: int) -> int:
        """Micro-batches of the next merged pass when `remaining` are left in the accumulation window: the window's rest is cut into the
        fewest passes of at most `text_group` micro-batches, of equal size up to one (32 -> 8 + 8 + 8 + 8 and 20 -> 10 + 10 at the default
        cap of 10, 18 -> 9 + 9 rather than 8 + 8 + 2: a short last pass is the one-micro-batch regime again - every [rows, 2560] output of
        a 2-micro-batch pass is 100 tiles for 256 CUs)."""
        cap = max(1, self.text_group)
        passes = (remaining + cap - 1) // cap
        return max(1, (remaining + passes - 1) // passes)

    def set_schedule(self, *, text_group: Optional[int] = None, grad_accum: Optional[int] = None) -> None:
        """Change micro-batches per pass and / or per optimiser step between two accumulation windows (bench.py times the same job
        under several schedules); the deferred weight-gradient depth follows as in __init__."""
        if self._merged_pending or self._opt_due or any(self.tm._wd_rows) or self.micro % self.grad_accum != 0:
            raise RuntimeError("set_schedule: only on an accumulation boundary (flush_pending() first)")
        self.sync_optimizer()
        if grad_accum is not None and max(1, int(grad_accum)) != self.grad_accum:
            self.grad_accum = max(1, int(grad_accum))
            # the micro-batch counter (max_steps, step_N checkpoint names, the saved `micro`) is never rewound: it moves UP to the next
            # multiple of the new window so that window boundaries and the counter stay in phase (bench.py resets it itself)
            self.micro = (self.micro + self.grad_accum - 1) // self.grad_accum * self.grad_accum
        if text_group is not None:
            self.text_group = max(1, int(text_group))
        self.tm.enable_wgrad_deferral(min(max(1, self._wgrad_defer // self.text_group), max(1, self.grad_accum)))

    def resync_master_from_weights(self) -> None:
        """The bf16 weights were replaced behind the trainer's back (a checkpoint load): re-derive the fp32 master copy, or
        the next AdamW step would write the old weights back. Adam moments are kept."""
        self.sync_optimizer()
        self.master.copy_(ops.cast(self.tm.flat_w, F32))
        off = 0
        for p in self.geom_params:
            self.geom_master[off:off + p.numel()].copy_(p.detach().reshape(-1).float())
            off += p.numel()

    # ------------------------------------------------------------------ communication
    def _allreduce_span(self, lo: int, hi: int):
        if not self.dist_on:
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            prof = self.comm_profile
            if prof is not None:        # HIP events on the stream the collective runs on (bench.py: allreduce_ms_per_opt_step)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.dp_mode == "sharded":
                dp.reduce_scatter_span(self.tm.flat_g, lo, hi, self.rank, self.world, group=self.pg)
            else:
                dp.allreduce_span(self.tm.flat_g, lo, hi, group=self.pg)
            if prof is not None:
                e1.record()
                prof.append((e0, e1, (hi - lo) * self.tm.flat_g.element_size()))

    def _layer_done(self, i: int):
        if i in self.buckets:
            self._fired.append(i)
            self._allreduce_span(*self.buckets[i])

    # ------------------------------------------------------------------ one micro-batch
    def micro_step(self, batch: dict, next_batch: Optional[dict] = None, upcoming=None) -> torch.Tensor:
        """forward + backward (+ gradient all-reduce, AdamW, schedule on accumulation boundaries). Returns the loss.
        next_batch (optional): its frozen vision-tower forward is enqueued on a second stream now, overlapping this
        micro-batch's text forward/backward; pass the same dict to the next call.
        upcoming (optional): the batches of the following micro-steps, in order. With text_group > 1 this call then runs ONE
        forward/backward over this batch + the next text_group - 1 of them that still belong to the current accumulation window
        (every micro-batch's loss is the mean over its own labelled rows; the gradient is that of their sum) and the following
        micro_step() calls - which must receive those very dicts, in order - only return their loss. With text_group == 1 the
        frozen vision tower alone is shared: it runs once over this batch + the first vision_group - 1 upcoming ones."""
        model, tm = self.model, self.tm
        if self._merged_pending:            # this micro-batch already ran as part of a merged pass: hand its loss out
            expected, loss = self._merged_pending[0]
            if expected is not batch:
                raise RuntimeError("micro_step: the batches passed as `upcoming` must come back, in order, as the same objects "
                                   "(flush_pending() accounts for the rest of the merged pass if the loop was interrupted)")
            self._merged_pending.pop(0)
            if not self._merged_pending and self._opt_due:
                # the window's optimiser step runs with the call that returns the window's LAST loss: `micro`, `opt_step`, lrs() and
                # the weights stay in phase for every call in between (a step_N save inside the group sees pre-step weights)
                self._opt_due = False
                self._optimizer_step()
            self.micro += 1
            return loss
        k = self.micro % self.grad_accum
        members = [batch]
        if self.text_group > 1 and upcoming:
            cand = [batch] + list(upcoming)[: self.pass_size(self.grad_accum - k) - 1]           # never across a window boundary
            if len(cand) > 1 and self._mergeable(cand):
                members = cand
        gsize = len(members)
        sizes = None
        if gsize > 1:
            batch_run, sizes = self._merge(members)
        else:
            batch_run = batch
            if self.vision_group > 1 and upcoming and not any(im is batch["pixel_values"] for im, _ in model._vis_group):
                group = [batch["pixel_values"]] + [b["pixel_values"] for b in list(upcoming)[: self.vision_group - 1]]
                model.precompute_vision(group)
        if next_batch is not None:
            model.prefetch_images(next_batch["pixel_values"])
        boundary = (k + gsize - 1 == self.grad_accum - 1)
        accumulate = k != 0
        st = model.forward_state(batch_run["pixel_values"], batch_run.get("geom_token"), batch_run["input_ids"],
                                 batch_run["attention_mask"], batch_run["labels"], need_grad=True, loss_groups=sizes)
        hook = self._layer_done if (boundary and self.dist_on) else None
        self._fired = []
        if not accumulate:
            self.geom_grad.zero_()
            if self.proj_on:
                self.proj_g32.zero_()
        # A micro-batch without any labelled token (the answer truncated away) has no gradient: _backward_text then zeroes
        # flat_g on the first micro-batch of a window and still fires every layer_done hook, so the ranks' collectives match.
        d_geom = model._backward_text(st, 1.0 / self.grad_accum, accumulate, layer_done=hook, flush=boundary)
        if d_geom is not None:
            g = model.geom_head_backward(st, d_geom)
            self.geom_grad[: self._gn] += torch.cat([g["0.weight"].reshape(-1), g["0.bias"].reshape(-1),
                                                     g["2.weight"].reshape(-1), g["2.bias"].reshape(-1)])
            self.geom_grad[self._gn:] += float(gsize)
        if gsize > 1:
            self.micro += 1
            losses = st["loss"]
            self._merged_pending = [(members[j], losses[j]) for j in range(1, gsize)]
            self._opt_due = boundary        # deferred to the micro_step() call that hands out the group's last loss
            return losses[0]
        if boundary:
            self._optimizer_step()
        self.micro += 1
        return st["loss"]

    def flush_pending(self) -> List[torch.Tensor]:
        """Account for the micro-batches of a merged pass whose micro_step() calls never came (the loop was interrupted, or is
        about to feed different batches): their gradient contribution is already in flat_g, so `micro` advances over them, the
        window's deferred optimiser step runs if it was due, and their losses are returned. A no-op outside a merged group."""
        losses = [loss for _, loss in self._merged_pending]
        self.micro += len(self._merged_pending)
        self._merged_pending = []
        if self._opt_due:
            self._opt_due = False
            self._optimizer_step()
        return losses

    @staticmethod
    def _mergeable(members) -> bool:
        b0 = members[0]
        for b in members[1:]:
            for key in ("pixel_values", "input_ids", "attention_mask", "labels"):
                if tuple(b[key].shape[1:]) != tuple(b0[key].shape[1:]) or b[key].dtype != b0[key].dtype or b[key].device != b0[key].device:
                    return False
            if (b.get("geom_token") is None) != (b0.get("geom_token") is None):
                return False
        return True

    def _merge(self, members):
        """Concatenate the collator dicts of several micro-batches along the sample axis (remembered for the last group of batch
        OBJECTS, so that a loop that re-feeds the same batches re-uses the tensors and the host-side facts memoised on them)."""
        ids = tuple(id(b) for b in members)
        mc = self._merge_cache
        if mc is not None and mc[0] == ids and all(a is b for a, b in zip(mc[1], members)):
            return mc[2], mc[3]
        merged = {k: torch.cat([b[k] for b in members], dim=0) for k in ("pixel_values", "input_ids", "attention_mask", "labels")}
        if members[0].get("geom_token") is not None:
            merged["geom_token"] = {k: torch.cat([b["geom_token"][k] for b in members], dim=0) for k in members[0]["geom_token"]}
        sizes = [int(b["input_ids"].shape[0]) for b in members]
        self._merge_cache = (ids, list(members), merged, sizes)
        return merged, sizes

    def lr_mult(self, opt_step: int) -> float:
        """Schedule multiplier used BY optimiser step `opt_step` (1-based): the schedule has been ticked
        (opt_step - 1) * sched_ticks times before it."""
        return schedule_multiplier(opt_step, self.sched_ticks, self.warmup, self.max_steps)

    def _optimizer_step(self):
        """The window's optimiser step. With `overlap_optimizer` on (fit() and bench.py turn it on; VQ3_OPT_OVERLAP=0 keeps it off) the
        whole step - outstanding all-reduces, the norm, AdamW over 112 GB of state, the W^T / e4m3 refresh: HBM-bound, ~23 ms at Qwen3-4B -
        is enqueued on a stream of its own behind everything the backward enqueued, and the caller's stream goes straight on: the NEXT
        pass starts with the frozen vision tower (compute-bound, no trainable weight), which runs beside it. The first reader of an updated
        tensor waits on the event left in model._weights_gate (vlm.forward_state: before the text model, geom_head and - when it trains -
        the projector); anything else that looks at weights or optimiser state calls sync_optimizer() first."""
        if not self.overlap_optimizer:
            return self._optimizer_step_impl()
        main = torch.cuda.current_stream()
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.tm.flat_w.device)
        self.sync_optimizer()                                   # (a previous step nobody consumed yet)
        self._opt_stream.wait_stream(main)
        with torch.cuda.stream(self._opt_stream):
            self._optimizer_step_impl()
            ev = torch.cuda.Event()
            ev.record(self._opt_stream)
        self.model._weights_gate = ev
        self.tm._weights_gate = ev           # (readers that only know the text model: generate(), enable_fp8_forward / requantize_fp8)

    def sync_optimizer(self) -> None:
        """Make the current stream wait for an optimiser step that is still running on the side stream (no-op otherwise)."""
        ev = getattr(self.model, "_weights_gate", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.model._weights_gate = None
            self.tm._weights_gate = None

    def grad_norm(self) -> Optional[torch.Tensor]:
        """Pre-clip global gradient norm of the last optimiser step (device scalar), ordered after a step that may still be running on the
        side stream (`last_grad_norm` read directly from the main stream is not)."""
        self.sync_optimizer()
        return self.last_grad_norm

    def check_kernels(self) -> None:
        """Raises if a split-K GEMM reducer gave up its bounded wait in any launch completed so far (its output tile - an activation or a
        gradient - is then incomplete: training on would be silently wrong). A host read of one word in host-mapped memory, no
        synchronisation: called once per optimiser step and at the end of fit()."""
        if ops.gemm_split_poll(clear=True):
            raise RuntimeError("vq3: a split-K GEMM launch gave up waiting for its partial tiles (broken launch); activations / gradients "
                               f"of the window ending at micro-batch {self.micro} are incomplete - the optimiser step was NOT applied")

    def _optimizer_step_impl(self):
        tm = self.tm
        self.check_kernels()                 # before the weights are touched
        if self.dist_on:
            for g0 in sorted(self.buckets, reverse=True):     # buckets the backward did not reach (no hook fired)
                if g0 not in self._fired:
                    self._allreduce_span(*self.buckets[g0])
            self._allreduce_span(*self.embed_span)
            if self.geom_on:                                   # unconditional: identical collective sequence on every rank
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    dp.allreduce_tensor(self.geom_grad, group=self.pg)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        proj_g16 = None
        if self.proj_on:
            # the projector's gradients leave as bf16 like the text model's: one bucket of their own (replicated in either dp_mode)
            proj_g16 = ops.cast(self.proj_g32, BF16)
            if self.dist_on:
                dp.allreduce_tensor(proj_g16, group=self.pg)
        self.opt_step += 1
        mult = self.lr_mult(self.opt_step)
        gscale = 1.0 / self.world
        geom_step = self.geom_on and float(self.geom_grad[self._gn].item()) > 0   # one host read per optimiser step
        clip = None
        sharded = self.dp_mode == "sharded"
        # the parts of flat_g / flat_w this rank owns: everything (replicated), or its shard of every bucket + the replicated tails
        if sharded:
            spans, tails = [], []
            for lo, hi in list(self.buckets.values()) + [self.embed_span]:
                sh, tail = dp.shard_layout(lo, hi, self.world)
                if sh:
                    spans.append((lo + self.rank * sh, lo + (self.rank + 1) * sh))
                if tail < hi:
                    tails.append((tail, hi))
        else:
            spans, tails = [(0, tm.flat_g.numel())], []
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            # global L2 norm over every trainable gradient (text model + geom_head), as DeepSpeed / clip_grad_norm_ take it
            self.norm_sq.zero_()
            for a, b in spans:
                ops.sumsq(tm.flat_g[a:b], self.norm_part, self.norm_sq)
            if not sharded or self.rank == 0:                       # replicated pieces are counted once when the sum is all-reduced
                for a, b in tails:
                    ops.sumsq(tm.flat_g[a:b], self.norm_part, self.norm_sq)
                if geom_step:
                    ops.sumsq(self.geom_grad[: self._gn], self.norm_part, self.norm_sq)
                if proj_g16 is not None:
                    ops.sumsq(proj_g16, self.norm_part, self.norm_sq)
            if sharded:
                dp.allreduce_tensor(self.norm_sq, group=self.pg)    # every rank gets the same total
            clip = (self.norm_sq, float(self.max_grad_norm))
            self.last_grad_norm = self.norm_sq.sqrt() * gscale
        for a, b in spans + tails:
            ops.adamw_step(self.master[a:b], self.m[a:b], self.v[a:b], tm.flat_g[a:b], tm.flat_w[a:b], self.lr * mult, self.betas[0],
                           self.betas[1], self.eps, self.wd, self.opt_step, gscale, clip=clip)
        if sharded:                                                 # the updated bf16 weights of every shard reach every rank
            for lo, hi in list(self.buckets.values()) + [self.embed_span]:
                dp.all_gather_span(tm.flat_w, lo, hi, self.rank, self.world, group=self.pg)
            self._shards_gathered = self.world == 1
        if proj_g16 is not None:
            ops.adamw_step(self.proj_w32, self.proj_m, self.proj_v, proj_g16, self.proj_w16, self.proj_lr * mult, self.betas[0],
                           self.betas[1], self.eps, self.wd, self.opt_step, gscale, clip=clip)
            self.model.projector._cc = None               # bf16 compute copies are rebuilt from the updated fp32 parameters
        tm.refresh_derived()     # e4m3 / W^T copies of the weights (no-ops unless enabled)
        if geom_step:
            ops.adamw_step(self.geom_master, self.geom_m, self.geom_v, ops.cast(self.geom_grad[: self._gn].contiguous(), BF16),
                           self.geom_w16, self.proj_lr * mult, self.betas[0], self.betas[1], self.eps, self.wd,
                           self.opt_step, gscale, clip=clip)
            off = 0
            with torch.no_grad():
                for p in self.geom_params:
                    p.copy_(self.geom_master[off:off + p.numel()].view_as(p))
                    off += p.numel()

    # ------------------------------------------------------------------ the reference's loop
    def fit(self, batches, *, log_every_steps: int = 10, save_every_steps: Optional[int] = None, output_dir=None,
            samples_per_batch: Optional[int] = None, log=print) -> List[dict]:
        """The body of train_sft.py:208-255 around micro_step(): `batches` is any iterable of collator dicts (restarted when
        exhausted); the loop stops after `max_steps` MICRO-batches (the reference's `step` counts micro-batches, :244-253);
        rank 0 logs loss, both learning rates, steps/s and samples/s every `log_every_steps`; every `save_every_steps` and at
        the end the model is written in the reference's key space (+ trainer state, when the step is an accumulation
        boundary). Returns the log records."""
        import time
        from pathlib import Path
        from . import checkpoint
        rank = dist.get_rank(self.pg) if self.dist_on else 0
        records: List[dict] = []
        overlap0 = self.overlap_optimizer
        self.overlap_optimizer = os.environ.get("VQ3_OPT_OVERLAP", "1") != "0"      # the optimiser step beside the next pass's tower
        it = iter(batches)
        t0 = time.perf_counter()
        start = self.micro
        from collections import deque
        ahead: deque = deque()          # look-ahead of vision_group batches: their images share one pass of the frozen vision tower

        def pull():
            nonlocal it
            try:
                return next(it)
            except StopIteration:
                it = iter(batches)
                return next(it)

        while self.micro < self.max_steps:
            while len(ahead) < max(self.vision_group, self.text_group) and self.micro + len(ahead) < self.max_steps:
                ahead.append(pull())
            batch = ahead.popleft()
            loss = self.micro_step(batch, upcoming=list(ahead))
            step = self.micro - 1
            if rank == 0 and step % log_every_steps == 0:
                dt = time.perf_counter() - t0
                done = self.micro - start
                nb = samples_per_batch if samples_per_batch is not None else int(batch["input_ids"].shape[0])
                lr, plr = self.lrs()
                rec = {"step": step, "loss": float(loss.item()), "lr": lr, "proj_lr": plr, "steps_per_s": done / dt,
                       "samples_per_s": done * nb * self.world / dt}
                records.append(rec)
                log(f"Step {step:5d}/{self.max_steps} | Loss: {rec['loss']:.4f} | LR: {lr:.2e}/{plr:.2e} | "
                    f"Speed: {rec['steps_per_s']:.2f} steps/s, {rec['samples_per_s']:.1f} samples/s")
            if save_every_steps and output_dir is not None and self.micro % save_every_steps == 0:
                self._save(Path(output_dir) / f"step_{self.micro}", rank, checkpoint)
        self.sync_optimizer()
        torch.cuda.current_stream().synchronize()
        self.check_kernels()                 # the last window's launches have completed
        self.overlap_optimizer = overlap0
        if output_dir is not None:
            self._save(Path(output_dir), rank, checkpoint)
        return records

    def gather_sharded_state(self) -> None:
        """dp_mode "sharded": fp32 master weights and Adam moments are current only inside this rank's shards; all-gather them so
        that rank 0 can write a complete trainer state (a no-op in the replicated mode)."""
        self.sync_optimizer()
        if self.dp_mode != "sharded":
            return
        torch.cuda.current_stream().synchronize()
        for lo, hi in list(self.buckets.values()) + [self.embed_span]:
            for t in (self.master, self.m, self.v):
                dp.all_gather_span(t, lo, hi, self.rank, self.world, group=self.pg)
        self._shards_gathered = True

    def _save(self, path, rank: int, checkpoint) -> None:
        self.sync_optimizer()
        if self.dist_on:
            dist.barrier(group=self.pg)
            if self.micro % self.grad_accum == 0:
                self.gather_sharded_state()
        if rank == 0:                       # replicas are identical: one writer (the reference makes every rank call save_state
            checkpoint.save_model(self.model, path)        # only because ZeRO-3 shards need gathering)
            if self.micro % self.grad_accum == 0:
                checkpoint.save_trainer_state(self, path)
        if self.dist_on:
            dist.barrier(group=self.pg)

    def lrs(self):
        mult = self.lr_mult(max(1, self.opt_step))
        return self.lr * mult, self.proj_lr * mult
