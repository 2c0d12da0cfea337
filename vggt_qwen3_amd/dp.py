"""Data-parallel gradient exchange: bucket planning over the flat gradient buffer and the all-reduce itself.
Pure torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests); no kernels here.

The path shards by samples only: every rank runs whole micro-batches, the one exchange is a SUM
all-reduce of the gradients before the optimiser step, bucketed so that it can be issued from inside the backward
(last layers first) and overlap the rest of it. The 1/world_size factor is applied by the AdamW kernel."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.distributed as dist


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def plan_buckets(table: Dict[str, tuple], num_layers: int, bucket_layers: int) -> Tuple[Dict[int, Tuple[int, int]], Tuple[int, int]]:
    """table: name -> (offset, shape) of the flat buffer laid out [embed | l0.* ... l{L-1}.* | norm].
    Returns ({first_layer_of_group: (lo, hi)}, embed_span). Groups are contiguous; the final norm rides with the last
    group; the tied embedding is its own bucket (its gradient is only complete at the very end of the backward)."""
    offs = sorted(o for o, _ in table.values())
    o_last, s_last = max(table.values(), key=lambda v: v[0])
    total = round_up(o_last + math.prod(s_last), 64)

    def span(first: str, last: str) -> Tuple[int, int]:
        """[offset of `first`, offset of the entry that follows `last`) - entries may carry alignment / zero-row padding."""
        lo = table[first][0]
        nxt = [o for o in offs if o > table[last][0]]
        return lo, (nxt[0] if nxt else total)
    buckets = {}
    for g0 in range(0, num_layers, bucket_layers):
        g1 = min(num_layers, g0 + bucket_layers) - 1
        buckets[g0] = span(f"l{g0}.qkv", "norm" if g1 == num_layers - 1 else f"l{g1}.kn")
    return buckets, span("embed", "embed")


def check_cover(buckets: Dict[int, Tuple[int, int]], embed_span: Tuple[int, int], total: int) -> None:
    """Every element of the flat buffer belongs to exactly one bucket."""
    spans = sorted([embed_span] + list(buckets.values()))
    pos = 0
    for lo, hi in spans:
        if lo != pos:
            raise AssertionError(f"gap or overlap at {pos} -> {lo}")
        pos = hi
    if pos != total:
        raise AssertionError(f"buckets end at {pos}, buffer has {total}")


def allreduce_tensor(t: torch.Tensor, group=None):
    """SUM all-reduce in place. RCCL ("nccl") reduces device memory directly over xGMI; the gloo backend - used only to
    rehearse several ranks on ONE GPU or on the CPU (tests/test_dp_gpu.py, tests/test_host_logic.py) - is given a host
    copy of a device tensor, since its device-tensor support depends on how torch was built."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().to("cpu", copy=True)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def allreduce_span(flat: torch.Tensor, lo: int, hi: int, group=None, async_op: bool = False):
    return allreduce_tensor(flat[lo:hi], group=group)


# ---- opt-in: reduce-scatter + all-gather (Stage1Trainer(dp_mode="sharded")) ---------------------------------------------------
# The same bytes as the all-reduce cross xGMI (a ring all-reduce IS a reduce-scatter followed by an all-gather), but the two halves
# are separate collectives with the optimiser between them: every rank clips and runs AdamW on 1/world of each bucket only (the
# 19 ms HBM-bound AdamW pass of the replicated design becomes 19 / world ms) and all-gathers the updated bf16 WEIGHTS instead of the
# summed gradients. What matters at grad_accum 1, where the exchange is paid every micro-batch.
SHARD_ALIGN = 64        # elements: shard boundaries stay 128-byte aligned (the fused AdamW kernel's vector width and more)


def shard_layout(lo: int, hi: int, world: int):
    """[lo, hi) = world equal shards of `s` elements (s a multiple of SHARD_ALIGN) + a replicated tail [lo + world * s, hi)."""
    s = ((hi - lo) // world) // SHARD_ALIGN * SHARD_ALIGN
    return s, lo + world * s


def reduce_scatter_span(flat: torch.Tensor, lo: int, hi: int, rank: int, world: int, group=None) -> None:
    """In place: afterwards rank r holds the SUM over ranks in its shard [lo + r s, lo + (r + 1) s) and in the tail; the other
    shards of [lo, hi) hold partial garbage (this rank's own contribution) and must not be read."""
    s, tail = shard_layout(lo, hi, world)
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"          # gloo rehearsals: through the host (see allreduce_tensor)
    if s:
        src = flat[lo:tail]
        if staged:
            h = src.detach().to("cpu", copy=True)
            out = torch.empty(s, dtype=h.dtype)
            dist.reduce_scatter_tensor(out, h, op=dist.ReduceOp.SUM, group=group)
            flat[lo + rank * s: lo + (rank + 1) * s].copy_(out)
        else:
            dist.reduce_scatter_tensor(flat[lo + rank * s: lo + (rank + 1) * s], src, op=dist.ReduceOp.SUM, group=group)
    if tail < hi:
        allreduce_tensor(flat[tail:hi], group=group)


def all_gather_span(flat: torch.Tensor, lo: int, hi: int, rank: int, world: int, group=None) -> None:
    """In place: every rank's shard of [lo, hi) is sent to all (the tail is replicated already)."""
    s, tail = shard_layout(lo, hi, world)
    if not s:
        return
    mine = flat[lo + rank * s: lo + (rank + 1) * s]
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(world * s, dtype=flat.dtype)
        dist.all_gather_into_tensor(h, mine.detach().to("cpu", copy=True), group=group)
        flat[lo:tail].copy_(h)
    else:
        dist.all_gather_into_tensor(flat[lo:tail], mine, group=group)


def allreduce_in_backward_order(flat: torch.Tensor, buckets: Dict[int, Tuple[int, int]], embed_span: Tuple[int, int],
                                group=None) -> None:
    """Reference schedule (used by the tests and as the no-overlap fallback): groups from the last layer to the
    first, then the embedding."""
    for g0 in sorted(buckets, reverse=True):
        allreduce_span(flat, *buckets[g0], group=group)
    allreduce_span(flat, *embed_span, group=group)
