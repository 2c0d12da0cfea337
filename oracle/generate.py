"""TEST INFRASTRUCTURE - CPU restatement of greedy `generate` as the reference's inference scripts call it
(src/inference/qa_inference.py:207-216, arkit_inference.py:274-284). The arithmetic lives in `transformers`
(GenerationMixin._sample greedy branch; RepetitionPenaltyLogitsProcessor; NoRepeatNGramLogitsProcessor), unpinned
upper bound in the reference's environment, 5.15.0 in this container. Deliberately naive: no KV cache - every step
re-runs the whole (unpadded) row through oracle.qwen3 - so it shares nothing with the product's cache logic.
Pinned by tests/golden/generate_tiny.npz, produced by transformers' own generate() (tools/make_golden_generate.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import qwen3 as oq


def repetition_penalty_(scores: torch.Tensor, ids: Sequence[int], penalty: float) -> None:
    """logits_process.py RepetitionPenaltyLogitsProcessor: gather, rescale (x*p if x<0 else x/p), scatter."""
    if penalty == 1.0 or len(ids) == 0:
        return
    idx = torch.tensor(sorted(set(ids)), dtype=torch.long)
    s = scores[idx]
    scores[idx] = torch.where(s < 0, s * penalty, s / penalty)


def no_repeat_ngram_(scores: torch.Tensor, ids: Sequence[int], n: int) -> None:
    """logits_process.py NoRepeatNGramLogitsProcessor / _calc_banned_ngram_tokens."""
    cur = len(ids)
    if n <= 0 or cur + 1 < n:
        return
    prefix = tuple(ids[cur + 1 - n:cur])
    for i in range(cur - n + 1):
        if tuple(ids[i:i + n - 1]) == prefix:
            scores[ids[i + n - 1]] = -float("inf")


def greedy_generate(sd: Dict[str, torch.Tensor], cfg: oq.Qwen3Cfg, inputs_embeds: Optional[torch.Tensor],
                    attention_mask: torch.Tensor, max_new_tokens: int, repetition_penalty: float = 1.0,
                    no_repeat_ngram_size: int = 0, eos_token_id=None, pad_token_id: int = 0,
                    input_ids: Optional[torch.Tensor] = None, trace: Optional[List] = None) -> torch.Tensor:
    emb_w = sd["model.embed_tokens.weight"]
    eos = [] if eos_token_id is None else ([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id))
    if input_ids is not None:
        inputs_embeds = F.embedding(input_ids, emb_w)
    B = inputs_embeds.shape[0]
    rows = [inputs_embeds[b][attention_mask[b] != 0] for b in range(B)]      # compact: positions 0..n-1
    seen: List[List[int]] = [input_ids[b].tolist() if input_ids is not None else [] for b in range(B)]
    out: List[List[int]] = [[] for _ in range(B)]
    finished = [False] * B
    for _ in range(max_new_tokens):
        for b in range(B):
            x = rows[b][None]
            h = oq.model_forward(x, torch.ones(1, x.shape[1], dtype=torch.long), sd, cfg)
            scores = F.linear(h[0, -1], emb_w).float()
            repetition_penalty_(scores, seen[b], repetition_penalty)
            no_repeat_ngram_(scores, seen[b], no_repeat_ngram_size)
            tok = int(scores.argmax())
            if trace is not None:
                top = scores.topk(2).values
                trace.append(float(top[0] - top[1]))
            if finished[b]:
                tok = pad_token_id
            out[b].append(tok)
            seen[b].append(tok)
            rows[b] = torch.cat([rows[b], emb_w[tok][None]], dim=0)
            if tok in eos:
                finished[b] = True
        if eos and all(finished):
            break
    new = torch.tensor(out, dtype=torch.long)
    return new if input_ids is None else torch.cat([input_ids, new], dim=1)
