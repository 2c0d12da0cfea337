"""Greedy generation on the HIP kernels with a resident KV cache - the surface the reference's inference scripts use:
`model.text_model.generate(inputs_embeds=, attention_mask=, max_new_tokens=, do_sample=False, num_beams=1,
repetition_penalty=, no_repeat_ngram_size=, eos_token_id=, pad_token_id=)` (src/inference/qa_inference.py:207-216,
arkit_inference.py:274-284) plus the vision-token INSERTION splice those scripts perform before calling it
(qa_inference.py:119-145) - different from training's overwrite splice.

Design (MI355X): prompts are prefilled without their padding (left-padded rows are compacted, so no pad token is ever
computed and positions are 0..n-1 exactly as transformers derives them from the mask); K/V live in one
[layers, B, Hkv, Lmax, 128] bf16 buffer per operand; a decode step streams every weight once through the skinny
GEMM; all step state is in device memory, so the whole step is captured once in a HIP graph and replayed per token -
the host only checks the `finished` flags every few steps. One row at Qwen3-4B's shape (the reference's case): the 36
layers of a step are ONE persistent launch (csrc/decode_layers.hip) + lm_head + pick; otherwise ~220 launches.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import ops
from .ops import BF16, F32, round_up


def insert_vision_tokens(input_ids: torch.Tensor, attention_mask: torch.Tensor, inputs_embeds: torch.Tensor,
                         vis_tokens: torch.Tensor, image_token_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """qa_inference.py:119-145 / arkit_inference.py:182-214: the FIRST <image> position found (row-major) is replaced,
    in every row, by the visual span; the sequence grows by vis_len - 1 and the span is attended. No <image> -> inputs
    returned unchanged."""
    positions = (input_ids == image_token_id).nonzero(as_tuple=False)
    if positions.numel() == 0:
        return inputs_embeds, attention_mask
    pos = int(positions[0, 1])
    vis_len = vis_tokens.shape[1]
    new_inputs = torch.cat([inputs_embeds[:, :pos, :], vis_tokens.to(inputs_embeds.dtype), inputs_embeds[:, pos + 1:, :]], dim=1)
    vis_attn = torch.ones((attention_mask.size(0), vis_len), device=attention_mask.device, dtype=attention_mask.dtype)
    new_mask = torch.cat([attention_mask[:, :pos], vis_attn, attention_mask[:, pos + 1:]], dim=1)
    return new_inputs, new_mask


def _row_spans(attention_mask: torch.Tensor) -> List[Tuple[int, int]]:
    """(first attended column, count) per row; the attended columns must be one contiguous run (left and/or right
    padding) - the only masks generate() produces or the reference feeds."""
    m = (attention_mask != 0).cpu()
    spans = []
    for r in m:
        nz = r.nonzero().flatten()
        if nz.numel() == 0:
            raise ValueError("generate: a row of attention_mask attends to nothing")
        lo, hi = int(nz[0]), int(nz[-1]) + 1
        if hi - lo != nz.numel():
            raise ValueError("generate: attention_mask rows must be contiguous runs of ones (padding on the sides only)")
        spans.append((lo, hi - lo))
    return spans


class DecodeState:
    """Device-resident state of one generate() call."""

    def __init__(self, tm, B: int, Lmax: int, gen_cols: int):
        c, dev = tm.config, tm.flat_w.device
        self.B, self.Lmax = B, Lmax
        nl = c.num_hidden_layers
        self.K = torch.zeros((nl, B, tm.Hkv, Lmax, tm.D), device=dev, dtype=BF16)
        self.V = torch.zeros((nl, B, tm.Hkv, Lmax, tm.D), device=dev, dtype=BF16)
        self.lens = torch.zeros(B, device=dev, dtype=torch.int32)
        self.step = torch.zeros(1, device=dev, dtype=torch.int32)
        self.finished = torch.zeros(B, device=dev, dtype=torch.int32)
        self.next_ids = torch.zeros(B, device=dev, dtype=torch.int32)
        self.generated = torch.zeros((B, gen_cols), device=dev, dtype=torch.int64)
        self.work = torch.zeros(B * 128 + 8, device=dev, dtype=F32)     # pick partials + overflow flag
        # B = 1 at Qwen3-4B's shape: all decoder layers of a step in ONE persistent launch (csrc/decode_layers.hip) instead of six
        # launches per layer; every other case decodes with one launch per projection
        self.persistent = None
        if (B == 1 and getattr(tm, "_fp8", None) is None and os.environ.get("VQ3_DECODE_PERSISTENT", "1") != "0"
                and ops.decode_layers_supported(c.hidden_size, c.intermediate_size, tm.Hq, tm.Hkv, tm.D, Lmax)):
            names = ("qkv", "o", "gu", "down", "ln1", "ln2", "qn", "kn")
            assert all(tm._w[f"l{i}.{n}"].is_contiguous() for i in range(nl) for n in names)
            self.persistent = dict(
                wtab=torch.tensor([[tm._w[f"l{i}.{n}"].data_ptr() for n in names] for i in range(nl)], dtype=torch.int64, device=dev),
                h=torch.empty((1, c.hidden_size), device=dev, dtype=BF16), workspace=ops.decode_layers_workspace(dev),
                barrier=torch.zeros(1024, device=dev, dtype=torch.int32),
                status=torch.zeros(1, device=dev, dtype=torch.int32))


def _prefill(tm, st: DecodeState, embeds: torch.Tensor, spans: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """Run the prompt through the training-path forward (batched MFMA GEMMs) and keep K/V. Rows of equal length go in
    one pass; otherwise row by row, each without its padding. Returns the last prompt position's hidden state [B, H]."""
    B, H = embeds.shape[0], embeds.shape[2]
    groups = {}
    for b, (lo, n) in enumerate(spans):
        groups.setdefault(n, []).append(b)
    h_last = torch.empty((B, H), device=embeds.device, dtype=BF16)
    for n, rows in groups.items():
        x = torch.stack([embeds[b, spans[b][0]:spans[b][0] + n] for b in rows], dim=0).contiguous()
        mask = torch.ones((len(rows), n), device=embeds.device, dtype=torch.int64)
        h, saved = tm.forward_hidden(x, mask, save=True)
        Lp = saved["L"]
        idx = torch.tensor(rows, device=embeds.device)
        for i, ctx in enumerate(saved["layers"]):
            st.K[i, idx, :, :n] = ctx["K"][:, :, :n]
            st.V[i, idx, :, :n] = ctx["V"][:, :, :n]
        h_last[idx] = h.view(len(rows), Lp, H)[:, n - 1]
        del saved
    st.lens.copy_(torch.tensor([n for _, n in spans], dtype=torch.int32))
    return h_last


def _logits_and_pick(tm, st: DecodeState, h: torch.Tensor, opts) -> None:
    c = tm.config
    logits = ops.skinny_linear(h, tm._w["embed"], n=tm.vocab, ln_w=tm._w["norm"], eps=c.rms_norm_eps)
    ops.greedy_pick(logits, st.work, st.generated, st.step, st.finished, opts["penalty"], opts["ngram"], opts["eos"],
                    opts["pad"], st.next_ids, tm.vocab)


def _decode_step(tm, st: DecodeState, cos, sin, opts) -> None:
    """One token for every row. Fixed launch sequence, every varying quantity read from device memory."""
    c = tm.config
    B, Hq, Hkv, D = st.B, tm.Hq, tm.Hkv, tm.D
    ps = st.persistent
    if ps is not None:
        h = ops.gather_rows(tm._w["embed"], st.next_ids, B, B, out=ps["h"])
        ops.decode_layers(ps["wtab"], h, ps["workspace"], cos, sin, st.lens, st.K, st.V, ps["barrier"], ps["status"], c.hidden_size,
                          c.intermediate_size, tm.Hq, tm.Hkv, c.rms_norm_eps, D ** -0.5)
        _logits_and_pick(tm, st, h, opts)
        ops.decode_advance(st.lens, B, st.step)
        return
    h = ops.gather_rows(tm._w["embed"], st.next_ids, B, B)
    f8 = tm._fp8 if (tm._fp8 is not None and B <= 2) else None     # e4m3 weight stream (config C5): half the bytes per token

    def proj(x, name, **kw):
        if f8 is not None:
            wq, ws = f8[name]
            return ops.skinny_linear_fp8(x, wq, ws, **kw)
        return ops.skinny_linear(x, tm._w[name], **kw)

    for i in range(c.num_hidden_layers):
        qkv = proj(h, f"l{i}.qkv", ln_w=tm._w[f"l{i}.ln1"], eps=c.rms_norm_eps)
        Q = ops.qwen_decode_qkprep(qkv, tm._w[f"l{i}.qn"], tm._w[f"l{i}.kn"], cos, sin, st.lens, st.K[i], st.V[i], B, Hq, Hkv,
                                   D, st.Lmax, c.rms_norm_eps)
        ao = ops.qwen_decode_attn(Q, st.K[i], st.V[i], st.lens, B, Hq, Hkv, D, st.Lmax, D ** -0.5)
        h_mid = proj(ao, f"l{i}.o", residual=h)
        gu = proj(h_mid, f"l{i}.gu", ln_w=tm._w[f"l{i}.ln2"], eps=c.rms_norm_eps)
        h = proj(gu, f"l{i}.down", residual=h_mid, swiglu=True)
    _logits_and_pick(tm, st, h, opts)
    ops.decode_advance(st.lens, B, st.step)         # the processed token is now cached; one more id has been picked


@torch.no_grad()
def generate(tm, inputs_embeds: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
             input_ids: Optional[torch.Tensor] = None, max_new_tokens: int = 20, do_sample: bool = False,
             num_beams: int = 1, repetition_penalty: float = 1.0, no_repeat_ngram_size: int = 0,
             eos_token_id: Union[int, Sequence[int], None] = None, pad_token_id: Optional[int] = None,
             use_graph: Optional[bool] = None, check_every: int = 8, return_stats: bool = False, **unused):
    """transformers `generate` for the greedy case. With `inputs_embeds` only the NEW tokens are returned (as
    transformers does); with `input_ids` the prompt is prepended and, like transformers, takes part in the repetition
    penalty and n-gram ban. Finished rows emit pad_token_id; generation stops when every row has produced eos or after
    max_new_tokens."""
    if do_sample or num_beams != 1:
        raise NotImplementedError("only greedy decoding (do_sample=False, num_beams=1) - what the reference's callers use")
    if (inputs_embeds is None) == (input_ids is None):
        raise ValueError("pass exactly one of inputs_embeds / input_ids")
    if max_new_tokens < 1:
        raise ValueError("max_new_tokens must be >= 1")
    dev = tm.flat_w.device
    if input_ids is not None:
        input_ids = input_ids.to(dev)
        embeds = tm.get_input_embeddings()(input_ids)
    else:
        embeds = inputs_embeds.to(dev, BF16)
    B, L0, H = embeds.shape
    if B > 8:
        raise ValueError("generate: at most 8 rows per call (skinny GEMM width)")
    if attention_mask is None:
        attention_mask = torch.ones((B, L0), device=dev, dtype=torch.int64)
    spans = _row_spans(attention_mask)
    eos = None
    if eos_token_id is not None:
        eos_list = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
        eos = torch.tensor(eos_list, device=dev, dtype=torch.int64)
        if pad_token_id is None:
            pad_token_id = eos_list[0]          # transformers' fallback (with a warning)
    opts = dict(penalty=float(repetition_penalty), ngram=int(no_repeat_ngram_size), eos=eos,
                pad=0 if pad_token_id is None else int(pad_token_id))
    n_prompt = L0 if input_ids is not None else 0
    Lmax = round_up(max(n for _, n in spans) + max_new_tokens, 64)
    st = DecodeState(tm, B, Lmax, n_prompt + max_new_tokens)
    if input_ids is not None:
        st.generated[:, :L0] = input_ids
        st.step.fill_(L0)
    cos, sin = tm.rope(Lmax)
    h_last = _prefill(tm, st, embeds, spans)
    _logits_and_pick(tm, st, h_last, opts)
    ops.decode_advance(None, B, st.step)
    if use_graph is None:
        use_graph = os.environ.get("VQ3_DECODE_GRAPH", "1") != "0"
    graph = None
    done_steps = 1
    if max_new_tokens > 1 and use_graph:
        # warm the allocator, then capture one step; replay mutates the same state tensors
        side = torch.cuda.Stream(device=dev)
        snap = None
        if st.persistent is not None:            # what the step mutates, should it have to be redone on the other route (see below)
            snap = (st.lens.clone(), st.step.clone(), st.finished.clone(), st.next_ids.clone(), st.generated.clone())
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _decode_step(tm, st, cos, sin, opts)
        torch.cuda.current_stream().wait_stream(side)
        done_steps += 1
        if st.persistent is not None and int(st.persistent["status"].item()) & 1:
            # The persistent kernel needs its 256 workgroups co-resident; a bounded barrier wait ran out (other tenants / streams hold CUs,
            # a CU mask): its step's hidden row and the cache rows it appended are garbage. Nothing is lost: the state is put back, the
            # K / V rows at position lens (the only ones the step wrote) are rewritten by the redo, and this call decodes with one launch
            # per projection from here on.
            import warnings
            warnings.warn("vq3: the persistent decode kernel's workgroups were not co-resident (a grid-barrier wait ran out); this "
                          "generate() call continues with one launch per projection (VQ3_DECODE_PERSISTENT=0 selects that route up front)")
            st.persistent = None
            for dst, src in zip((st.lens, st.step, st.finished, st.next_ids, st.generated), snap):
                dst.copy_(src)
            _decode_step(tm, st, cos, sin, opts)
        elif st.persistent is not None:
            ops.decode_layers_status(st.persistent["status"])   # (cache full)
        if max_new_tokens > 2:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                _decode_step(tm, st, cos, sin, opts)
    while done_steps < max_new_tokens:
        if eos is not None and (done_steps % check_every == 0 or graph is None) and bool(st.finished.all()):
            break
        if graph is not None:
            graph.replay()
            if st.persistent is not None and done_steps % check_every == 0:
                ops.decode_layers_status(st.persistent["status"])       # a failure mid-way surfaces within check_every tokens, not after all of them
        else:
            _decode_step(tm, st, cos, sin, opts)
            if st.persistent is not None and done_steps % check_every == 0:
                ops.decode_layers_status(st.persistent["status"])
        done_steps += 1
    if st.persistent is not None:
        ops.decode_layers_status(st.persistent["status"])
    gen = st.generated[:, : n_prompt + done_steps].cpu()
    # transformers stops right after the step in which the last row finished: trim what ran past it
    stop = done_steps
    if eos is not None:
        new = gen[:, n_prompt:]
        hit = torch.zeros_like(new, dtype=torch.bool)
        for e in eos.tolist():
            hit |= new == e
        first = torch.where(hit.any(1), hit.float().argmax(1), torch.full((B,), 1 << 30))
        if int(first.max()) < (1 << 30):
            stop = min(done_steps, int(first.max()) + 1)
    out = gen[:, : n_prompt + stop].to(dev)
    if return_stats:
        return out, {"steps_run": done_steps, "graph": graph is not None, "Lmax": Lmax, "persistent": st.persistent is not None}
    return out
