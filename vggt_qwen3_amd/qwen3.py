"""Qwen3 causal LM on the HIP kernels: forward to the loss and a hand-written backward.

Mirrors transformers' Qwen3ForCausalLM (modeling_qwen3.py:49-508) as the reference's VGGTQwen3VLM uses it
(src/models/vggt_qwen3_vlm.py:36-42,190,196-201): same module/parameter names (so reference checkpoints load),
`get_input_embeddings()`, `resize_token_embeddings()`, `config.hidden_size`. All arithmetic is in libvq3hip.so.

Memory layout (sized for 288 GB HBM3E: everything replicated, nothing sharded):
  flat_w   bf16  every parameter, one buffer; per layer [qkv | o | gate_up | down | 4 norm vectors] so the fused
                 QKV and gate|up GEMMs read one contiguous weight and a layer is one contiguous all-reduce bucket
  flat_g   bf16  gradients, same layout (parameters' .grad are views)
No transposed weight or activation copies exist: dgrad (dX = dY . W) reads W as the k-major B operand, wgrad
(dW = dY^T . X) reads dY and X as k-major A / B operands, and the attention products read K, V, Q, dO as stored
(vq3_gemm_bf16_nt transA / transB; ds_read_b64_tr_b16 fragment reads).
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .hostplan import PLAN
from .ops import BF16, F32, round_up


@dataclass
class Qwen3Config:
    hidden_size: int = 2560
    num_hidden_layers: int = 36
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 9728
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 5_000_000.0
    tie_word_embeddings: bool = True
    initializer_range: float = 0.02

    @classmethod
    def qwen3_4b(cls) -> "Qwen3Config":
        return cls()

    @classmethod
    def from_json(cls, path) -> "Qwen3Config":
        d = json.loads(Path(path).read_text())
        rp = d.get("rope_parameters") or {}
        return cls(hidden_size=d["hidden_size"], num_hidden_layers=d["num_hidden_layers"],
                   num_attention_heads=d["num_attention_heads"], num_key_value_heads=d["num_key_value_heads"],
                   head_dim=d.get("head_dim") or d["hidden_size"] // d["num_attention_heads"],
                   intermediate_size=d["intermediate_size"], vocab_size=d["vocab_size"],
                   rms_norm_eps=d.get("rms_norm_eps", 1e-6),
                   rope_theta=float(rp.get("rope_theta", d.get("rope_theta", 10000.0))),
                   tie_word_embeddings=d.get("tie_word_embeddings", True),
                   initializer_range=d.get("initializer_range", 0.02))


class _W(nn.Module):
    """Parameter holder with the HF module name (q_proj, input_layernorm, ...). Never called."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=True)


class _Embedding(_W):
    """`get_input_embeddings()(input_ids)` as the reference's inference scripts call it (qa_inference.py:189-190)."""

    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        ids = input_ids.to(self.weight.device).reshape(-1).to(torch.int32).contiguous()
        n = ids.numel()
        rows = ops.gather_rows(self.weight.detach(), ids, n, n)
        return rows.view(*input_ids.shape, self.weight.shape[1])


class _Attn(nn.Module):
    pass


class _MLP(nn.Module):
    pass


class _Layer(nn.Module):
    pass


class _Body(nn.Module):
    pass


class Qwen3ForCausalLM(nn.Module):
    def __init__(self, config: Qwen3Config, device="cuda", seed: Optional[int] = 0):
        super().__init__()
        self.config = config
        c = config
        H, I, D = c.hidden_size, c.intermediate_size, c.head_dim
        self.Hq, self.Hkv, self.D = c.num_attention_heads, c.num_key_value_heads, D
        if D != 128:
            raise ops._lib.Vq3Error("Qwen3 HIP path requires head_dim == 128")
        if H % 64 or I % 64 or (self.Hq * D) % 64:
            raise ops._lib.Vq3Error("hidden/intermediate sizes must be multiples of 64")
        self.nqkv = (self.Hq + 2 * self.Hkv) * D
        self.device_ = torch.device(device)
        self._rope_cache: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._wgrad_stream = None
        if self.device_.type == "cuda" and os.environ.get("VQ3_WGRAD_STREAM", "1") != "0":
            self._wgrad_stream = torch.cuda.Stream(device=self.device_)
        self._fp8 = None
        self._fp8T = None
        self._wt = None
        # deferred weight-gradient GEMMs (enable_wgrad_deferral): operand slabs, rows pending, forward/backward pairing ticket
        self._wd_depth = 1
        self._wd_slabs = None
        self._wd_cap = 0
        self._wd_rows = [0] * config.num_hidden_layers          # per layer: token rows waiting in its slabs
        self._wd_first_acc = [False] * config.num_hidden_layers  # per layer: does its next product add to flat_g or overwrite it
        self._wd_ticket = 0
        self._wd_count = 0                                       # backward passes since the last full flush
        # fused causal attention (csrc/qwen_flash.hip) covers up to 4 query heads per kv head (Qwen3-4B: 4);
        # VQ3_QWEN_FLASH=0 keeps the batched GEMM + softmax chain
        self._flash = self.Hq % self.Hkv == 0 and self.Hq // self.Hkv <= 4 and os.environ.get("VQ3_QWEN_FLASH", "1") != "0"
        self._alloc(c.vocab_size, seed)
        # weights written through load_state_dict() (any route) invalidate the e4m3 copies
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh_derived())

    # ------------------------------------------------------------------ storage
    def _layout(self, vocab: int):
        c = self.config
        H, I, D = c.hidden_size, c.intermediate_size, c.head_dim
        ent = [("embed", (vocab, H))]
        for i in range(c.num_hidden_layers):
            ent += [(f"l{i}.qkv", (self.nqkv, H)), (f"l{i}.o", (H, self.Hq * D)), (f"l{i}.gu", (2 * I, H)),
                    (f"l{i}.down", (H, I)), (f"l{i}.ln1", (H,)), (f"l{i}.ln2", (H,)), (f"l{i}.qn", (D,)),
                    (f"l{i}.kn", (D,))]
        ent.append(("norm", (H,)))
        off, table = 0, {}
        for name, shape in ent:
            n = math.prod(shape)
            table[name] = (off, shape)
            if name == "embed":  # zero rows up to a multiple of 64: the vocabulary is a GEMM contraction / k-major extent
                n = round_up(shape[0], 64) * shape[1]
            off += round_up(n, 64)  # keep every tensor 128-byte aligned
        return table, off

    def _alloc(self, vocab: int, seed: Optional[int], old: Optional[Dict[str, torch.Tensor]] = None):
        c = self.config
        dev = self.device_
        self.vocab = vocab
        self.table, total = self._layout(vocab)
        self.flat_w = torch.zeros(total, device=dev, dtype=BF16)
        self.flat_g = torch.zeros(total, device=dev, dtype=BF16)
        self._w = {n: self.flat_w[o:o + math.prod(s)].view(s) for n, (o, s) in self.table.items()}
        self._g = {n: self.flat_g[o:o + math.prod(s)].view(s) for n, (o, s) in self.table.items()}
        self.vocab_p = round_up(vocab, 64)
        if seed is not None:
            g = torch.Generator(device="cpu").manual_seed(seed)
            for n, w in self._w.items():
                if w.dim() == 2:
                    # chunked host-side normal init (std = initializer_range), like HF _init_weights
                    rows = w.shape[0]
                    step = max(1, (1 << 24) // w.shape[1])
                    for r in range(0, rows, step):
                        blk = torch.randn((min(step, rows - r), w.shape[1]), generator=g) * c.initializer_range
                        w[r:r + blk.shape[0]].copy_(blk.to(BF16))
                else:
                    w.fill_(1.0)
        if old is not None:
            for n, t in old.items():
                if n == "embed":
                    k = min(t.shape[0], vocab)
                    self._w["embed"][:k].copy_(t[:k])
                    if vocab > k:  # HF mean-resizing is random; use the mean row (deterministic)
                        self._w["embed"][k:].copy_(t.float().mean(0, keepdim=True).to(BF16).expand(vocab - k, -1))
                else:
                    self._w[n].copy_(t)
        self._build_modules()

    def _build_modules(self):
        c = self.config
        H, I, D = c.hidden_size, c.intermediate_size, c.head_dim
        body = _Body()
        body.embed_tokens = _Embedding(self._w["embed"])
        layers = []
        for i in range(c.num_hidden_layers):
            L = _Layer()
            a = _Attn()
            qkv = self._w[f"l{i}.qkv"]
            a.q_proj = _W(qkv[: self.Hq * D])
            a.k_proj = _W(qkv[self.Hq * D:(self.Hq + self.Hkv) * D])
            a.v_proj = _W(qkv[(self.Hq + self.Hkv) * D:])
            a.o_proj = _W(self._w[f"l{i}.o"])
            a.q_norm = _W(self._w[f"l{i}.qn"])
            a.k_norm = _W(self._w[f"l{i}.kn"])
            m = _MLP()
            gu = self._w[f"l{i}.gu"]
            m.gate_proj = _W(gu[:I])
            m.up_proj = _W(gu[I:])
            m.down_proj = _W(self._w[f"l{i}.down"])
            L.self_attn, L.mlp = a, m
            L.input_layernorm = _W(self._w[f"l{i}.ln1"])
            L.post_attention_layernorm = _W(self._w[f"l{i}.ln2"])
            layers.append(L)
        body.layers = nn.ModuleList(layers)
        body.norm = _W(self._w["norm"])
        self.model = body
        self.lm_head = _W(self._w["embed"])
        self.lm_head.weight = body.embed_tokens.weight  # tied (modeling_qwen3.py `_tied_weights_keys`)
        self._bind_grads()

    def _grad_views(self) -> Dict[str, torch.Tensor]:
        c = self.config
        I, D = c.intermediate_size, c.head_dim
        out = {"model.embed_tokens.weight": self._g["embed"], "model.norm.weight": self._g["norm"]}
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            qkv, gu = self._g[f"l{i}.qkv"], self._g[f"l{i}.gu"]
            out[p + "self_attn.q_proj.weight"] = qkv[: self.Hq * D]
            out[p + "self_attn.k_proj.weight"] = qkv[self.Hq * D:(self.Hq + self.Hkv) * D]
            out[p + "self_attn.v_proj.weight"] = qkv[(self.Hq + self.Hkv) * D:]
            out[p + "self_attn.o_proj.weight"] = self._g[f"l{i}.o"]
            out[p + "self_attn.q_norm.weight"] = self._g[f"l{i}.qn"]
            out[p + "self_attn.k_norm.weight"] = self._g[f"l{i}.kn"]
            out[p + "mlp.gate_proj.weight"] = gu[:I]
            out[p + "mlp.up_proj.weight"] = gu[I:]
            out[p + "mlp.down_proj.weight"] = self._g[f"l{i}.down"]
            out[p + "input_layernorm.weight"] = self._g[f"l{i}.ln1"]
            out[p + "post_attention_layernorm.weight"] = self._g[f"l{i}.ln2"]
        return out

    def _bind_grads(self):
        self.grad_views = self._grad_views()

    def publish_grads(self):
        """Expose the flat gradient buffer through the parameters' .grad (for stock torch optimisers / DDP)."""
        named = dict(self.named_parameters())
        for n, g in self.grad_views.items():
            named[n].grad = g

    def zero_grad_flat(self):
        self.flat_g.zero_()

    # ------------------------------------------------------------------ HF-compatible surface
    def get_input_embeddings(self):
        return self.model.embed_tokens

    def resize_token_embeddings(self, n: int):
        if n == self.vocab:
            return self.model.embed_tokens
        old = {k: v.clone() for k, v in self._w.items()}
        self.config.vocab_size = n
        self._alloc(n, None, old)
        return self.model.embed_tokens

    def load_hf_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        named = dict(self.named_parameters())
        missing = []
        with torch.no_grad():
            for n, p in named.items():
                if n in sd:
                    p.copy_(sd[n].to(device=p.device, dtype=p.dtype))
                elif n == "lm_head.weight" and "model.embed_tokens.weight" in sd:
                    pass
                else:
                    missing.append(n)
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}...")
        self.refresh_derived()
        return missing

    @classmethod
    def from_pretrained_dir(cls, path, device="cuda") -> "Qwen3ForCausalLM":
        """Local directory with config.json + *.safetensors (no network access is ever attempted)."""
        from safetensors.torch import load_file
        path = Path(path)
        cfg = Qwen3Config.from_json(path / "config.json")
        m = cls(cfg, device=device, seed=None)
        sd = {}
        for f in sorted(path.glob("*.safetensors")):
            sd.update(load_file(str(f)))
        m.load_hf_state_dict(sd, strict=True)
        return m

    def _pass_weights_gate(self) -> None:
        """An optimiser step Stage1Trainer left running on its side stream (overlap_optimizer) must be complete before anything reads the
        weights from another stream: the trainer leaves its event here (and on the VLM, vlm.py: _pass_weights_gate)."""
        ev = getattr(self, "_weights_gate", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)       # (the event stays: the VLM's own gate clears both)

    def generate(self, *args, **kwargs):
        """Greedy decoding with a KV cache (vggt_qwen3_amd/generate.py); transformers-compatible keywords."""
        from .generate import generate as _generate
        self._pass_weights_gate()
        return _generate(self, *args, **kwargs)

    # ------------------------------------------------------------------ helpers
    def rope(self, L: int):
        """cos/sin tables as HF builds them: fp32 angles, cat(freqs, freqs), cast to bf16 (modeling_qwen3.py:104-146).
        Parameter-free table, built once per sequence length on the host."""
        if L not in self._rope_cache:
            D, th = self.D, self.config.rope_theta
            inv = 1.0 / (th ** (torch.arange(0, D, 2, dtype=torch.float) / D))
            fr = (inv[:, None] @ torch.arange(L, dtype=torch.float)[None, :]).transpose(0, 1)
            emb = torch.cat((fr, fr), dim=-1)
            self._rope_cache[L] = (emb.cos().to(BF16).to(self.device_).contiguous(),
                                   emb.sin().to(BF16).to(self.device_).contiguous())
        return self._rope_cache[L]

    # ------------------------------------------------------------------ fp8 forward (BASELINE config C5)
    def enable_fp8_forward(self, on: bool = True, dgrad: Optional[bool] = None) -> None:
        """BASELINE config C5 ("Qwen3-4B fp8 weights"): the projections (q|k|v, o, gate|up, down of every layer) through the e4m3
        block-scaled-MFMA GEMMs: weights quantised per output channel (refresh with requantize_fp8() after every weight update),
        activations per token on the fly, fp32 accumulation - in the forward AND (dgrad, default on; VQ3_FP8_DGRAD=0 / dgrad=False =
        round 3's forward-only form) in the four input-gradient GEMMs dX = dY . W of every layer, which read the SAME e4m3 weights
        through a byte-transposed copy: W's per-output-channel scales run along that contraction, so they are folded into dY before
        its rows are quantised (vq3_quant_fp8_rows_scaled). Embedding / lm_head, norms, attention and the weight-gradient GEMMs stay
        bf16 (they read the saved bf16 activations and bf16 dY)."""
        self._pass_weights_gate()                 # (quantises the weights: an optimiser step on the trainer's side stream must be complete)
        self._fp8 = {} if on else None
        if dgrad is None:
            dgrad = os.environ.get("VQ3_FP8_DGRAD", "1") != "0"
        self._fp8T = {} if (on and dgrad) else None
        if on:
            self.requantize_fp8()

    def requantize_fp8(self) -> None:
        if getattr(self, "_fp8", None) is None:
            return
        f8t = getattr(self, "_fp8T", None)
        for i in range(self.config.num_hidden_layers):
            for k in ("qkv", "o", "gu", "down"):
                name = f"l{i}.{k}"
                w = self._w[name]
                if w.shape[1] % 128:
                    raise ops._lib.Vq3Error(f"fp8 forward needs in_features % 128 == 0, {name} has {w.shape[1]}")
                wq, ws = ops.quant_fp8_rows(w)
                self._fp8[name] = (wq, ws)
                # W^T in e4m3 for dX = dY . W (contraction = out_features: % 128; rows of 16 bytes: in_features % 16)
                if f8t is not None and w.shape[0] % 128 == 0 and w.shape[1] % 16 == 0:
                    prev = f8t.get(name)
                    f8t[name] = (ops.transpose_u8(wq, out=prev[0] if prev is not None else None), ws)

    # ------------------------------------------------------------------ W^T copies for the dgrad GEMMs
    # measured cold at 1200 rows: NT beats the k-major form by 10-18 % on q|k|v, o, gate|up and loses on down_proj; at the 9600 rows
    # of a merged pass the SwiGLU-fused down_proj dgrad runs at 0.77 PF/s k-major against > 1 PF/s NT, so it gets its copy too
    DGRAD_NT = tuple(os.environ.get("VQ3_DGRAD_NT_SET", "qkv,o,gu,down").split(","))

    def enable_dgrad_transposes(self, on: bool = True) -> None:
        """dX = dY . W reads W k-major (as stored) through transposed LDS reads - 10-18 % slower than the NT form on the
        q|k|v, o and gate|up shapes when the weights stream cold from HBM. With this on, those three keep a W^T copy per
        layer (+5.4 GB at Qwen3-4B) that refresh_derived() rebuilds after every weight update (10.9 GB of traffic, 2.7
        ms): worth it when updates are rarer than every ~4 micro-batches, which is what the trainer checks."""
        self._wt = {} if on else None
        if on:
            self.refresh_transposes()

    def refresh_transposes(self) -> None:
        if getattr(self, "_wt", None) is None:
            return
        for i in range(self.config.num_hidden_layers):
            for k in self.DGRAD_NT:
                name = f"l{i}.{k}"
                w = self._w[name]
                dst = self._wt.get(name)
                if dst is None:
                    dst = self._wt[name] = torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=BF16)
                ops.transpose_raw(w, dst, w.shape[0], w.shape[1], w.shape[0], w.stride(0), w.shape[0])

    def refresh_derived(self) -> None:
        """Rebuild everything computed FROM the weights (e4m3 copies, W^T copies); call after any weight update."""
        self.requantize_fp8()
        self.refresh_transposes()

    def _proj(self, x: torch.Tensor, name: str, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        f8 = getattr(self, "_fp8", None)
        if f8 is not None:
            wq, ws = f8[name]
            return ops.linear_fp8(x, wq, ws, residual=residual)
        return ops.linear(x, self._w[name], residual=residual)

    # ------------------------------------------------------------------ forward
    def _attention_fwd(self, i, xn, B, L, keymask, ctx, ao_out=None):
        """L is a multiple of 8 here (forward_hidden pads). Scores are materialised per (b, head) - at L = 200 the
        attention is < 1 % of the FLOPs - as batched GEMMs that read Q, K, V exactly as the prep kernel wrote them."""
        c = self.config
        Hq, Hkv, D, G = self.Hq, self.Hkv, self.D, self.Hq // self.Hkv
        cos, sin = self.rope(L)
        qkv = self._proj(xn, f"l{i}.qkv")
        Q, K, V, qr, kr = ops.qwen_qkprep_fwd(qkv, self._w[f"l{i}.qn"], self._w[f"l{i}.kn"], cos, sin, B, L, Hq, Hkv,
                                              D, c.rms_norm_eps, want_rstd=ctx is not None)
        if self._flash:
            # fused causal GQA attention: scores stay in registers; LSE is all the backward needs
            ao, lse = ops.qwen_flash_fwd(Q, K, V, keymask, B, L, Hq, Hkv, D, D ** -0.5, out=ao_out)
            if ctx is not None:
                ctx.update(qkv=qkv, Q=Q, K=K, V=V, qr=qr, kr=kr, lse=lse, ao=ao)
            return ao
        S = torch.empty((B * Hq, L, L), device=xn.device, dtype=F32)
        ops.gemm_raw(Q, K, S, L, L, D, D, D, L, nb1=B, nb2=Hq, b2divB=G, sA=(Hq * L * D, L * D),
                     sB=(Hkv * L * D, L * D), sC=(Hq * L * L, L * L), alpha=D ** -0.5)
        P = ops.softmax_fwd(S, keymask, Hq, L, L, True)
        ao = ao_out if ao_out is not None else torch.empty((B * L, Hq * D), device=xn.device, dtype=BF16)
        # O[b,h] = P[b,h] . V[b,h/G]  (V [L, D] is the k-major B operand)
        ops.gemm_raw(P, V, ao, L, D, L, L, D, Hq * D, nb1=B, nb2=Hq, b2divB=G, sA=(Hq * L * L, L * L),
                     sB=(Hkv * L * D, L * D), sC=(L * Hq * D, D), transB=True)
        if ctx is not None:
            ctx.update(qkv=qkv, Q=Q, K=K, V=V, qr=qr, kr=kr, P=P, ao=ao)
        return ao

    def forward_hidden(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, save: bool, plan_key=None):
        """36x decoder layer (modeling_qwen3.py:294-323, 367-427). Returns (h_last [B*L,H] pre-final-norm, saved)."""
        B, L0, H = inputs_embeds.shape
        c = self.config
        L = round_up(L0, 8)
        if L != L0:
            # exact: the extra positions are masked as keys and carry no label (k-major GEMM operands want L % 8 == 0)
            inputs_embeds = torch.cat([inputs_embeds, inputs_embeds.new_zeros((B, L - L0, H))], dim=1)
            attention_mask = torch.cat([attention_mask, attention_mask.new_zeros((B, L - L0))], dim=1)
        keymask = (attention_mask != 0).to(torch.uint8).contiguous()
        kv_parts = 1
        dkv_lds = os.environ.get("VQ3_QWEN_DKV_LDS", "1") != "0"
        if "VQ3_QWEN_KV_PARTS" in os.environ:
            kv_parts = int(os.environ["VQ3_QWEN_KV_PARTS"])
        elif save and self._flash and L >= 128 and not dkv_lds:
            # The dK/dV pass walks, per 32-key tile, every query block behind it - its longest serial chain. Key tiles without an
            # attended key leave at once, so when most of a padded batch's tiles are empty there are CUs to spare and two
            # workgroups share each live tile's walk (partial slabs, summed by the q/k-prep backward); with dense masks every
            # workgroup is live and splitting would only add rounds. One tiny host read per forward decides.
            nkb = (L + 31) // 32
            pad = nkb * 32 - L
            km = torch.nn.functional.pad(keymask, (0, pad)) if pad else keymask
            count = lambda: int(km.view(B, nkb, 32).any(-1).sum().item())
            live = PLAN.get(("live_key_tiles", L), plan_key, count) if plan_key is not None else count()
            kv_parts = 2 if live * 2 <= B * nkb else 1
        # (round 3: with the LDS-staged dK/dV kernel - the default - a q-block step costs ~1 us instead of ~4, a workgroup's fixed cost
        # dominates and ONE workgroup per live key tile wins (133.7 against 132.4 samples/s): no split, and no host read to decide it)
        h = inputs_embeds.reshape(B * L, H)
        saved: List[dict] = []
        wd_off = self._wd_begin(B * L) if save else None       # row offset of this micro-batch in the deferred-wgrad slabs (or None)
        wv = (lambda i, key: self._wd_view(i, key, wd_off, B * L)) if wd_off is not None else (lambda i, key: None)
        for i in range(c.num_hidden_layers):
            ctx = {} if save else None
            xn1, r1 = ops.rmsnorm_fwd(h, self._w[f"l{i}.ln1"], c.rms_norm_eps, want_rstd=True, out=wv(i, "qkv.X"))
            ao = self._attention_fwd(i, xn1, B, L, keymask, ctx, ao_out=wv(i, "o.X"))
            h_mid = self._proj(ao, f"l{i}.o", residual=h)
            xn2, r2 = ops.rmsnorm_fwd(h_mid, self._w[f"l{i}.ln2"], c.rms_norm_eps, want_rstd=True, out=wv(i, "gu.X"))
            if getattr(self, "_fp8", None) is None and ops.swiglu_fwd_fusable(B * L, c.intermediate_size, H):
                # gate|up projection with silu(gate) * up in its epilogue: act leaves with gu, no second pass over [rows, 2 I]
                gu, act = ops.gemm_swiglu_fwd(xn2, self._w[f"l{i}.gu"], act_out=wv(i, "down.X"), keep_gu=save)   # (no backward: act only)
            elif getattr(self, "_fp8", None) is not None and ops.swiglu_fwd_fusable(B * L, c.intermediate_size, H) and H % 128 == 0:
                wq, ws = self._fp8[f"l{i}.gu"]                      # the same fusion on the e4m3 kernel
                xq, xs = ops.quant_fp8_rows(xn2)
                gu, act = ops.gemm_fp8_ex(xq, xs, wq, ws, mode=1, out=wv(i, "down.X"), keep_gu=save)
            else:
                gu = self._proj(xn2, f"l{i}.gu")
                act = ops.silu_mul_fwd(gu, out=wv(i, "down.X"))
            h_out = self._proj(act, f"l{i}.down", residual=h_mid)
            if save:
                ctx.update(h_in=h, r1=r1, xn1=xn1, h_mid=h_mid, r2=r2, xn2=xn2, gu=gu, act=act)
                saved.append(ctx)
            h = h_out
        out = {"layers": saved, "B": B, "L": L, "L0": L0, "keymask": keymask, "kv_parts": kv_parts}
        if wd_off is not None:
            out["wd_off"], out["wd_ticket"] = wd_off, self._wd_ticket
        return h, out

    @staticmethod
    def label_rows(labels: torch.Tensor):
        """loss_utils.py:49-71: position t predicts labels[t+1]; rows whose shifted label is -100 are ignored.
        Returns (row indices i32 [n], targets i32 [n]) - host-visible n (one small sync, like the reference's
        `.nonzero()` on the <image> positions)."""
        B, L = labels.shape
        shift = torch.full_like(labels, -100)
        shift[:, :-1] = labels[:, 1:]
        flat = shift.reshape(-1)
        idx = (flat != -100).nonzero(as_tuple=False).squeeze(1)
        return idx.to(torch.int32), flat[idx].to(torch.int32)

    def loss_head(self, h_last: torch.Tensor, labels: torch.Tensor, save: bool, L: Optional[int] = None, plan_key=None,
                  groups: Optional[List[int]] = None):
        """Final RMSNorm + tied lm_head + shifted mean cross-entropy, evaluated only on the rows that carry a label
        (the other rows of the reference's [B,L,V] logits never reach the loss). Also leaves d(loss)/d(logits) in
        place for the backward.
        groups (sample counts, sum = B): the batch is several of the reference's micro-batches concatenated; each one's loss is the
        mean over ITS labelled rows (loss_utils.py:49-71 is applied per micro-batch) and the gradient of the SUM of those losses is
        prepared - exactly what the micro-batches produce one by one. Returns a loss vector [len(groups)] then."""
        c = self.config
        H = c.hidden_size
        if L is not None and labels.shape[1] < L:   # h_last rows follow the padded length of forward_hidden
            labels = torch.cat([labels, labels.new_full((labels.shape[0], L - labels.shape[1]), -100)], dim=1)
        if plan_key is not None:
            idx, tgt = PLAN.get(("label_rows", tuple(labels.shape)), plan_key, lambda: self.label_rows(labels))
        else:
            idx, tgt = self.label_rows(labels)
        n = int(idx.numel())
        if n == 0:
            return torch.full((len(groups),) if groups is not None else (), float("nan"), device=h_last.device, dtype=F32), None
        n8 = round_up(n, 8)                                  # zero rows: contribute nothing, keep K % 8 == 0 in backward
        hs = ops.gather_rows(h_last, idx, n, n8)
        hn, rstd = ops.rmsnorm_fwd(hs, self._w["norm"], c.rms_norm_eps, want_rstd=True)
        ldl = self.vocab_p
        logits = torch.zeros((n8, ldl), device=h_last.device, dtype=BF16)
        ops.gemm_raw(hn, self._w["embed"], logits, n, self.vocab, H, H, H, ldl)
        if groups is not None and len(groups) > 1:
            def rows_of_groups():
                edges = torch.tensor(groups, device=idx.device).cumsum(0)                          # sample index where each group ends
                gid = torch.bucketize(idx.long() // labels.shape[1], edges, right=True)          # group of every labelled row
                cnt = torch.bincount(gid, minlength=len(groups)).to(F32)
                scale = torch.zeros(n8, device=idx.device, dtype=F32)
                scale[:n] = 1.0 / cnt[gid]
                return gid, cnt, scale
            gid, cnt, scale = PLAN.get(("label_groups", tuple(labels.shape), tuple(groups)), plan_key, rows_of_groups) \
                if plan_key is not None else rows_of_groups()
            row_loss = torch.zeros(n8, device=h_last.device, dtype=F32)
            ops.cross_entropy_rows(logits, tgt, scale, row_loss, n, self.vocab)
            loss = torch.zeros(len(groups), device=h_last.device, dtype=F32).index_add_(0, gid, row_loss[:n]) / cnt   # 0/0 = NaN: no labels
        else:
            loss_sum = torch.zeros(1, device=h_last.device, dtype=F32)
            ops.cross_entropy_fwd_bwd(logits, tgt, loss_sum, n, self.vocab, 1.0 / n)
            loss = (loss_sum / n).reshape(())
            if groups is not None:
                loss = loss.reshape(1)
        ctx = dict(idx=idx, n=n, n8=n8, hs=hs, hn=hn, rstd=rstd, dlogits=logits) if save else None
        return loss, ctx

    def logits_all(self, h_last: torch.Tensor) -> torch.Tensor:
        """Full [rows, vocab] logits (inference / parity checks only)."""
        c = self.config
        hn = ops.rmsnorm_fwd(h_last, self._w["norm"], c.rms_norm_eps)
        out = torch.empty((h_last.shape[0], self.vocab_p), device=h_last.device, dtype=BF16)
        ops.gemm_raw(hn, self._w["embed"], out, h_last.shape[0], self.vocab, c.hidden_size, c.hidden_size,
                     c.hidden_size, self.vocab_p)
        return out[:, : self.vocab]

    # ------------------------------------------------------------------ deferred weight gradients
    # dW = sum over the window's micro-batches of dY_j^T . X_j is ONE product over the concatenated token rows: [dY_1; dY_2; ..]^T .
    # [X_1; X_2; ..]. Run per micro-batch (contraction = 1200 rows) every weight-gradient GEMM re-reads and re-writes its whole bf16
    # gradient (16 GB of HBM traffic per micro-batch at Qwen3-4B) behind 19 K steps; run once per `depth` micro-batches the same kernels
    # reach 0.9-1.1 PF/s instead of 0.54-0.70 (tools/bench_wgrad_k.py) and the partial sums of the group stay in the MFMA's f32
    # accumulators. Cost: the operands must outlive their micro-batch - per layer eight bf16 slabs [depth * rows, cols] (99 KB per
    # token row: 34 GB at depth 8, rows 1200, 36 layers), which the producing kernels write in place (`out=`), so nothing is copied.
    # The layers do not all multiply on the same micro-batch: layer i does when (micro-batch index + i) % depth == depth - 1, so every
    # backward pass carries 1 / depth of the layers' (long) products on the weight-gradient stream, where they fill the CUs that the
    # 200-tile dgrad GEMMs of the main stream leave idle - all of them on one micro-batch in `depth` would run alone after its backward.
    WD_KEYS = ("down.dY", "down.X", "gu.dY", "gu.X", "o.dY", "o.X", "qkv.dY", "qkv.X")

    def _wd_cols(self, key: str) -> int:
        c = self.config
        I, H = c.intermediate_size, c.hidden_size
        return {"down.dY": H, "down.X": I, "gu.dY": 2 * I, "gu.X": H, "o.dY": H, "o.X": self.Hq * self.D,
                "qkv.dY": (self.Hq + 2 * self.Hkv) * self.D, "qkv.X": H}[key]

    def enable_wgrad_deferral(self, depth: int) -> None:
        """depth micro-batches share one weight-gradient GEMM per projection (1 = off: one per micro-batch, no slabs). The caller's
        loop must alternate forward_hidden(save=True) / backward_hidden and pass flush=True on the last micro-batch of every
        accumulation window (Stage1Trainer does)."""
        if any(self._wd_rows):
            raise RuntimeError("enable_wgrad_deferral: a group of micro-batches is pending; flush it first")
        self._wd_depth = max(1, int(depth))
        if self._wd_depth == 1:
            self._wd_slabs, self._wd_cap = None, 0

    def _wd_begin(self, rows: int):
        if self._wd_depth <= 1:
            return None
        if self._wd_slabs is None or rows > self._wd_cap:
            if any(self._wd_rows):
                self.flush_deferred()
            per_row = sum(self._wd_cols(k) for k in self.WD_KEYS) * 2 * self.config.num_hidden_layers
            depth = self._wd_depth
            free = torch.cuda.mem_get_info(self.device_)[0] if self.device_.type == "cuda" else 1 << 62
            held = 0 if self._wd_slabs is None else per_row * self._wd_cap
            while depth > 1 and per_row * rows * depth > 0.6 * (free + held):     # leave room for the step itself
                depth //= 2
            if depth <= 1:
                self._wd_depth, self._wd_slabs, self._wd_cap = 1, None, 0
                return None
            self._wd_slabs = None                                  # release before the new allocation
            cap = rows * depth
            self._wd_slabs = {(i, k): torch.zeros((cap, self._wd_cols(k)), device=self.device_, dtype=BF16)
                              for i in range(self.config.num_hidden_layers) for k in self.WD_KEYS}
            self._wd_cap = cap
        if max(self._wd_rows) + rows > self._wd_cap:               # (rows vary with trimmed padding) no room left: flush what is pending
            self.flush_deferred()
        self._wd_ticket += 1
        return list(self._wd_rows)                                 # this micro-batch's row offset in every layer's slabs

    def _wd_view(self, i: int, key: str, off, rows: int) -> torch.Tensor:
        return self._wd_slabs[(i, key)][off[i]:off[i] + rows]

    def _wd_flush_layer(self, i: int, rows: int):
        for name in ("down", "gu", "o", "qkv"):
            self._wgrad(f"l{i}.{name}", self._wd_slabs[(i, name + ".dY")][:rows], self._wd_slabs[(i, name + ".X")][:rows],
                        self._wd_first_acc[i], slab=True)
        self._wd_rows[i] = 0

    def flush_deferred(self, layer_done=None) -> None:
        """Weight-gradient GEMMs over the rows still pending (a window that ends on a micro-batch without labels, a slab resize, a
        look at flat_g in the middle of a window)."""
        for i in reversed(range(self.config.num_hidden_layers)):
            if self._wd_rows[i]:
                self._wd_flush_layer(i, self._wd_rows[i])
            if layer_done is not None:
                self.join_wgrad_stream()
                layer_done(i)
        self.join_wgrad_stream()
        self._wd_count = 0

    # ------------------------------------------------------------------ backward
    def _wgrad(self, name: str, dY: torch.Tensor, X: torch.Tensor, accumulate: bool, slab: bool = False):
        """dW[N,K] (+)= dY^T[N,M] . X[M,K]: dY and X are read as stored (k-major A and B operands, contraction = token
        rows). Weight gradients have no consumer inside the backward, so they are enqueued on a second HIP stream:
        their GEMM tails overlap the dgrad chain on the main one."""
        side = self._wgrad_stream
        if side is None:
            self._wgrad_now(name, dY, X, accumulate)
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._wgrad_now(name, dY, X, accumulate)
        if not slab:                                     # (slabs live as long as the model)
            dY.record_stream(side)
            X.record_stream(side)

    def _wgrad_now(self, name: str, dY: torch.Tensor, X: torch.Tensor, accumulate: bool):
        M, N = dY.shape
        K = X.shape[1]
        g = self._g[name]
        ops.gemm_raw(dY, X, g, N, K, M, dY.stride(0), X.stride(0), g.shape[1], accumulate=accumulate, transA=True,
                     transB=True)

    def join_wgrad_stream(self):
        """Make the current stream wait for every weight-gradient kernel enqueued so far."""
        if self._wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self._wgrad_stream)

    def _dgrad(self, dY: torch.Tensor, name: str) -> torch.Tensor:
        """dX[M,K] = dY[M,N] . W[N,K]: W as stored is the k-major B operand (contraction = its rows)."""
        W = self._w[name]
        M, N = dY.shape
        f8t = getattr(self, "_fp8T", None)
        if f8t is not None and name in f8t and dY.is_contiguous():
            wqt, ws = f8t[name]                                     # e4m3 W^T [K, N]; dY's columns carry W's per-channel scales
            q, t = ops.quant_fp8_rows_scaled(dY, ws)
            return ops.gemm_fp8_ex(q, t, wqt, None, mode=0)
        wt = getattr(self, "_wt", None)
        if wt is not None and name in wt and dY.is_contiguous():
            return ops.linear(dY, wt[name])                      # NT: B = W^T [K, N], contraction along its rows' columns
        dX = torch.empty((M, W.shape[1]), device=dY.device, dtype=BF16)
        ops.gemm_raw(dY, W, dX, M, W.shape[1], N, dY.stride(0), W.shape[1], W.shape[1], transB=True)
        return dX

    def _norm_wgrad(self, name: str, dw_f32: torch.Tensor, accumulate: bool):
        ops.f32_to_bf16_acc(dw_f32, self._g[name], accumulate)

    def _attention_bwd(self, i, ctx, d_ao, B, L):
        """d_ao: d(loss)/d(attention output), head-major [B, Hq, L, D]. No operand is transposed in memory."""
        Hq, Hkv, D, G = self.Hq, self.Hkv, self.D, self.Hq // self.Hkv
        dev = d_ao.device
        Q, K, V, P = ctx["Q"], ctx["K"], ctx["V"], ctx["P"]
        # dP[b,h] = dAO[b,h] . V[b,h/G]^T
        dP = torch.empty((B * Hq, L, L), device=dev, dtype=F32)
        ops.gemm_raw(d_ao, V, dP, L, L, D, D, D, L, nb1=B, nb2=Hq, b2divB=G, sA=(Hq * L * D, L * D),
                     sB=(Hkv * L * D, L * D), sC=(Hq * L * L, L * L))
        dS = ops.softmax_bwd(P, dP, L, D ** -0.5)
        # dQ[b,h] = dS[b,h] . K[b,h/G]            (K [L, D]: k-major B)
        dQ = torch.empty((B, Hq, L, D), device=dev, dtype=BF16)
        ops.gemm_raw(dS, K, dQ, L, D, L, L, D, D, nb1=B, nb2=Hq, b2divB=G, sA=(Hq * L * L, L * L),
                     sB=(Hkv * L * D, L * D), sC=(Hq * L * D, L * D), transB=True)
        # dK[b,kv] = sum_g dS[b,h]^T . Q[b,h] ; dV[b,kv] = sum_g P[b,h]^T . dAO[b,h]: the G heads of a kv group are
        # contiguous, so (g, query) is one contraction of length G*L over k-major operands [G*L, L] and [G*L, D]
        dK = torch.empty((B, Hkv, L, D), device=dev, dtype=BF16)
        dV = torch.empty((B, Hkv, L, D), device=dev, dtype=BF16)
        for A_, B_, C_ in ((dS, Q, dK), (P, d_ao, dV)):
            ops.gemm_raw(A_, B_, C_, L, D, G * L, L, D, D, nb1=B, nb2=Hkv, sA=(Hq * L * L, G * L * L),
                         sB=(Hq * L * D, G * L * D), sC=(Hkv * L * D, L * D), transA=True, transB=True)
        return dQ, dK, dV

    def backward_hidden(self, saved, dh: torch.Tensor, accumulate: bool, gscale_hook=None, layer_done=None, flush: bool = True):
        """Backward through the decoder stack. dh: d(loss)/d(h_last) [B*L, H] bf16. Gradients go to flat_g
        (accumulate=False overwrites). layer_done(i) is called when layer i's gradients are final (DP overlap hook).
        flush (only with enable_wgrad_deferral): False keeps this micro-batch's weight-gradient operands in the slabs for a later
        micro-batch of the same window to multiply; the window's last micro-batch must pass True. Returns d(inputs_embeds) [B*L, H]."""
        c = self.config
        B, L = saved["B"], saved["L"]
        H, D = c.hidden_size, self.D
        dev = dh.device
        cos, sin = self.rope(L)
        norm_jobs: list = []       # the layer's four norm-weight column sums, run in one launch at its end
        M = B * L
        wd_off = saved.get("wd_off")
        if wd_off is not None:
            if saved.get("wd_ticket") != self._wd_ticket or wd_off != self._wd_rows:
                raise RuntimeError("deferred weight gradients: forward_hidden(save=True) and backward_hidden must alternate "
                                   "(another forward has reused this micro-batch's operand rows)")
            wv = lambda i, key: self._wd_view(i, key, wd_off, M)
            top = wv(c.num_hidden_layers - 1, "down.dY")
            top.copy_(dh)
            dh = top
            phase = self._wd_count
            self._wd_count += 1
        else:
            wv = lambda i, key: None
        for i in reversed(range(c.num_hidden_layers)):
            ctx = saved["layers"][i]
            # down_proj
            # d(act) = dh . W_down with the SwiGLU backward in the GEMM epilogue: d(gate | up) leaves directly
            wt = getattr(self, "_wt", None)
            f8t = getattr(self, "_fp8T", None)
            if f8t is not None and f"l{i}.down" in f8t and dh.is_contiguous() and ctx["gu"].is_contiguous() and \
                    (wv(i, "gu.dY") is None or wv(i, "gu.dY").is_contiguous()):
                wqt, ws = f8t[f"l{i}.down"]
                q, t = ops.quant_fp8_rows_scaled(dh, ws)
                dgu = ops.gemm_fp8_ex(q, t, wqt, None, mode=2, gu=ctx["gu"], out=wv(i, "gu.dY"))
            elif wt is not None and f"l{i}.down" in wt and dh.is_contiguous():
                dgu = ops.gemm_swiglu_bwd(dh, wt[f"l{i}.down"], ctx["gu"], transB=False, out=wv(i, "gu.dY"))
            else:
                dgu = ops.gemm_swiglu_bwd(dh, self._w[f"l{i}.down"], ctx["gu"], transB=True, out=wv(i, "gu.dY"))
            if wd_off is None:
                self._wgrad(f"l{i}.down", dh, ctx["act"], accumulate)
            d_xn2 = self._dgrad(dgu, f"l{i}.gu")
            if wd_off is None:
                self._wgrad(f"l{i}.gu", dgu, ctx["xn2"], accumulate)
            dh_mid = ops.rmsnorm_bwd(d_xn2, ctx["h_mid"], self._w[f"l{i}.ln2"], ctx["r2"], dh, self._g[f"l{i}.ln2"],
                                     accumulate, defer=norm_jobs, out=wv(i, "o.dY"))
            # o_proj
            # d(attention out) head-major [B, Hq, L, D]: one batch per (b, head) over the column block of W_o
            Wo = self._w[f"l{i}.o"]
            if self._flash:
                d_ao = self._dgrad(dh_mid, f"l{i}.o")                   # token-major [B*L, Hq*D]: one plain GEMM
                if wd_off is None:
                    self._wgrad(f"l{i}.o", dh_mid, ctx["ao"], accumulate)
                dQ, dK, dV = ops.qwen_flash_bwd(ctx["Q"], ctx["K"], ctx["V"], saved["keymask"], ctx["ao"], d_ao, ctx["lse"],
                                                B, L, self.Hq, self.Hkv, D, D ** -0.5, kv_parts=saved.get("kv_parts", 1))
            else:
                d_ao = torch.empty((B, self.Hq, L, D), device=dev, dtype=BF16)
                ops.gemm_raw(dh_mid, Wo, d_ao, L, D, H, H, self.Hq * D, D, nb1=B, nb2=self.Hq, sA=(L * H, 0), sB=(0, D),
                             sC=(self.Hq * L * D, L * D), transB=True)
                if wd_off is None:
                    self._wgrad(f"l{i}.o", dh_mid, ctx["ao"], accumulate)
                dQ, dK, dV = self._attention_bwd(i, ctx, d_ao, B, L)
            dqkv = ops.qwen_qkprep_bwd(dQ, dK, dV, ctx["qkv"], self._w[f"l{i}.qn"], self._w[f"l{i}.kn"], cos, sin,
                                       ctx["qr"], ctx["kr"], self._g[f"l{i}.qn"], self._g[f"l{i}.kn"], accumulate, B, L,
                                       self.Hq, self.Hkv, D, defer=norm_jobs, out=wv(i, "qkv.dY"))
            d_xn1 = self._dgrad(dqkv, f"l{i}.qkv")
            if wd_off is None:
                self._wgrad(f"l{i}.qkv", dqkv, ctx["xn1"], accumulate)
            else:
                if wd_off[i] == 0:
                    self._wd_first_acc[i] = accumulate     # the group's products overwrite or add as its first micro-batch would have
                if flush or (phase + i) % self._wd_depth == self._wd_depth - 1 or wd_off[i] + 2 * M > self._wd_cap:
                    self._wd_flush_layer(i, wd_off[i] + M)   # the layer's four products over every pending row, this micro-batch's included
                else:
                    self._wd_rows[i] = wd_off[i] + M
            dh = ops.rmsnorm_bwd(d_xn1, ctx["h_in"], self._w[f"l{i}.ln1"], ctx["r1"], dh_mid, self._g[f"l{i}.ln1"],
                                 accumulate, defer=norm_jobs, out=wv(i - 1, "down.dY") if (wd_off is not None and i > 0) else None)
            ops.colsum_flush(norm_jobs)
            if layer_done is not None:
                self.join_wgrad_stream()
                layer_done(i)
        self.join_wgrad_stream()
        if wd_off is not None and flush:
            self._wd_count = 0
        return dh

    @staticmethod
    def _lm_dgrad_slices(Vp: int, ntile: int) -> int:
        """K slices of the lm_head's input-gradient product: the divisor of Vp / 64 (slices stay whole 64-element K tiles) nearest to one
        workgroup per CU for `ntile` output tiles; 1 = no such divisor (the f32-atomic split takes over; VQ3_LMHEAD_DGRAD_ATOMIC=1 forces it)."""
        if os.environ.get("VQ3_LMHEAD_DGRAD_ATOMIC") == "1" or Vp % 64:
            return 1
        q = Vp // 64
        want = max(2, min(64, 256 // max(1, ntile)))
        divs = [d for d in range(2, min(q, 128) + 1) if q % d == 0]
        if not divs:
            return 1
        return min(divs, key=lambda d: (abs(d - want), d))

    def backward_loss_head(self, head_ctx, rows: int, gscale: float, accumulate: bool) -> torch.Tensor:
        """Backward of loss_head: returns d(loss)/d(h_last) [rows, H] (zero on rows without a label) and writes the
        lm_head contribution to the tied embedding gradient."""
        c = self.config
        H = c.hidden_size
        dev = head_ctx["hn"].device
        n, n8, dlog = head_ctx["n"], head_ctx["n8"], head_ctx["dlogits"]
        Vp = self.vocab_p
        E = self.flat_w[self.table["embed"][0]: self.table["embed"][0] + Vp * H].view(Vp, H)      # incl. the zero pad rows
        dE = self.flat_g[self.table["embed"][0]: self.table["embed"][0] + Vp * H].view(Vp, H)
        # d(hn)[n8, H] = dlogits[n8, Vp] . E[Vp, H]       (E: k-major B; pad rows/cols are zero on both sides)
        # n8 is a handful of rows and K = 152 000: split the contraction over ~one workgroup per CU
        ntile = ((n8 + 127) // 128) * ((H + 127) // 128)
        S = self._lm_dgrad_slices(Vp, ntile)
        if S > 1:
            # K slices as a BATCHED product into f32 slabs [S, n8, H], summed in slice order (vq3_colsum_f32): no atomics, so the whole
            # backward is a fixed sequence of f32 sums - the same bits run after run (round 5; the atomic split below let a 1e-7
            # difference here grow into bf16-ulp noise on every gradient downstream). S * n8 * H * 4 B of slab: 20-40 MB per pass.
            Ks = Vp // S
            slabs = torch.empty((S, n8 * H), device=dev, dtype=F32)
            ops.gemm_raw(dlog, E, slabs, n8, H, Ks, Vp, H, H, alpha=gscale, transB=True, nb1=S, sA=(Ks, 0), sB=(Ks * H, 0), sC=(n8 * H, 0))
            d_hn32 = ops.colsum_f32(slabs).view(n8, H)
        else:
            d_hn32 = torch.zeros((n8, H), device=dev, dtype=F32)
            ops.gemm_raw(dlog, E, d_hn32, n8, H, Vp, Vp, H, H, alpha=gscale, transB=True, ksplit=max(2, 256 // ntile))
        d_hn = ops.cast(d_hn32, BF16)
        # dE[Vp, H] (+)= dlogits^T[Vp, n8] . hn[n8, H]    (both k-major; contraction = the n8 selected rows)
        ops.gemm_raw(dlog, head_ctx["hn"], dE, Vp, H, n8, Vp, H, H, accumulate=accumulate, alpha=gscale, transA=True,
                     transB=True)
        d_hs = ops.rmsnorm_bwd(d_hn, head_ctx["hs"], self._w["norm"], head_ctx["rstd"], None, self._g["norm"], accumulate)
        dh = torch.zeros((rows, H), device=dev, dtype=BF16)
        ops.scatter_rows(d_hs, head_ctx["idx"], dh, n, False)
        return dh
