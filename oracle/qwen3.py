"""Qwen3 causal LM, restated functionally on a HF-named state dict (see package docstring for file:line anchors)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F


@dataclass
class Qwen3Cfg:
    hidden_size: int = 2560
    num_hidden_layers: int = 36
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 9728
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 5_000_000.0


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """modeling_qwen3.py:59-64 - note the cast back to the input dtype BEFORE the weight multiply."""
    dt = x.dtype
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(dt)


def rope_tables(L: int, head_dim: int, theta: float, dtype) -> tuple:
    """modeling_qwen3.py:104-146: fp32 freqs, cat(freqs, freqs), cos/sin cast to the activation dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    pos = torch.arange(L, dtype=torch.float)
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def causal_padding_bias(attention_mask: torch.Tensor, L: int, dtype) -> torch.Tensor:
    """Additive mask [B,1,L,L]: key j visible to query i iff j <= i and attention_mask[b,j] != 0
    (modeling_qwen3.py:403 create_causal_mask)."""
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool))
    vis = causal[None, :, :] & (attention_mask[:, None, :] != 0)
    bias = torch.zeros(vis.shape, dtype=dtype).masked_fill(~vis, torch.finfo(dtype).min)
    return bias[:, None]


def attention(x, sd, pre, cfg: Qwen3Cfg, cos, sin, bias):
    """modeling_qwen3.py:241-280 with the eager attention of :185-207."""
    B, L, _ = x.shape
    hd, nq, nkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    q = F.linear(x, sd[pre + "q_proj.weight"]).view(B, L, nq, hd)
    k = F.linear(x, sd[pre + "k_proj.weight"]).view(B, L, nkv, hd)
    v = F.linear(x, sd[pre + "v_proj.weight"]).view(B, L, nkv, hd)
    q = rmsnorm(q, sd[pre + "q_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    k = rmsnorm(k, sd[pre + "k_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    v = v.transpose(1, 2)
    c, s = cos[None, None], sin[None, None]
    q = (q * c) + (rotate_half(q) * s)
    k = (k * c) + (rotate_half(k) * s)
    rep = nq // nkv
    k = k.repeat_interleave(rep, dim=1)
    v = v.repeat_interleave(rep, dim=1)
    w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    w = w + bias
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, L, nq * hd)
    return F.linear(o, sd[pre + "o_proj.weight"])


def mlp(x, sd, pre):
    """modeling_qwen3.py:81-83."""
    return F.linear(F.silu(F.linear(x, sd[pre + "gate_proj.weight"])) * F.linear(x, sd[pre + "up_proj.weight"]),
                    sd[pre + "down_proj.weight"])


def decoder_layer(x, sd, i, cfg, cos, sin, bias):
    """modeling_qwen3.py:294-323."""
    p = f"model.layers.{i}."
    h = x + attention(rmsnorm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps), sd, p + "self_attn.", cfg, cos,
                      sin, bias)
    return h + mlp(rmsnorm(h, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps), sd, p + "mlp.")


def model_forward(inputs_embeds, attention_mask, sd, cfg: Qwen3Cfg, collect: Optional[List] = None):
    """modeling_qwen3.py:367-427 -> final-norm hidden states [B,L,H]."""
    B, L, _ = inputs_embeds.shape
    cos, sin = rope_tables(L, cfg.head_dim, cfg.rope_theta, inputs_embeds.dtype)
    bias = causal_padding_bias(attention_mask, L, inputs_embeds.dtype)
    h = inputs_embeds
    for i in range(cfg.num_hidden_layers):
        h = decoder_layer(h, sd, i, cfg, cos, sin, bias)
        if collect is not None:
            collect.append(h)
    return rmsnorm(h, sd["model.norm.weight"], cfg.rms_norm_eps)


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """loss_utils.py:49-71: upcast, shift labels left by one (pad with -100), mean CE over non-ignored."""
    V = logits.shape[-1]
    logits = logits.float()
    labels = F.pad(labels, (0, 1), value=-100)[..., 1:].contiguous()
    return F.cross_entropy(logits.view(-1, V), labels.view(-1), ignore_index=-100, reduction="mean")


def causal_lm(inputs_embeds, attention_mask, labels, sd, cfg: Qwen3Cfg, collect=None):
    """modeling_qwen3.py:448-508 (tied lm_head). Returns (loss or None, logits)."""
    h = model_forward(inputs_embeds, attention_mask, sd, cfg, collect)
    w = sd.get("lm_head.weight", sd["model.embed_tokens.weight"])
    logits = F.linear(h, w)
    loss = causal_lm_loss(logits, labels) if labels is not None else None
    return loss, logits
