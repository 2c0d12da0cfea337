"""Perceiver projector (projector_perceiver.py:30-82) restated on its state dict; eval mode (dropout off)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def mha_cross(latents, context, sd, pre, heads):
    """nn.MultiheadAttention(batch_first=True)(latents, context, context), math path
    (torch/nn/functional.py multi_head_attention_forward): packed in_proj rows [q|k|v], q scaled by
    head_dim**-0.5 BEFORE q.k^T, softmax, PV, out_proj."""
    B, N, D = latents.shape
    T = context.shape[1]
    W, b = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    q = F.linear(latents, W[:D], b[:D])
    k = F.linear(context, W[D:2 * D], b[D:2 * D])
    v = F.linear(context, W[2 * D:], b[2 * D:])
    hd = D // heads
    q = q.view(B, N, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    q = q * (hd ** -0.5)
    w = torch.softmax(torch.matmul(q, k.transpose(-2, -1)), dim=-1)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, N, D)
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def layer(latents, context, sd, pre, heads):
    """projector_perceiver.py:44-50 - post-norm; every layer cross-attends to the same context."""
    D = latents.shape[-1]
    a = mha_cross(latents, context, sd, pre + "self_attn.", heads)
    x = F.layer_norm(latents + a, (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    m = F.linear(F.gelu(F.linear(x, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])), sd[pre + "mlp.3.weight"],
                 sd[pre + "mlp.3.bias"])
    return F.layer_norm(x + m, (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)


def projector(tokens, sd, heads: int, num_layers: int, pre: str = ""):
    """projector_perceiver.py:70-82."""
    B = tokens.shape[0]
    ctx = F.linear(tokens, sd[pre + "in_proj.weight"], sd[pre + "in_proj.bias"])
    lat = sd[pre + "latents"].unsqueeze(0).expand(B, -1, -1)
    for i in range(num_layers):
        lat = layer(lat, ctx, sd, f"{pre}layers.{i}.", heads)
    return F.linear(lat, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])
