"""VGGT aggregator restated on CPU (PARITY UNPINNED for the alternating frame/global stage; the DINOv2 backbone stage
is pinned by tests/golden/dinov2_tiny.npz - see dino_tokens() and the package docstring).

The reference repository holds neither the `vggt` package nor its weights nor any vector pinning its output; the
only in-repo facts are the call site `VGGT(img_size=518, patch_size=14, embed_dim=1024, ...).aggregator(images)`
(src/models/vggt_qwen3_vlm.py:75-83,144) and the consumer's expectations (`list[-1]` is [B, S, P, 2*embed_dim],
:148-156). This file restates the published architecture of facebookresearch/vggt (pinned version: none in the
reference - env/environment.yml does not list it): models/aggregator.py (alternating frame/global attention,
camera+register tokens, ImageNet normalisation), layers/block.py + attention.py (pre-LN, qkv, q/k LayerNorm, SDPA,
LayerScale), layers/rope.py (2-D RoPE, frequency 100) and layers/vision_transformer.py (DINOv2 ViT with register
tokens, bicubic-antialias pos-embed interpolation). It checks the HIP path for self-consistency only.
State-dict keys are the upstream names relative to `aggregator.`.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def rope2d(t: torch.Tensor, pos: torch.Tensor, freq: float) -> torch.Tensor:
    """t [G, NH, N, 64], pos [G, N, 2] (y, x) integer positions."""
    fd = t.shape[-1] // 2
    maxpos = int(pos.max()) + 1
    inv = 1.0 / (freq ** (torch.arange(0, fd, 2).float() / fd))
    ang = torch.einsum("i,j->ij", torch.arange(maxpos).float(), inv).to(t.dtype)
    ang = torch.cat((ang, ang), dim=-1)
    cos_c, sin_c = ang.cos(), ang.sin()

    def rot(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)

    def one(x, p):
        c = F.embedding(p, cos_c)[:, None]
        s = F.embedding(p, sin_c)[:, None]
        return x * c + rot(x) * s

    v, h = t.chunk(2, dim=-1)
    return torch.cat((one(v, pos[..., 0]), one(h, pos[..., 1])), dim=-1)


def block(x, sd, pre, NH, pos, *, qk_norm, eps, freq):
    """x [G, N, C]. Block.forward: x + ls1(attn(norm1(x))) ; x + ls2(mlp(norm2(x)))."""
    G, N, C = x.shape
    hd = C // NH
    h = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps)
    qkv = F.linear(h, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"]).reshape(G, N, 3, NH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if qk_norm:
        q = F.layer_norm(q, (hd,), sd[pre + "attn.q_norm.weight"], sd[pre + "attn.q_norm.bias"], 1e-5)
        k = F.layer_norm(k, (hd,), sd[pre + "attn.k_norm.weight"], sd[pre + "attn.k_norm.bias"], 1e-5)
    if pos is not None:
        q, k = rope2d(q, pos, freq), rope2d(k, pos, freq)
    a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(G, N, C)
    a = F.linear(a, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    x = x + a * sd[pre + "ls1.gamma"]
    h = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    m = F.linear(F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"],
                 sd[pre + "mlp.fc2.bias"])
    return x + m * sd[pre + "ls2.gamma"]


def interpolate_pos(pos_embed: torch.Tensor, Hp: int, Wp: int, dtype):
    pe = pos_embed.float()
    C = pe.shape[-1]
    M = int(math.sqrt(pe.shape[1] - 1))
    if M * M == Hp * Wp and Hp == Wp:
        return pos_embed
    pp = F.interpolate(pe[:, 1:].reshape(1, M, M, C).permute(0, 3, 1, 2), size=(Hp, Wp), mode="bicubic",
                       antialias=True).permute(0, 2, 3, 1).reshape(1, -1, C)
    return torch.cat((pe[:, :1], pp), dim=1).to(dtype)


def dino_tokens(x: torch.Tensor, sd: Dict[str, torch.Tensor], *, patch_size=14, num_heads=16, dino_depth=24,
                num_register_tokens=4) -> torch.Tensor:
    """The DINOv2-with-registers backbone VGGT uses as `patch_embed` (layers/vision_transformer.py): normalised images
    [N, 3, H, W] -> all tokens after the final LayerNorm [N, 1 + R + Np, C] in the order [cls | registers | patches].
    This stage IS pinned: tests/golden/dinov2_tiny.npz holds the output of transformers' Dinov2WithRegistersModel (an
    independent implementation of the same published model) on the same weights (tools/make_golden_dinov2.py)."""
    p = patch_size
    N, _, H, W = x.shape
    Hp, Wp = H // p, W // p
    t = F.conv2d(x, sd["patch_embed.patch_embed.proj.weight"], sd["patch_embed.patch_embed.proj.bias"], stride=p)
    t = t.flatten(2).transpose(1, 2)                                    # [N, Np, C]
    C = t.shape[-1]
    t = torch.cat((sd["patch_embed.cls_token"].expand(N, -1, -1), t), dim=1)
    t = t + interpolate_pos(sd["patch_embed.pos_embed"], Hp, Wp, x.dtype)
    t = torch.cat((t[:, :1], sd["patch_embed.register_tokens"].expand(N, -1, -1), t[:, 1:]), dim=1)
    for i in range(dino_depth):
        t = block(t, sd, f"patch_embed.blocks.{i}.", num_heads, None, qk_norm=False, eps=1e-6, freq=0.0)
    return F.layer_norm(t, (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], 1e-6)


def aggregator(images: torch.Tensor, sd: Dict[str, torch.Tensor], *, patch_size=14, num_heads=16, depth=24,
               dino_depth=24, num_register_tokens=4, rope_freq=100.0, dtype=torch.bfloat16) -> List[torch.Tensor]:
    """images [B,S,3,H,W] in [0,1]; returns the list of per-iteration tokens [B,S,P,2C] (all iterations)."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    B, S, _, H, W = images.shape
    p = patch_size
    Hp, Wp = H // p, W // p
    x = images.to(dtype)
    mean = torch.tensor(MEAN).view(1, 1, 3, 1, 1).to(dtype)
    std = torch.tensor(STD).view(1, 1, 3, 1, 1).to(dtype)
    x = ((x - mean) / std).reshape(B * S, 3, H, W)
    # DINOv2 ViT with registers
    t = dino_tokens(x, sd, patch_size=p, num_heads=num_heads, dino_depth=dino_depth, num_register_tokens=num_register_tokens)
    C = t.shape[-1]
    patch_tokens = t[:, 1 + num_register_tokens:]
    # special tokens: slot 0 for the first frame, slot 1 for the others
    def expand(tok):  # [1, 2, X, C] -> [B*S, X, C]
        first = tok[:, 0:1].expand(B, 1, *tok.shape[2:])
        rest = tok[:, 1:].expand(B, S - 1, *tok.shape[2:])
        return torch.cat([first, rest], dim=1).reshape(B * S, *tok.shape[2:])
    tokens = torch.cat([expand(sd["camera_token"]), expand(sd["register_token"]), patch_tokens], dim=1)
    ps = 1 + num_register_tokens
    P = tokens.shape[1]
    yy, xx = torch.meshgrid(torch.arange(Hp), torch.arange(Wp), indexing="ij")
    pos = torch.stack([yy.reshape(-1), xx.reshape(-1)], dim=-1) + 1
    pos = torch.cat([torch.zeros(ps, 2, dtype=pos.dtype), pos], dim=0)  # [P, 2]
    pos_f = pos[None].expand(B * S, -1, -1)
    pos_g = pos[None, None].expand(B, S, -1, -1).reshape(B, S * P, 2)
    out = []
    for i in range(depth):
        tokens = block(tokens.reshape(B * S, P, C), sd, f"frame_blocks.{i}.", num_heads, pos_f, qk_norm=True, eps=1e-5,
                       freq=rope_freq)
        fr = tokens.reshape(B, S, P, C)
        tokens = block(tokens.reshape(B, S * P, C), sd, f"global_blocks.{i}.", num_heads, pos_g, qk_norm=True, eps=1e-5,
                       freq=rope_freq)
        out.append(torch.cat([fr, tokens.reshape(B, S, P, C)], dim=-1))
    return out
