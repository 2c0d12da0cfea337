"""Perceiver projector on the HIP kernels. Same constructor, parameter names/shapes and forward contract as the
reference's src/models/projector_perceiver.py:20-82 (PerceiverConfig, PerceiverLayer, PerceiverProjector), so the
reference's checkpoints and its name-based optimiser grouping (train_sft.py:139-145) keep working.

Forward only: the reference runs the projector under @torch.no_grad() (vggt_qwen3_vlm.py:128,162), so no gradient
ever reaches it. Its four nn.Dropout sites per layer (projector_perceiver.py:33,37,42,46-49: attention weights, attention
output, after GELU, MLP output) are nevertheless ACTIVE whenever the module is in train mode - no_grad does not switch
dropout off - so they are applied here under the same condition (`self.training and cfg.dropout > 0`), with a
counter-based mask (statistically, not bitwise, torch's). `.eval()` (the inference scripts) disables them. Parameters stay fp32 like the reference's; bf16 compute copies of the matrices feed the MFMA GEMMs,
while the residual / LayerNorm stream stays fp32."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops
from .ops import BF16, F32, round_up


@dataclass
class PerceiverConfig:
    latent_dim: int = 4096
    num_latents: int = 128
    num_heads: int = 8
    num_layers: int = 6
    ffn_dim: int = 16384
    dropout: float = 0.1


def _xavier_(w: torch.Tensor) -> None:
    nn.init.xavier_uniform_(w)


class _Lin(nn.Module):
    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout))
        _xavier_(self.weight)


class _MHA(nn.Module):
    """Parameter layout of nn.MultiheadAttention: packed in_proj [3D, D] (rows q|k|v), out_proj Linear."""

    def __init__(self, dim: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = _Lin(dim, dim)
        _xavier_(self.in_proj_weight)


class _LN(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Placeholder(nn.Module):
    """Keeps the reference's nn.Sequential indices (0: Linear, 1: GELU, 2: Dropout, 3: Linear)."""


class PerceiverLayer(nn.Module):
    def __init__(self, dim: int, heads: int, ffn_dim: int, dropout: float) -> None:
        super().__init__()
        self.self_attn = _MHA(dim)
        self.mlp = nn.ModuleList([_Lin(dim, ffn_dim), _Placeholder(), _Placeholder(), _Lin(ffn_dim, dim)])
        self.norm1 = _LN(dim)
        self.norm2 = _LN(dim)


class PerceiverProjector(nn.Module):
    """Resample VGGT aggregated tokens to fixed-length latents (projector_perceiver.py:53-82)."""

    def __init__(self, config: PerceiverConfig, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.cfg = config
        self.in_dim, self.out_dim = in_dim, out_dim
        D = config.latent_dim
        if D % config.num_heads or (D // config.num_heads) % 64 or D % 64 or config.ffn_dim % 64:
            raise ops._lib.Vq3Error("Perceiver HIP path: latent_dim/ffn_dim must be multiples of 64 and head_dim % 64 == 0")
        self.latents = nn.Parameter(torch.randn(config.num_latents, D) * 0.02)
        self.in_proj = _Lin(in_dim, D)
        self.layers = nn.ModuleList([PerceiverLayer(D, config.num_heads, config.ffn_dim, config.dropout)
                                     for _ in range(config.num_layers)])
        self.out_proj = _Lin(D, out_dim)
        self._cc = None  # bf16 compute copies
        self._drop_seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        self._drop_offset = 0

    # ------------------------------------------------------------------
    def refresh_compute_copies(self) -> None:
        """bf16 copies of the GEMM weights (K padded to 64) + fp32 biases on the parameters' device."""
        def w16(w):
            n, k = w.shape
            kp = round_up(k, 64)
            out = torch.zeros((n, kp), device=w.device, dtype=BF16)
            out[:, :k] = ops.cast(w.detach().contiguous(), BF16)
            return out
        cc = {"in": w16(self.in_proj.weight), "out": w16(self.out_proj.weight), "layers": []}
        for l in self.layers:
            D = self.cfg.latent_dim
            W = l.self_attn.in_proj_weight.detach()
            cc["layers"].append({"q": w16(W[:D].contiguous()), "kv": w16(W[D:].contiguous()),
                                 "o": w16(l.self_attn.out_proj.weight), "f1": w16(l.mlp[0].weight),
                                 "f2": w16(l.mlp[3].weight)})
        self._cc = cc

    def _apply(self, fn, *a, **k):
        self._cc = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._cc = None
        return super().load_state_dict(*a, **k)

    # ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [B, T, in_dim] (any float dtype) -> [B, num_latents, out_dim] fp32."""
        if self._cc is None:
            self.refresh_compute_copies()
        cfg, cc = self.cfg, self._cc
        B, T, Cin = tokens.shape
        D, N, Hh = cfg.latent_dim, cfg.num_latents, cfg.num_heads
        hd = D // Hh
        Tp = round_up(T, 64)
        dev = tokens.device
        pdrop = float(cfg.dropout) if self.training else 0.0

        def drop(t):
            if pdrop > 0.0:
                ops.dropout_(t, pdrop, self._drop_seed, self._drop_offset)
                self._drop_offset += t.numel()
            return t
        kin = cc["in"].shape[1]
        x = torch.zeros((B * T, kin), device=dev, dtype=BF16)
        x[:, :Cin] = tokens.reshape(B * T, Cin).to(BF16)
        ctx = ops.linear(x, cc["in"], bias=self.in_proj.bias)                       # [B*T, D] bf16
        lat32 = self.latents.detach().to(F32).unsqueeze(0).expand(B, N, D).reshape(B * N, D).contiguous()
        lat16 = ops.cast(lat32, BF16)
        for li, l in enumerate(self.layers):
            w = cc["layers"][li]
            bq, bkv = l.self_attn.in_proj_bias[:D], l.self_attn.in_proj_bias[D:]
            q = ops.linear(lat16, w["q"], bias=bq.contiguous())                      # [B*N, D]
            kv = ops.linear(ctx, w["kv"], bias=bkv.contiguous())                     # [B*T, 2D] = k | v
            S = torch.empty((B * Hh, N, Tp), device=dev, dtype=F32)
            ops.gemm_raw(q, kv, S, N, T, hd, D, 2 * D, Tp, nb1=B, nb2=Hh, sA=(N * D, hd), sB=(T * 2 * D, hd),
                         sC=(Hh * N * Tp, N * Tp), alpha=hd ** -0.5)
            P = drop(ops.softmax_fwd(S, None, 1, T, Tp, False))                      # MHA's attention-weight dropout
            o = torch.empty((B * N, D), device=dev, dtype=BF16)
            if T % 8 == 0:
                # O[b,h] = P[b,h] . V[b,h]: V (columns D + h*hd .. of kv) is read in place as the k-major B operand
                ops.gemm_raw(P, kv, o, N, hd, T, Tp, 2 * D, D, nb1=B, nb2=Hh, sA=(Hh * N * Tp, N * Tp),
                             sB=(T * 2 * D, hd), sC=(N * D, hd), b_off=D, transB=True)
            else:
                Vt = torch.empty((B, Hh, hd, Tp), device=dev, dtype=BF16)
                ops.transpose_raw(kv, Vt, T, hd, Tp, 2 * D, Tp, n=(1, B, Hh), s=(0, T * 2 * D, hd),
                                  d=(0, Hh * hd * Tp, hd * Tp), src_off=D)
                ops.gemm_raw(P, Vt, o, N, hd, Tp, Tp, Tp, D, nb1=B, nb2=Hh, sA=(Hh * N * Tp, N * Tp),
                             sB=(Hh * hd * Tp, hd * Tp), sC=(N * D, hd))
            if pdrop > 0.0:
                # x = LN1(x + drop(attn)) ; x = LN2(x + drop(W2 drop(gelu(W1 x)))): the residual add moves out of the GEMM
                # epilogue into the LayerNorm kernel so that the dropout sits between them
                a = drop(ops.linear(o, w["o"], bias=l.self_attn.out_proj.bias, out_dtype=F32))
                lat16, lat32 = ops.layernorm_fwd(a, l.norm1.weight, l.norm1.bias, 1e-5, res=lat32, want_bf16=True, want_f32=True)
                h = drop(ops.linear(lat16, w["f1"], bias=l.mlp[0].bias, act=ops.ACT_GELU))
                mo = drop(ops.linear(h, w["f2"], bias=l.mlp[3].bias, out_dtype=F32))
                lat16, lat32 = ops.layernorm_fwd(mo, l.norm2.weight, l.norm2.bias, 1e-5, res=lat32, want_bf16=True, want_f32=True)
                continue
            x1 = ops.linear(o, w["o"], bias=l.self_attn.out_proj.bias, residual=lat32, out_dtype=F32)
            lat16, lat32 = ops.layernorm_fwd(x1, l.norm1.weight, l.norm1.bias, 1e-5, want_bf16=True, want_f32=True)
            h = ops.linear(lat16, w["f1"], bias=l.mlp[0].bias, act=ops.ACT_GELU)
            x2 = ops.linear(h, w["f2"], bias=l.mlp[3].bias, residual=lat32, out_dtype=F32)
            lat16, lat32 = ops.layernorm_fwd(x2, l.norm2.weight, l.norm2.bias, 1e-5, want_bf16=True, want_f32=True)
        out = ops.linear(lat16, cc["out"], bias=self.out_proj.bias, out_dtype=F32)
        return out.view(B, N, self.out_dim)
