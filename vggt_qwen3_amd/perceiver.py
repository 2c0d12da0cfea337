"""Perceiver projector on the HIP kernels. Same constructor, parameter names/shapes and forward contract as the
reference's src/models/projector_perceiver.py:20-82 (PerceiverConfig, PerceiverLayer, PerceiverProjector), so the
reference's checkpoints and its name-based optimiser grouping (train_sft.py:139-145) keep working.

`forward` has no gradient: the reference runs the projector under @torch.no_grad() (vggt_qwen3_vlm.py:128,162), so no
gradient ever reaches it although train_sft.py:138-145 gives it its own learning-rate group. `forward_train` + `backward` are the
"corrected" mode SURVEY.md 3.1 asks to keep behind a flag (VisionLanguageConfig.train_projector, default off = reference-faithful):
the same forward saving what autograd would, and a hand-written backward (LayerNorm, GELU, softmax / cross-attention, every GEMM)
that leaves fp32 gradients in `.grad` of the fp32 parameters. The projector's four nn.Dropout sites per layer
(projector_perceiver.py:33,37,42,46-49: attention weights, attention output, after GELU, MLP output) are ACTIVE whenever the module
is in train mode - no_grad does not switch dropout off - so they are applied here under the same condition (`self.training and cfg.dropout > 0`), with a
counter-based mask (statistically, not bitwise, torch's). `.eval()` (the inference scripts) disables them. Parameters stay fp32 like the reference's; bf16 compute copies of the matrices feed the MFMA GEMMs,
while the residual / LayerNorm stream stays fp32."""
from __future__ import annotations

import os
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


def _fused_xattn(hd: int) -> bool:
    """One-launch cross-attention (csrc/perceiver_attn.hip) for the head sizes it is instantiated for; VQ3_PERCEIVER_FUSED=0 keeps the
    batched-GEMM / softmax / batched-GEMM route (the A/B switch of tests/test_kernels_gpu.py::test_perceiver_xattn_*)."""
    return hd in (64, 128, 256, 512) and os.environ.get("VQ3_PERCEIVER_FUSED", "1") != "0"


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
    def _xattn_three_launches(self, q, kv, B, T, Tp, drop):
        """The cross-attention as batched GEMM -> softmax (+ dropout) -> batched GEMM with the f32 scores in HBM: what the fused
        kernel replaces (kept as its A/B partner, VQ3_PERCEIVER_FUSED=0, and for head sizes it is not instantiated for)."""
        cfg = self.cfg
        D, N, Hh = cfg.latent_dim, cfg.num_latents, cfg.num_heads
        hd, dev = D // Hh, q.device
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
        return o

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
        fused = _fused_xattn(hd)
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
            if fused:
                # QK^T, softmax, MHA's attention-weight dropout and PV in one launch per layer (csrc/perceiver_attn.hip); the mask
                # is the one dropout_ would apply to the [B*H, N, Tp] softmax tensor, and the offset advances as if it had
                o = ops.perceiver_xattn(q, kv, B, Hh, N, T, hd, Tp, pdrop, self._drop_seed, self._drop_offset)
                if pdrop > 0.0:
                    self._drop_offset += B * Hh * N * Tp
            else:
                o = self._xattn_three_launches(q, kv, B, T, Tp, drop)
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


# =====================================================================================================================
# "corrected" mode (VisionLanguageConfig.train_projector): forward that saves for backward + hand-written backward.
# Arithmetic of forward() (same kernels, same rounding points); gradients as torch autograd defines them for
# projector_perceiver.py:30-82 (nn.MultiheadAttention math path, post-norm LayerNorms, exact GELU, dropout masks re-applied).
# =====================================================================================================================
def _ones_rows(M: int, dev) -> torch.Tensor:
    """[8, M] bf16 with row 0 = 1: a GEMM against it sums a bf16 matrix over its rows (bias gradients) in f32."""
    o = torch.zeros((8, M), device=dev, dtype=BF16)
    o[0] = 1.0
    return o


def _colsum_bf16_into(dY: torch.Tensor, ones: torch.Tensor, out: torch.Tensor) -> None:
    """out[n] += sum_m dY[m, n] (dY bf16 [M, N], out f32 [N])."""
    M, N = dY.shape
    tmp = torch.empty((8, N), device=dY.device, dtype=F32)
    ops.gemm_raw(ones, dY, tmp, 8, N, M, M, N, N, transB=True)
    ops.colsum_f32(tmp, out=out, accumulate=True)


def _wgrad_into(dY: torch.Tensor, X: torch.Tensor, gW: torch.Tensor, row_off: int = 0) -> None:
    """gW[row_off : row_off + Nout, :K] += dY^T . X   (dY bf16 [M, Nout], X bf16 [M, K], gW f32 [*, K]); both operands as stored."""
    M, Nout = dY.shape
    K = X.shape[1]
    ops.gemm_raw(dY, X, gW, Nout, K, M, Nout, K, gW.shape[1], c_off=row_off * gW.shape[1], transA=True, transB=True, accumulate=True)


def _grad(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, dtype=F32)
    return p.grad


def perceiver_forward_train(self: "PerceiverProjector", tokens: torch.Tensor):
    """tokens [B, T, in_dim] -> (out f32 [B, N, out_dim], ctx). Dropout sites active in train mode (masks regenerated in backward)."""
    if self._cc is None:
        self.refresh_compute_copies()
    cfg, cc = self.cfg, self._cc
    B, T, Cin = tokens.shape
    D, N, Hh = cfg.latent_dim, cfg.num_latents, cfg.num_heads
    hd = D // Hh
    if T % 8 or N % 8:
        raise ops._lib.Vq3Error("Perceiver backward: context length and num_latents must be multiples of 8")
    Tp = round_up(T, 64)
    dev = tokens.device
    pdrop = float(cfg.dropout) if self.training else 0.0

    def drop(t):
        off = self._drop_offset
        if pdrop > 0.0:
            ops.dropout_(t, pdrop, self._drop_seed, off)
            self._drop_offset += t.numel()
        return off
    kin = cc["in"].shape[1]
    x = torch.zeros((B * T, kin), device=dev, dtype=BF16)
    x[:, :Cin] = tokens.reshape(B * T, Cin).to(BF16)
    ctxt = ops.linear(x, cc["in"], bias=self.in_proj.bias)
    lat32 = self.latents.detach().to(F32).unsqueeze(0).expand(B, N, D).reshape(B * N, D).contiguous()
    lat16 = ops.cast(lat32, BF16)
    saved = []
    for li, l in enumerate(self.layers):
        w = cc["layers"][li]
        sv = {"lat32_in": lat32, "lat16_in": lat16}
        q = ops.linear(lat16, w["q"], bias=l.self_attn.in_proj_bias[:D].contiguous())
        kv = ops.linear(ctxt, w["kv"], bias=l.self_attn.in_proj_bias[D:].contiguous())
        if _fused_xattn(hd):
            sv["off_P"] = self._drop_offset
            o, sv["P"], P = ops.perceiver_xattn(q, kv, B, Hh, N, T, hd, Tp, pdrop, self._drop_seed, self._drop_offset, keep_p=True)
            if pdrop > 0.0:
                self._drop_offset += B * Hh * N * Tp
        else:
            S = torch.empty((B * Hh, N, Tp), device=dev, dtype=F32)
            ops.gemm_raw(q, kv, S, N, T, hd, D, 2 * D, Tp, nb1=B, nb2=Hh, sA=(N * D, hd), sB=(T * 2 * D, hd),
                         sC=(Hh * N * Tp, N * Tp), alpha=hd ** -0.5)
            P = ops.softmax_fwd(S, None, 1, T, Tp, False)
            sv["P"] = P.clone() if pdrop > 0.0 else P
            sv["off_P"] = drop(P)
            o = torch.empty((B * N, D), device=dev, dtype=BF16)
            ops.gemm_raw(P, kv, o, N, hd, T, Tp, 2 * D, D, nb1=B, nb2=Hh, sA=(Hh * N * Tp, N * Tp), sB=(T * 2 * D, hd),
                         sC=(N * D, hd), b_off=D, transB=True)
        a = ops.linear(o, w["o"], bias=l.self_attn.out_proj.bias, out_dtype=F32)
        sv["off_a"] = drop(a)
        lat16_1, lat32_1 = ops.layernorm_fwd(a, l.norm1.weight, l.norm1.bias, 1e-5, res=lat32, want_bf16=True, want_f32=True)
        z = ops.linear(lat16_1, w["f1"], bias=l.mlp[0].bias)
        h = ops.gelu_fwd(z)
        sv["off_h"] = drop(h)
        mo = ops.linear(h, w["f2"], bias=l.mlp[3].bias, out_dtype=F32)
        sv["off_mo"] = drop(mo)
        lat16, lat32 = ops.layernorm_fwd(mo, l.norm2.weight, l.norm2.bias, 1e-5, res=lat32_1, want_bf16=True, want_f32=True)
        sv.update(q=q, kv=kv, Pd=P, o=o, a=a, lat16_1=lat16_1, lat32_1=lat32_1, z=z, h=h, mo=mo)
        saved.append(sv)
    out = ops.linear(lat16, cc["out"], bias=self.out_proj.bias, out_dtype=F32)
    ctx = dict(saved=saved, x=x, ctxt=ctxt, lat16_out=lat16, B=B, T=T, Tp=Tp, pdrop=pdrop, seed=self._drop_seed)
    return out.view(B, N, self.out_dim), ctx


def perceiver_backward(self: "PerceiverProjector", ctx: dict, d_out: torch.Tensor) -> None:
    """d_out f32 [B, N, out_dim] = d(loss)/d(forward_train's output). Accumulates fp32 gradients into `.grad` of every parameter
    (latents, in_proj, the six layers, out_proj)."""
    cfg, cc = self.cfg, self._cc
    B, T, Tp, pdrop, seed = ctx["B"], ctx["T"], ctx["Tp"], ctx["pdrop"], ctx["seed"]
    D, N, Hh = cfg.latent_dim, cfg.num_latents, cfg.num_heads
    hd = D // Hh
    dev = d_out.device
    M = B * N
    ones_m, ones_t = _ones_rows(M, dev), _ones_rows(B * T, dev)

    def undrop(t, off):      # gradient of inverted dropout = the same mask and scale
        if pdrop > 0.0:
            ops.dropout_(t, pdrop, seed, off)
        return t
    d_out2 = d_out.reshape(M, self.out_dim).contiguous().to(F32)
    dy16 = ops.cast(d_out2, BF16)
    _wgrad_into(dy16, ctx["lat16_out"], _grad(self.out_proj.weight))
    ops.colsum_f32(d_out2, out=_grad(self.out_proj.bias), accumulate=True)
    d_lat = torch.empty((M, D), device=dev, dtype=F32)                      # d(loss)/d(lat32 after the last layer)
    ops.gemm_raw(dy16, cc["out"], d_lat, M, D, self.out_dim, self.out_dim, cc["out"].shape[1], D, transB=True)
    d_ctx = torch.zeros((B * T, D), device=dev, dtype=F32)
    for li in reversed(range(len(self.layers))):
        l, w, sv = self.layers[li], cc["layers"][li], ctx["saved"][li]
        # ---- LayerNorm 2 over (lat32_1 + mo)
        dx2, _, _ = ops.layernorm_bwd(d_lat, sv["mo"], l.norm2.weight.detach(), 1e-5, res=sv["lat32_1"],
                                      dw_out=_grad(l.norm2.weight), db_out=_grad(l.norm2.bias))
        d_mo = undrop(dx2.clone(), sv["off_mo"])                             # dx2 itself is the residual branch's gradient
        d_mo16 = ops.cast(d_mo, BF16)
        _wgrad_into(d_mo16, sv["h"], _grad(l.mlp[3].weight))
        ops.colsum_f32(d_mo, out=_grad(l.mlp[3].bias), accumulate=True)
        F_ = cfg.ffn_dim
        d_h = torch.empty((M, F_), device=dev, dtype=BF16)
        ops.gemm_raw(d_mo16, w["f2"], d_h, M, F_, D, D, F_, F_, transB=True)
        undrop(d_h, sv["off_h"])
        dz = ops.gelu_bwd(d_h, sv["z"])
        _wgrad_into(dz, sv["lat16_1"], _grad(l.mlp[0].weight))
        _colsum_bf16_into(dz, ones_m, _grad(l.mlp[0].bias))
        ops.gemm_raw(dz, w["f1"], dx2, M, D, F_, F_, D, D, transB=True, accumulate=True)     # dx2 += dz . W1  -> d(lat32_1)
        # ---- LayerNorm 1 over (lat32_in + a)
        dx1, _, _ = ops.layernorm_bwd(dx2, sv["a"], l.norm1.weight.detach(), 1e-5, res=sv["lat32_in"],
                                      dw_out=_grad(l.norm1.weight), db_out=_grad(l.norm1.bias))
        d_a = undrop(dx1.clone(), sv["off_a"])
        d_a16 = ops.cast(d_a, BF16)
        _wgrad_into(d_a16, sv["o"], _grad(l.self_attn.out_proj.weight))
        ops.colsum_f32(d_a, out=_grad(l.self_attn.out_proj.bias), accumulate=True)
        d_o = torch.empty((M, D), device=dev, dtype=BF16)
        ops.gemm_raw(d_a16, w["o"], d_o, M, D, D, D, D, D, transB=True)
        # ---- cross-attention: o = Pd . V, Pd = dropout(P), P = softmax(alpha q k^T)
        q, kv, P, Pd = sv["q"], sv["kv"], sv["P"], sv["Pd"]
        dP = torch.empty((B * Hh, N, Tp), device=dev, dtype=F32)
        ops.gemm_raw(d_o, kv, dP, N, T, hd, D, 2 * D, Tp, nb1=B, nb2=Hh, sA=(N * D, hd), sB=(T * 2 * D, hd),
                     sC=(Hh * N * Tp, N * Tp), b_off=D)
        d_kv = torch.empty((B * T, 2 * D), device=dev, dtype=BF16)
        ops.gemm_raw(Pd, d_o, d_kv, T, hd, N, Tp, D, 2 * D, nb1=B, nb2=Hh, sA=(Hh * N * Tp, N * Tp), sB=(N * D, hd),
                     sC=(T * 2 * D, hd), c_off=D, transA=True, transB=True)                       # dV
        undrop(dP, sv["off_P"])
        dS = ops.softmax_bwd(P, dP, T, hd ** -0.5)
        dq = torch.empty((M, D), device=dev, dtype=BF16)
        ops.gemm_raw(dS, kv, dq, N, hd, T, Tp, 2 * D, D, nb1=B, nb2=Hh, sA=(Hh * N * Tp, N * Tp), sB=(T * 2 * D, hd),
                     sC=(N * D, hd), transB=True)
        ops.gemm_raw(dS, q, d_kv, T, hd, N, Tp, D, 2 * D, nb1=B, nb2=Hh, sA=(Hh * N * Tp, N * Tp), sB=(N * D, hd),
                     sC=(T * 2 * D, hd), transA=True, transB=True)                                # dK
        gin_w, gin_b = _grad(l.self_attn.in_proj_weight), _grad(l.self_attn.in_proj_bias)
        _wgrad_into(dq, sv["lat16_in"], gin_w, 0)
        _wgrad_into(d_kv, ctx["ctxt"], gin_w, D)
        _colsum_bf16_into(dq, ones_m, gin_b[:D])
        _colsum_bf16_into(d_kv, ones_t, gin_b[D:])
        ops.gemm_raw(dq, w["q"], dx1, M, D, D, D, D, D, transB=True, accumulate=True)             # dx1 += dq . Wq -> d(lat32_in)
        ops.gemm_raw(d_kv, w["kv"], d_ctx, B * T, D, 2 * D, 2 * D, D, D, transB=True, accumulate=True)
        d_lat = dx1
    # latents: the same [N, D] parameter expanded over the batch
    ops.colsum_f32(d_lat.view(B, N * D), out=_grad(self.latents).view(-1), accumulate=True)
    d_ctx16 = ops.cast(d_ctx, BF16)
    gw_in = _grad(self.in_proj.weight)
    Cin = self.in_dim
    if ctx["x"].shape[1] == Cin:
        _wgrad_into(d_ctx16, ctx["x"], gw_in)
    else:                                            # K padded to 64 for the forward GEMM: gradient of the padded matrix, then cut
        tmp = torch.zeros((D, ctx["x"].shape[1]), device=dev, dtype=F32)
        _wgrad_into(d_ctx16, ctx["x"], tmp)
        gw_in.add_(tmp[:, :Cin])
    ops.colsum_f32(d_ctx, out=_grad(self.in_proj.bias), accumulate=True)


PerceiverProjector.forward_train = perceiver_forward_train
PerceiverProjector.backward = perceiver_backward
