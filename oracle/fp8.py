"""TEST INFRASTRUCTURE - CPU statement of the fp8 contract (forward projections and, since round 4, their input-gradient GEMMs) of BASELINE config C5 ("Qwen3 linear
weights in OCP fp8-e4m3 + per-output-channel fp32 scales; activations bf16 -> fp8 on the fly; fp32 accumulate").
The reference contains no fp8 code, so there is nothing of its own to pin against: **parity unpinned** for this
config; the fixed points are the OCP e4m3 encoding itself (torch.float8_e4m3fn, round-to-nearest-even) and the
unquantised product, against which the quantisation error is bounded in the tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import torch

E4M3_MAX = 448.0


def quant_rows(x: torch.Tensor):
    """Per-row symmetric quantisation: scale = amax / 448 (1 for a zero row); q = e4m3(x * (448 / amax)), both in fp32."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    nz = amax > 0
    inv = torch.where(nz, torch.tensor(E4M3_MAX) / amax, torch.ones_like(amax))
    scale = torch.where(nz, amax / torch.tensor(E4M3_MAX), torch.ones_like(amax))
    q = (xf * inv[:, None]).to(torch.float8_e4m3fn)
    return q, scale


def linear(x: torch.Tensor, w: torch.Tensor, residual: torch.Tensor = None) -> torch.Tensor:
    """bf16( sx[m] * sw[n] * sum_k q_x[m,k] q_w[n,k] ) (+ residual, rounded again) - fp32 accumulation."""
    xq, xs = quant_rows(x)
    wq, ws = quant_rows(w)
    acc = xq.float() @ wq.float().t()
    out = (acc * (xs[:, None] * ws[None, :])).to(torch.bfloat16)
    if residual is not None:
        out = (out.float() + residual.float()).to(torch.bfloat16)
    return out


def quant_rows_scaled(x: torch.Tensor, colmul: torch.Tensor):
    """quant_rows of x * colmul[None, :] (fp32 product of the bf16 value and the fp32 multiplier): how dY enters the dgrad GEMMs."""
    return quant_rows(x.float() * colmul.float()[None, :])


def dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dX = dY . W under the C5 contract (round 4): W [N, K] is the SAME e4m3 tensor the forward multiplies by (one fp32 scale per
    output channel n); those scales run along this product's contraction, so they are folded into dY before its rows are quantised:
    dX[m, k] = bf16( t[m] * sum_n q(dY[m, n] s[n]) q_w[n, k] ), fp32 accumulation."""
    wq, ws = quant_rows(w)
    dq, t = quant_rows_scaled(dy, ws)
    return ((dq.float() @ wq.float()) * t[:, None]).to(torch.bfloat16)


class fp8_projections:
    """Context manager: inside it, the 2-D weight products of oracle.qwen3's attention / MLP (the q, k, v, o, gate, up,
    down projections) follow the fp8 contract above; lm_head (called outside the decoder layers) stays bf16, like the
    product. Per-output-channel weight scales make the fused q|k|v and gate|up GEMMs of the product identical to the
    separate projections here."""

    def __enter__(self):
        from . import qwen3 as oq
        self._oq = oq
        self._orig = (oq.attention, oq.mlp)
        import torch.nn.functional as F
        real_linear = F.linear

        def fp8_linear(x, w, b=None):
            assert b is None
            shp = x.shape
            y = linear(x.reshape(-1, shp[-1]).to(torch.bfloat16), w)
            return y.reshape(*shp[:-1], w.shape[0]).to(x.dtype)

        class _F:
            def __getattr__(self, name):
                return fp8_linear if name == "linear" else getattr(F, name)

        self._saved_F = oq.F
        self._proxy = _F()
        orig_attention, orig_mlp = self._orig

        def attention(*a, **k):
            oq.F = self._proxy
            try:
                return orig_attention(*a, **k)
            finally:
                oq.F = self._saved_F

        def mlp(*a, **k):
            oq.F = self._proxy
            try:
                return orig_mlp(*a, **k)
            finally:
                oq.F = self._saved_F

        oq.attention, oq.mlp = attention, mlp
        return self

    def __exit__(self, *exc):
        self._oq.attention, self._oq.mlp = self._orig
        self._oq.F = self._saved_F
        return False
