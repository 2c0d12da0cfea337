"""Tensor-level wrappers over the C ABI (include/vq3_hip.h): they take torch CUDA tensors, check shapes/dtypes on the
host, and enqueue the HIP kernel on torch's current stream. No math happens in Python or in torch here."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from ._lib import GemmDesc, check

BF16 = torch.bfloat16
F32 = torch.float32

ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2

# When set to a list, every GEMM launch is bracketed by HIP events on the launch stream and
# (algorithmic FLOPs, algorithmic bytes, start, end) is appended: bench.py's live roofline measurement.
GEMM_PROFILE = None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise _lib.Vq3Error(f"{name}: expected a CUDA (HIP) tensor; the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise _lib.Vq3Error(f"{name}: expected dtype {dtype}, got {t.dtype}")


def _out2d(out: Optional[torch.Tensor], rows: int, cols: int, like: torch.Tensor, name: str) -> torch.Tensor:
    """The op's bf16 [rows, cols] result: a fresh tensor, or the caller's (a row block of a longer-lived slab, e.g. the deferred
    weight-gradient operands of Qwen3ForCausalLM) when it has exactly that shape and dense rows."""
    if out is None:
        return torch.empty((rows, cols), device=like.device, dtype=BF16)
    if out.dtype != BF16 or tuple(out.shape) != (rows, cols) or not out.is_contiguous() or out.device != like.device:
        raise _lib.Vq3Error(f"{name}: out must be a contiguous bf16 [{rows}, {cols}] tensor on {like.device}")
    return out


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------------------------- GEMM
_TUNE_WS = None      # None: not set up yet; else {"tune": {device: tensor}, "split": [tensors]} - memory the library asked the provider for
_WS_PROVIDER = None  # the ctypes callback object (must outlive its registration)


def _ws_provider(nbytes: int, device: int, kind: int) -> int:
    """vq3_ws_provider_t: called by the library, from inside a GEMM call and never under graph capture, the first time it needs device
    memory - kind 0: the kernel-choice tuner is about to MEASURE a shape on `device` (trial output + 320 MiB cache flush; at least
    VQ3_GEMM_TUNE_WS_MB, default 1024, so that later shapes fit without another call); kind 1: a split-K launch's first use of a stream
    (48 MiB of partial tiles, kept for the life of the process). Returns 0 ("none": the shape is not measured / the launch runs unsplit)
    rather than raising through the C frame."""
    import os
    try:
        dev = torch.device("cuda", int(device))
        if kind == 0:
            mb = int(os.environ.get("VQ3_GEMM_TUNE_WS_MB", "1024"))
            if mb <= 0:
                return 0
            t = torch.empty(max(int(nbytes), mb << 20), dtype=torch.uint8, device=dev)
            _TUNE_WS["tune"][int(device)] = t          # (a smaller block of an earlier call is released here)
        else:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            _TUNE_WS["split"].append(t)
        return t.data_ptr()
    except Exception:      # out of memory, no such device: the library falls back (and says so)
        return 0


def gemm_tune_setup(force: bool = False) -> None:
    """Once per process, before the first GEMM: register the workspace PROVIDER (nothing is allocated until the tuner actually measures a
    shape or a split-K launch first runs on a stream: an inference-only process whose shapes are all in the shipped table, or a rank of a
    multi-rank job, never holds the 1 GiB tuning workspace) and, in a multi-rank job, stop the tuner from measuring at all (the shipped
    table, then the heuristic, answer - the same function on every rank; VQ3_GEMM_AUTOTUNE_DIST=1 lets every rank measure as a single
    process would) and release what it held. Stage1Trainer calls it again once torch.distributed is up (force=True, with its process group)."""
    global _TUNE_WS, _WS_PROVIDER
    if _TUNE_WS is not None and not force:
        return
    import ctypes as C
    import os
    lib = _lib.load()
    if _TUNE_WS is None:
        _TUNE_WS = {"tune": {}, "split": []}
        if torch.cuda.is_available():
            _WS_PROVIDER = C.CFUNCTYPE(C.c_void_p, C.c_int64, C.c_int32, C.c_int32)(_ws_provider)
            check(lib.vq3_gemm_workspace_provider(C.cast(_WS_PROVIDER, C.c_void_p)), "vq3_gemm_workspace_provider")
    hold = _multi_rank(_TUNE_GROUP) and os.environ.get("VQ3_GEMM_AUTOTUNE_DIST", "0") != "1"
    check(lib.vq3_gemm_autotune_hold(1 if hold else 0), "vq3_gemm_autotune_hold")
    if hold and _TUNE_WS["tune"]:
        check(lib.vq3_gemm_tune_workspace(None, 0), "vq3_gemm_tune_workspace")        # nothing will be measured: give the memory back
        _TUNE_WS["tune"].clear()


_TUNE_GROUP = None   # the process group whose size decides "multi-rank" (Stage1Trainer sets its own; None = the default group)


def _multi_rank(group=None) -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def gemm_raw(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, lda: int, ldb: int, ldc: int,
             *, bias=None, colscale=None, R=None, ldr: int = 0, nb1: int = 1, nb2: int = 1, b2divB: int = 1,
             sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), act: int = 0, accumulate: bool = False, alpha: float = 1.0,
             a_off: int = 0, b_off: int = 0, c_off: int = 0, r_off: int = 0, transA: bool = False,
             transB: bool = False, ksplit: int = 1, ln_fold=None) -> None:
    """Raw descriptor launch. A/B bf16; C bf16 or f32 (decides out_f32). Offsets are in elements.
    transA / transB: the operand is stored k-major ([K, ld]) instead of [M or N, ld]. ln_fold: a GemmLnFold (see ln_fold())."""
    _req(A, BF16, "gemm A"); _req(B, BF16, "gemm B")
    if C.dtype not in (BF16, F32):
        raise _lib.Vq3Error(f"gemm C: dtype must be bf16 or f32, got {C.dtype}")
    if R is not None and R.dtype != C.dtype:
        raise _lib.Vq3Error("gemm: residual dtype must equal C dtype")
    if bias is not None: _req(bias, F32, "gemm bias")
    if colscale is not None: _req(colscale, F32, "gemm colscale")
    d = GemmDesc()
    d.A = A.data_ptr() + 2 * a_off
    d.B = B.data_ptr() + 2 * b_off
    d.C = C.data_ptr() + C.element_size() * c_off
    d.bias = _p(bias); d.colscale = _p(colscale)
    d.R = None if R is None else R.data_ptr() + R.element_size() * r_off
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = M, N, K, lda, ldb, ldc, ldr
    d.sA1, d.sA2 = sA; d.sB1, d.sB2 = sB; d.sC1, d.sC2 = sC; d.sR1, d.sR2 = sR
    d.nb1, d.nb2, d.b2divB = nb1, nb2, b2divB
    d.act = act; d.out_f32 = 1 if C.dtype == F32 else 0; d.accumulate = 1 if accumulate else 0
    d.alpha = alpha
    d.transA = 1 if transA else 0; d.transB = 1 if transB else 0; d.ksplit = ksplit
    lib = _lib.load()
    if _TUNE_WS is None:
        gemm_tune_setup()
    launch = (lambda: check(lib.vq3_gemm_bf16_nt(d, _stream()), "vq3_gemm_bf16_nt")) if ln_fold is None else \
             (lambda: check(lib.vq3_gemm_bf16_nt_ln(d, ln_fold, _stream()), "vq3_gemm_bf16_nt_ln"))
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        GEMM_PROFILE.append((2.0 * M * N * K * nb1 * nb2,
                             float(nb1 * nb2) * (2.0 * M * K + 2.0 * N * K / b2divB + C.element_size() * M * N *
                                                 (1 + (R is not None) + bool(accumulate))), e0, e1, (M, N, K, nb1 * nb2)))
        return
    launch()


def ln_fold(stats_in: Optional[torch.Tensor] = None, eps: float = 0.0, colsum: Optional[torch.Tensor] = None,
            stats_out: Optional[torch.Tensor] = None):
    """Descriptor of a LayerNorm folded into the GEMMs either side of it (include/vq3_hip.h: vq3_gemm_ln_fold). stats_in f32
    [M, parts, 2] = (sum, sum of squares) of the rows of the RAW activation the GEMM reads as A, colsum f32 [N] = row sums of the
    gamma-scaled weight; stats_out f32 [M, N / 128, 2]: the GEMM leaves the same pairs for its own output rows. The tensors must
    outlive the launch (the caller keeps them)."""
    f = _lib.GemmLnFold()
    if stats_in is not None:
        _req(stats_in, F32, "ln_fold stats_in"); _req(colsum, F32, "ln_fold colsum")
        assert stats_in.is_contiguous() and stats_in.dim() == 3 and stats_in.shape[2] == 2 and colsum.is_contiguous()
        f.stats_in, f.parts_in, f.eps, f.colsum = stats_in.data_ptr(), stats_in.shape[1], eps, colsum.data_ptr()
    if stats_out is not None:
        _req(stats_out, F32, "ln_fold stats_out"); assert stats_out.is_contiguous()
        f.stats_out = stats_out.data_ptr()
    return f


def rowstats128(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(sum, sum of squares) per row and 128-column group of a bf16 matrix -> f32 [rows, cols / 128, 2] (vq3_rowstats128)."""
    _req(x, BF16, "rowstats128 x"); assert x.dim() == 2 and x.is_contiguous() and x.shape[1] % 128 == 0
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols // 128, 2), device=x.device, dtype=F32)
    assert out.shape == (rows, cols // 128, 2) and out.dtype == F32 and out.is_contiguous()
    check(_lib.load().vq3_rowstats128(x.data_ptr(), out.data_ptr(), rows, cols, _stream()), "vq3_rowstats128")
    return out


def linear(x: torch.Tensor, w: torch.Tensor, *, bias=None, colscale=None, residual=None, act: int = 0,
           out: Optional[torch.Tensor] = None, out_dtype=BF16, accumulate: bool = False,
           alpha: float = 1.0, ln_fold=None) -> torch.Tensor:
    """out[M,N] = epilogue(x[M,K] @ w[N,K]^T). x, w: 2-D bf16 with unit inner stride."""
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N, K2 = w.shape
    if K != K2:
        raise _lib.Vq3Error(f"linear: K mismatch {K} vs {K2}")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=out_dtype)
    assert out.stride(1) == 1 and out.shape == (M, N)
    ldr = 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    gemm_raw(x, w, out, M, N, K, x.stride(0), w.stride(0), out.stride(0), bias=bias, colscale=colscale, R=residual,
             ldr=ldr, act=act, accumulate=accumulate, alpha=alpha, ln_fold=ln_fold)
    return out


# ----------------------------------------------------------------------------------------------- norms
def rmsnorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float, want_rstd: bool = False, out: Optional[torch.Tensor] = None):
    _req(x, BF16, "rmsnorm x"); _req(w, BF16, "rmsnorm w")
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    y = _out2d(out, rows, cols, x, "rmsnorm_fwd")
    rstd = torch.empty((rows,), device=x.device, dtype=F32) if want_rstd else None
    check(_lib.load().vq3_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), _p(rstd), rows, cols, x.stride(0),
                                      y.stride(0), eps, _stream()), "vq3_rmsnorm_fwd")
    return (y, rstd) if want_rstd else y


def colsum_flush(jobs: list) -> None:
    """Run the deferred column sums (part f32 [nrows, cols] -> out bf16 [cols] (+)=) collected in `jobs`, 8 per launch."""
    lib = _lib.load()
    for i in range(0, len(jobs), 8):
        chunk = jobs[i:i + 8]
        arr = (_lib.ColsumJob * len(chunk))()
        for k, (part, nrows, cols, out, acc) in enumerate(chunk):
            arr[k].part, arr[k].out_bf16 = part.data_ptr(), out.data_ptr()
            arr[k].nrows, arr[k].cols, arr[k].accumulate = nrows, cols, 1 if acc else 0
        check(lib.vq3_colsum_multi(arr, len(chunk), _stream()), "vq3_colsum_multi")
    jobs.clear()


def rmsnorm_bwd(dy, x, w, rstd, dres, dw_out: torch.Tensor, accumulate: bool, eps: float = 0.0, defer: Optional[list] = None,
                out: Optional[torch.Tensor] = None):
    """Returns dx (bf16) = [dres +] d/dx; dw_out (bf16 [cols]) (+)= the weight gradient (partial slab + column sum).
    defer: a list - the column sum is appended to it instead of launched (colsum_flush runs a layer's sums in one launch)."""
    _req(dy, BF16, "rmsnorm_bwd dy"); _req(x, BF16, "rmsnorm_bwd x"); _req(dw_out, BF16, "rmsnorm_bwd dw")
    assert dy.is_contiguous() and x.is_contiguous() and (dres is None or dres.is_contiguous())
    rows, cols = x.shape
    dx = _out2d(out, rows, cols, x, "rmsnorm_bwd")
    # rows per workgroup = per partial dw row (vq3_rmsnorm_bwd_rows). Measured at 9 600 x 2560 (a merged pass): 16 rows per workgroup
    # shrink the slab from 24.6 MB to 6.1 MB and its column sum from 29 to 23 us, but the kernel itself goes from 51 to 64 us (a wave
    # then walks 4 rows one memory round trip after the other) - a net loss, so 4 stays.
    rpp = int(os.environ.get("VQ3_RMSNORM_BWD_ROWS", "4"))
    nblk = (rows + rpp - 1) // rpp
    part = torch.empty((nblk, cols), device=x.device, dtype=F32)
    lib = _lib.load()
    check(lib.vq3_rmsnorm_bwd_rows(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), _p(dres), dx.data_ptr(),
                                   part.data_ptr(), rows, cols, rpp, _stream()), "vq3_rmsnorm_bwd_rows")
    if defer is not None:
        defer.append((part, nblk, cols, dw_out, accumulate))
        return dx
    check(lib.vq3_colsum_f32_to_bf16(part.data_ptr(), nblk, cols, dw_out.data_ptr(), 1 if accumulate else 0,
                                     _stream()), "vq3_colsum_f32_to_bf16")
    return dx


def layernorm_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, *, res=None, want_bf16=True,
                  want_f32=False):
    assert x.is_contiguous() and x.dim() == 2
    _req(w, F32, "layernorm w"); _req(b, F32, "layernorm b")
    if x.dtype not in (BF16, F32):
        raise _lib.Vq3Error("layernorm: x must be bf16 or f32")
    if res is not None:
        assert res.dtype == x.dtype and res.is_contiguous() and res.shape == x.shape
    rows, cols = x.shape
    yb = torch.empty((rows, cols), device=x.device, dtype=BF16) if want_bf16 else None
    yf = torch.empty((rows, cols), device=x.device, dtype=F32) if want_f32 else None
    check(_lib.load().vq3_layernorm_fwd(x.data_ptr(), _p(res), 1 if x.dtype == F32 else 0, w.data_ptr(), b.data_ptr(),
                                        _p(yb), _p(yf), rows, cols, eps, _stream()), "vq3_layernorm_fwd")
    return yb, yf


# ----------------------------------------------------------------------------------------------- element-wise
def silu_mul_fwd(gu: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(gu, BF16, "silu_mul gu")
    assert gu.is_contiguous() and gu.dim() == 2 and gu.shape[1] % 2 == 0
    rows, two_i = gu.shape
    act = _out2d(out, rows, two_i // 2, gu, "silu_mul_fwd")
    check(_lib.load().vq3_silu_mul_fwd(gu.data_ptr(), act.data_ptr(), rows, two_i // 2, _stream()), "vq3_silu_mul_fwd")
    return act


def silu_mul_bwd(dact: torch.Tensor, gu: torch.Tensor) -> torch.Tensor:
    _req(dact, BF16, "silu_mul_bwd dact"); _req(gu, BF16, "silu_mul_bwd gu")
    assert dact.is_contiguous() and gu.is_contiguous()
    rows, two_i = gu.shape
    dgu = torch.empty_like(gu)
    check(_lib.load().vq3_silu_mul_bwd(dact.data_ptr(), gu.data_ptr(), dgu.data_ptr(), rows, two_i // 2, _stream()),
          "vq3_silu_mul_bwd")
    return dgu


def transpose_raw(src, dst, R, C, Rpad, lds, ldd, n=(1, 1, 1), s=(0, 0, 0), d=(0, 0, 0), src_off=0, dst_off=0):
    _req(src, BF16, "transpose src"); _req(dst, BF16, "transpose dst")
    check(_lib.load().vq3_transpose_bf16(src.data_ptr() + 2 * src_off, dst.data_ptr() + 2 * dst_off, R, C, Rpad, lds,
                                         ldd, n[0], n[1], n[2], s[0], s[1], s[2], d[0], d[1], d[2], _stream()),
          "vq3_transpose_bf16")


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, eps: float, res: Optional[torch.Tensor] = None,
                  dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """torch.nn.LayerNorm backward over f32 rows: x (+ res) = the tensor that was normalised, dy = d(output).
    -> (dx f32 [rows, cols], dw f32 [cols], db f32 [cols]); dw_out / db_out (f32 [cols]): ACCUMULATE the parameter gradients there."""
    _req(dy, F32, "layernorm_bwd dy"); _req(x, F32, "layernorm_bwd x"); _req(w, F32, "layernorm_bwd w")
    assert dy.is_contiguous() and x.is_contiguous() and dy.shape == x.shape and x.dim() == 2
    if res is not None:
        _req(res, F32, "layernorm_bwd res"); assert res.is_contiguous() and res.shape == x.shape
    rows, cols = x.shape
    dx = torch.empty_like(x)
    nblk = (rows + 15) // 16
    part = torch.empty((2, nblk, cols), device=x.device, dtype=F32)
    lib = _lib.load()
    check(lib.vq3_layernorm_bwd(dy.data_ptr(), x.data_ptr(), _p(res), w.data_ptr(), dx.data_ptr(), part[0].data_ptr(),
                                part[1].data_ptr(), rows, cols, eps, _stream()), "vq3_layernorm_bwd")
    dw = colsum_f32(part[0], out=dw_out, accumulate=dw_out is not None)
    db = colsum_f32(part[1], out=db_out, accumulate=db_out is not None)
    return dx, dw, db


def colsum_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[c] (+)= sum_r x[r, c] for an f32 matrix (bias gradients of f32 activations' gradients)."""
    _req(x, F32, "colsum_f32 x"); assert x.is_contiguous() and x.dim() == 2
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, device=x.device, dtype=F32)
        accumulate = False
    _req(out, F32, "colsum_f32 out")
    check(_lib.load().vq3_colsum_f32(x.data_ptr(), rows, cols, out.data_ptr(), 1 if accumulate else 0, _stream()), "vq3_colsum_f32")
    return out


def gelu_fwd(z: torch.Tensor) -> torch.Tensor:
    _req(z, BF16, "gelu_fwd z"); assert z.is_contiguous() and z.numel() % 8 == 0
    h = torch.empty_like(z)
    check(_lib.load().vq3_gelu_fwd(z.data_ptr(), h.data_ptr(), z.numel(), _stream()), "vq3_gelu_fwd")
    return h


def gelu_bwd(dh: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    _req(dh, BF16, "gelu_bwd dh"); _req(z, BF16, "gelu_bwd z")
    assert dh.is_contiguous() and z.is_contiguous() and dh.shape == z.shape and z.numel() % 8 == 0
    dz = torch.empty_like(z)
    check(_lib.load().vq3_gelu_bwd(dh.data_ptr(), z.data_ptr(), dz.data_ptr(), z.numel(), _stream()), "vq3_gelu_bwd")
    return dz


def transpose2d(x: torch.Tensor, pad_to: int = 1) -> torch.Tensor:
    """[R, C] bf16 -> [C, round_up(R, pad_to)] with zero padding."""
    assert x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    Rp = round_up(R, pad_to)
    out = torch.empty((Cc, Rp), device=x.device, dtype=BF16)
    transpose_raw(x, out, R, Cc, Rp, x.stride(0), Rp)
    return out


def cast(x: torch.Tensor, dtype) -> torch.Tensor:
    assert x.is_contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    if x.dtype == F32 and dtype == BF16:
        d = 0
    elif x.dtype == BF16 and dtype == F32:
        d = 1
    else:
        raise _lib.Vq3Error(f"cast: unsupported {x.dtype} -> {dtype}")
    if x.numel():
        check(_lib.load().vq3_cast(x.data_ptr(), y.data_ptr(), x.numel(), d, _stream()), "vq3_cast")
    return y


def f32_to_bf16_acc(src: torch.Tensor, acc: torch.Tensor, accumulate: bool) -> None:
    _req(src, F32, "f32_to_bf16_acc src"); _req(acc, BF16, "f32_to_bf16_acc acc")
    assert src.is_contiguous() and acc.is_contiguous() and src.numel() == acc.numel()
    check(_lib.load().vq3_f32_to_bf16_acc(src.data_ptr(), acc.data_ptr(), src.numel(), 1 if accumulate else 0,
                                          _stream()), "vq3_f32_to_bf16_acc")


def gather_rows(src: torch.Tensor, idx: torch.Tensor, n: int, n_pad: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(src, BF16, "gather_rows src"); _req(idx, torch.int32, "gather_rows idx")
    assert src.is_contiguous() and src.dim() == 2
    if out is None:
        out = torch.empty((n_pad, src.shape[1]), device=src.device, dtype=BF16)
    else:
        _req(out, BF16, "gather_rows out")
        assert out.is_contiguous() and tuple(out.shape) == (n_pad, src.shape[1])
    check(_lib.load().vq3_gather_rows(src.data_ptr(), idx.data_ptr(), out.data_ptr(), n, n_pad, src.shape[1],
                                      _stream()), "vq3_gather_rows")
    return out


def scatter_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor, n: int, accumulate: bool) -> None:
    _req(src, BF16, "scatter_rows src"); _req(idx, torch.int32, "scatter_rows idx"); _req(dst, BF16, "scatter dst")
    assert src.is_contiguous() and dst.is_contiguous() and src.shape[1] == dst.shape[1]
    if n > 0:
        check(_lib.load().vq3_scatter_rows(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), n, src.shape[1],
                                           1 if accumulate else 0, _stream()), "vq3_scatter_rows")


# ----------------------------------------------------------------------------------------------- Qwen3 attention
def qwen_qkprep_fwd(qkv, q_w, k_w, cos, sin, B, L, Hq, Hkv, D, eps, want_rstd=True):
    _req(qkv, BF16, "qkprep qkv"); assert qkv.is_contiguous() and qkv.shape == (B * L, (Hq + 2 * Hkv) * D)
    _req(cos, BF16, "qkprep cos"); _req(sin, BF16, "qkprep sin")
    assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[-2:] == (L, D)
    dev = qkv.device
    Q = torch.empty((B, Hq, L, D), device=dev, dtype=BF16)
    K = torch.empty((B, Hkv, L, D), device=dev, dtype=BF16)
    V = torch.empty((B, Hkv, L, D), device=dev, dtype=BF16)
    qr = torch.empty((B * L * Hq,), device=dev, dtype=F32) if want_rstd else None
    kr = torch.empty((B * L * Hkv,), device=dev, dtype=F32) if want_rstd else None
    check(_lib.load().vq3_qwen_qkprep_fwd(qkv.data_ptr(), q_w.data_ptr(), k_w.data_ptr(), cos.data_ptr(),
                                          sin.data_ptr(), Q.data_ptr(), K.data_ptr(), V.data_ptr(), _p(qr), _p(kr), B,
                                          L, Hq, Hkv, D, eps, _stream()), "vq3_qwen_qkprep_fwd")
    return Q, K, V, qr, kr


QKPREP_BWD_TOKENS_PER_PART = 8      # include/vq3_hip.h: VQ3_QKPREP_BWD_TOKENS_PER_PART (tests/test_abi.py keeps the two equal)


def qwen_qkprep_bwd(dQ, dK, dV, qkv, q_w, k_w, cos, sin, qr, kr, dq_w_out, dk_w_out, accumulate, B, L, Hq, Hkv, D,
                    defer: Optional[list] = None, out: Optional[torch.Tensor] = None):
    """dq_w_out / dk_w_out: bf16 [D] gradient vectors, (+)= per `accumulate`."""
    for t in (dQ, dK, dV, qkv):
        _req(t, BF16, "qkprep_bwd"); assert t.is_contiguous()
    kv_parts = dK.shape[0] if dK.dim() == 5 else 1            # [parts, B, Hkv, L, D] partial slabs of the split dK/dV pass
    assert dV.shape == dK.shape
    dqkv = _out2d(out, qkv.shape[0], qkv.shape[1], qkv, "qwen_qkprep_bwd")
    rows = (B * L + QKPREP_BWD_TOKENS_PER_PART - 1) // QKPREP_BWD_TOKENS_PER_PART      # one partial row per workgroup
    part = torch.empty((2, rows, D), device=qkv.device, dtype=F32)
    lib = _lib.load()
    check(lib.vq3_qwen_qkprep_bwd(dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), qkv.data_ptr(), q_w.data_ptr(),
                                  k_w.data_ptr(), cos.data_ptr(), sin.data_ptr(), qr.data_ptr(), kr.data_ptr(),
                                  dqkv.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), kv_parts, B, L, Hq, Hkv, D,
                                  _stream()), "vq3_qwen_qkprep_bwd")
    if defer is not None:
        defer.append((part[0], rows, D, dq_w_out, accumulate))
        defer.append((part[1], rows, D, dk_w_out, accumulate))
        return dqkv
    acc = 1 if accumulate else 0
    check(lib.vq3_colsum_f32_to_bf16(part[0].data_ptr(), rows, D, dq_w_out.data_ptr(), acc, _stream()), "colsum dq_w")
    check(lib.vq3_colsum_f32_to_bf16(part[1].data_ptr(), rows, D, dk_w_out.data_ptr(), acc, _stream()), "colsum dk_w")
    return dqkv


def softmax_fwd(S: torch.Tensor, keymask: Optional[torch.Tensor], heads_per_mask: int, Lk: int, ldP: int,
                causal: bool) -> torch.Tensor:
    """S f32 [nb, Lq, ldS] -> P bf16 [nb, Lq, ldP]."""
    _req(S, F32, "softmax S"); assert S.is_contiguous() and S.dim() == 3
    nb, Lq, ldS = S.shape
    if keymask is not None:
        _req(keymask, torch.uint8, "softmax keymask"); assert keymask.is_contiguous()
    P = torch.empty((nb, Lq, ldP), device=S.device, dtype=BF16)
    check(_lib.load().vq3_softmax_fwd(S.data_ptr(), P.data_ptr(), _p(keymask), nb, heads_per_mask, Lq, Lk, ldS, ldP,
                                      1 if causal else 0, _stream()), "vq3_softmax_fwd")
    return P


def softmax_bwd(P: torch.Tensor, dP: torch.Tensor, Lk: int, scale: float) -> torch.Tensor:
    _req(P, BF16, "softmax_bwd P"); _req(dP, F32, "softmax_bwd dP")
    assert P.is_contiguous() and dP.is_contiguous()
    nb, Lq, ldP = P.shape
    ldS = dP.shape[2]
    dS = torch.empty_like(P)
    check(_lib.load().vq3_softmax_bwd(P.data_ptr(), dP.data_ptr(), dS.data_ptr(), nb, Lq, Lk, ldS, ldP, scale,
                                      _stream()), "vq3_softmax_bwd")
    return dS


# ----------------------------------------------------------------------------------------------- embedding / loss
def embed_splice_fwd(ids, table, feat, srcmap, B, L, H, S):
    _req(ids, torch.int64, "embed ids"); _req(table, BF16, "embed table"); _req(srcmap, torch.int32, "embed srcmap")
    assert ids.is_contiguous() and table.is_contiguous() and srcmap.is_contiguous()
    if feat is not None:
        _req(feat, BF16, "embed feat"); assert feat.is_contiguous() and feat.shape == (B, S, H)
    out = torch.empty((B, L, H), device=table.device, dtype=BF16)
    check(_lib.load().vq3_embed_splice_fwd(ids.data_ptr(), table.data_ptr(), _p(feat), srcmap.data_ptr(),
                                           out.data_ptr(), B, L, H, S if feat is not None else 0, _stream()),
          "vq3_embed_splice_fwd")
    return out


def embed_splice_bwd(sorted_ids, order, srcmap, dout, dtable, dfeat_f32, B, L, H, S):
    _req(dout, BF16, "embed_bwd dout"); assert dout.is_contiguous()
    check(_lib.load().vq3_embed_splice_bwd(sorted_ids.data_ptr(), order.data_ptr(), srcmap.data_ptr(), dout.data_ptr(),
                                           _p(dtable), _p(dfeat_f32), B, L, H, S, _stream()), "vq3_embed_splice_bwd")


def cross_entropy_fwd_bwd(logits: torch.Tensor, targets: torch.Tensor, loss_sum: torch.Tensor, n: int, V: int,
                          gscale: float) -> None:
    _req(logits, BF16, "ce logits"); _req(targets, torch.int32, "ce targets"); _req(loss_sum, F32, "ce loss_sum")
    assert logits.dim() == 2 and logits.stride(1) == 1
    check(_lib.load().vq3_cross_entropy_fwd_bwd(logits.data_ptr(), targets.data_ptr(), loss_sum.data_ptr(), n, V,
                                                logits.stride(0), gscale, _stream()), "vq3_cross_entropy_fwd_bwd")


def cross_entropy_rows(logits: torch.Tensor, targets: torch.Tensor, row_scale: torch.Tensor, row_loss: torch.Tensor, n: int, V: int) -> None:
    """Per-row form: dlogits[i] = (softmax_i - onehot_i) * row_scale[i] in place, row_loss[i] = the row's cross entropy."""
    _req(logits, BF16, "ce logits"); _req(targets, torch.int32, "ce targets"); _req(row_scale, F32, "ce row_scale"); _req(row_loss, F32, "ce row_loss")
    assert logits.dim() == 2 and logits.stride(1) == 1 and row_scale.numel() >= n and row_loss.numel() >= n
    check(_lib.load().vq3_cross_entropy_rows(logits.data_ptr(), targets.data_ptr(), row_scale.data_ptr(), row_loss.data_ptr(), n, V,
                                             logits.stride(0), _stream()), "vq3_cross_entropy_rows")


def adamw_step(master, m, v, grad, w, lr, beta1, beta2, eps, wd, step, gscale=1.0, clip=None) -> None:
    """clip = (sumsq f32[1] device tensor, max_norm): global-norm clipping, coefficient computed on the device as
    min(1, max_norm / (sqrt(sumsq) * gscale + 1e-6)) (torch.nn.utils.clip_grad_norm_'s formula on the scaled gradients)."""
    _req(master, F32, "adamw master"); _req(m, F32, "adamw m"); _req(v, F32, "adamw v")
    _req(grad, BF16, "adamw grad"); _req(w, BF16, "adamw w")
    n = master.numel()
    assert m.numel() == n and v.numel() == n and grad.numel() == n and w.numel() == n
    sq, mx = (None, 0.0) if clip is None else clip
    if sq is not None:
        _req(sq, F32, "adamw clip sumsq")
    check(_lib.load().vq3_adamw_step(master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), w.data_ptr(), n,
                                     lr, beta1, beta2, eps, wd, step, gscale, _p(sq), mx, _stream()), "vq3_adamw_step")


def perceiver_xattn(q: torch.Tensor, kv: torch.Tensor, B: int, H: int, N: int, T: int, hd: int, Tp: int, p_drop: float = 0.0,
                    seed: int = 0, offset: int = 0, keep_p: bool = False):
    """Fused Perceiver cross-attention (projector_perceiver.py:33,44): q bf16 [B*N, H*hd], kv bf16 [B*T, 2*H*hd] (k | v) ->
    o bf16 [B*N, H*hd]; with keep_p also (P, Pd) bf16 [B*H, N, Tp] as softmax_fwd + dropout_ would have left them (Pd is P when
    p_drop == 0). The dropout mask is the one dropout_(P, p_drop, seed, offset) applies to the [B*H, N, Tp] tensor."""
    _req(q, BF16, "perceiver_xattn q"); _req(kv, BF16, "perceiver_xattn kv")
    D = H * hd
    assert q.is_contiguous() and kv.is_contiguous() and q.shape == (B * N, D) and kv.shape == (B * T, 2 * D)
    o = torch.empty((B * N, D), device=q.device, dtype=BF16)
    P = Pd = None
    if keep_p:
        P = torch.empty((B * H, N, Tp), device=q.device, dtype=BF16)
        Pd = torch.empty_like(P) if p_drop > 0.0 else None
    check(_lib.load().vq3_perceiver_xattn_fwd(q.data_ptr(), kv.data_ptr(), o.data_ptr(), _p(P), _p(Pd), B, H, N, T, hd, D, 2 * D, D, D, Tp,
                                              float(hd) ** -0.5, float(p_drop), seed & (2 ** 64 - 1), offset & (2 ** 64 - 1), _stream()),
          "vq3_perceiver_xattn_fwd")
    if keep_p:
        return o, P, (Pd if Pd is not None else P)
    return o


def dropout_(x: torch.Tensor, p: float, seed: int, offset: int) -> torch.Tensor:
    """In-place inverted dropout (bf16 or f32), mask = hash(seed, offset + element index)."""
    assert x.is_contiguous() and x.is_cuda and x.dtype in (BF16, F32)
    if p > 0.0 and x.numel():
        check(_lib.load().vq3_dropout(x.data_ptr(), 1 if x.dtype == F32 else 0, x.numel(), float(p), seed & (2 ** 64 - 1),
                                      offset & (2 ** 64 - 1), _stream()), "vq3_dropout")
    return x


def sumsq(x: torch.Tensor, partials: torch.Tensor, accum: torch.Tensor) -> None:
    """accum[0] += sum(x^2) (bf16 or f32 x), deterministic: per-block partials (scratch, >= 1024 floats) summed in a fixed order."""
    assert x.is_contiguous() and x.is_cuda and x.dtype in (BF16, F32)
    _req(partials, F32, "sumsq partials"); _req(accum, F32, "sumsq accum")
    assert partials.numel() >= 1024
    if x.numel():
        check(_lib.load().vq3_sumsq(x.data_ptr(), 1 if x.dtype == F32 else 0, x.numel(), partials.data_ptr(),
                                    accum.data_ptr(), _stream()), "vq3_sumsq")


# ----------------------------------------------------------------------------------------------- VGGT
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def im2col_norm(images: torch.Tensor, p: int, Kp: int) -> torch.Tensor:
    """images f32 [NI,3,H,W] -> bf16 [NI*(H/p)*(W/p), Kp] normalised patches."""
    import ctypes as C
    _req(images, F32, "im2col images"); assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    NI, _, H, W = images.shape
    out = torch.empty((NI * (H // p) * (W // p), Kp), device=images.device, dtype=BF16)
    mean = (C.c_float * 3)(*IMAGENET_MEAN); std = (C.c_float * 3)(*IMAGENET_STD)
    check(_lib.load().vq3_im2col_norm(images.data_ptr(), out.data_ptr(), NI, H, W, p, Kp, mean, std, _stream()),
          "vq3_im2col_norm")
    return out


def vit_qkprep(qkv, N, NH, *, qn=None, kn=None, cos=None, sin=None, tokens_per_frame=0, patch_start=0, Wp=0,
               eps=1e-5):
    """qkv bf16 [T, 3*NH*64] -> Q, K, V bf16 [T/N, NH, N, 64]."""
    _req(qkv, BF16, "vit_qkprep qkv"); assert qkv.is_contiguous()
    T = qkv.shape[0]
    G = T // N
    dev = qkv.device
    Q = torch.empty((G, NH, N, 64), device=dev, dtype=BF16)
    K = torch.empty_like(Q); V = torch.empty_like(Q)
    use_norm = qn is not None
    use_rope = cos is not None
    check(_lib.load().vq3_vit_qkprep(qkv.data_ptr(), _p(qn[0]) if use_norm else None, _p(qn[1]) if use_norm else None,
                                     _p(kn[0]) if use_norm else None, _p(kn[1]) if use_norm else None, _p(cos), _p(sin),
                                     Q.data_ptr(), K.data_ptr(), V.data_ptr(), T, N, NH, 64, tokens_per_frame,
                                     patch_start, Wp, 1 if use_norm else 0, 1 if use_rope else 0, eps, _stream()),
          "vq3_vit_qkprep")
    return Q, K, V


def gemm_swiglu_bwd(dY: torch.Tensor, W: torch.Tensor, gu: torch.Tensor, transB: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu_mul_bwd(dY @ W (or dY @ W^T), gu) in one launch: dY bf16 [M, K]; W bf16 [K, N] (transB: k-major, as a weight [out=K, in=N]
    is stored) or [N, K]; gu bf16 [M, 2N] -> dgu bf16 [M, 2N]. d(act) is never materialised (vq3_gemm_swiglu_bwd)."""
    _req(dY, BF16, "swiglu_bwd dY"); _req(W, BF16, "swiglu_bwd W"); _req(gu, BF16, "swiglu_bwd gu")
    assert dY.dim() == 2 and W.dim() == 2 and dY.stride(1) == 1 and W.stride(1) == 1 and gu.is_contiguous()
    M, K = dY.shape
    N = W.shape[1] if transB else W.shape[0]
    assert (W.shape[0] if transB else W.shape[1]) == K and gu.shape == (M, 2 * N)
    dgu = _out2d(out, M, 2 * N, gu, "gemm_swiglu_bwd")
    d = GemmDesc()
    d.A = dY.data_ptr(); d.B = W.data_ptr(); d.C = None
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = M, N, K, dY.stride(0), W.stride(0), N, 0
    d.nb1 = d.nb2 = d.b2divB = 1
    d.alpha = 1.0
    d.transA, d.transB = 0, 1 if transB else 0
    if _TUNE_WS is None:
        gemm_tune_setup()
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().vq3_gemm_swiglu_bwd(d, gu.data_ptr(), dgu.data_ptr(), _stream()), "vq3_gemm_swiglu_bwd")
        e1.record()
        GEMM_PROFILE.append((2.0 * M * N * K, 2.0 * M * K + 2.0 * N * K + 2.0 * M * N * 4, e0, e1, (M, N, K, 1)))
    else:
        check(_lib.load().vq3_gemm_swiglu_bwd(d, gu.data_ptr(), dgu.data_ptr(), _stream()), "vq3_gemm_swiglu_bwd")
    return dgu


def swiglu_fwd_fusable(M: int, inter: int, K: int) -> bool:
    """Shapes vq3_gemm_swiglu_fwd takes (the 8-phase NT kernels: K % 64 == 0; gate and up of a feature in one tile: I % 128 == 0)."""
    return K % 64 == 0 and inter % 128 == 0 and M >= 1 and os.environ.get("VQ3_SWIGLU_FWD_FUSED", "1") != "0"


def gemm_swiglu_fwd(x: torch.Tensor, w_gu: torch.Tensor, gu_out: Optional[torch.Tensor] = None, act_out: Optional[torch.Tensor] = None,
                    keep_gu: bool = True):
    """(gu, act) with gu = x @ w_gu^T [M, 2 I] = gate | up and act = silu_mul_fwd(gu) [M, I], in ONE launch (vq3_gemm_swiglu_fwd):
    x bf16 [M, K], w_gu bf16 [2 I, K] (gate rows, then up rows - the fused weight of modeling_qwen3.py:81-83).
    keep_gu=False (a forward without a backward): gate|up is not written, (None, act) is returned."""
    _req(x, BF16, "swiglu_fwd x"); _req(w_gu, BF16, "swiglu_fwd w")
    assert x.dim() == 2 and w_gu.dim() == 2 and x.stride(1) == 1 and w_gu.stride(1) == 1 and x.shape[1] == w_gu.shape[1]
    M, K = x.shape
    N = w_gu.shape[0]
    gu = _out2d(gu_out, M, N, x, "gemm_swiglu_fwd gu") if keep_gu else None
    act = _out2d(act_out, M, N // 2, x, "gemm_swiglu_fwd act")
    assert gu is None or gu.is_contiguous()
    gu_ptr = gu.data_ptr() if gu is not None else None
    d = GemmDesc()
    d.A = x.data_ptr(); d.B = w_gu.data_ptr(); d.C = act.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = M, N, K, x.stride(0), w_gu.stride(0), act.stride(0), 0
    d.nb1 = d.nb2 = d.b2divB = 1
    d.alpha = 1.0
    if _TUNE_WS is None:
        gemm_tune_setup()
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().vq3_gemm_swiglu_fwd(d, gu_ptr, _stream()), "vq3_gemm_swiglu_fwd")
        e1.record()
        GEMM_PROFILE.append((2.0 * M * N * K, 2.0 * M * K + 2.0 * N * K + 2.0 * M * N * (1.5 if keep_gu else 0.5), e0, e1, (M, N, K, 1)))
    else:
        check(_lib.load().vq3_gemm_swiglu_fwd(d, gu_ptr, _stream()), "vq3_gemm_swiglu_fwd")
    return gu, act


def linear_vit_qkv(x: torch.Tensor, w: torch.Tensor, bias, N: int, NH: int, *, qn=None, kn=None, cos=None, sin=None,
                   tokens_per_frame=0, patch_start=0, Wp=0, eps=1e-5, ln_fold=None):
    """vit_qkprep(linear(x, w, bias)) in ONE launch: x bf16 [T, C] @ w[3*NH*64, C]^T (+ bias) -> Q, K, V bf16 [T/N, NH, N, 64]
    with the per-head LayerNorm / 2-D RoPE applied in the GEMM epilogue (vq3_gemm_vit_qkv); qkv is never materialised."""
    _req(x, BF16, "vit_qkv x"); _req(w, BF16, "vit_qkv w")
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] == w.shape[1]
    T, K = x.shape
    assert w.shape[0] == 3 * NH * 64 and T % N == 0
    G = T // N
    Q = torch.empty((G, NH, N, 64), device=x.device, dtype=BF16)
    Kt = torch.empty_like(Q); V = torch.empty_like(Q)
    d = GemmDesc()
    d.A = x.data_ptr(); d.B = w.data_ptr(); d.C = None
    d.bias = _p(bias); d.colscale = None; d.R = None
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = T, 3 * NH * 64, K, x.stride(0), w.stride(0), 3 * NH * 64, 0
    d.nb1 = d.nb2 = d.b2divB = 1
    d.alpha = 1.0
    e = _lib.VitQkvEpilogue()
    e.Q, e.K, e.V = Q.data_ptr(), Kt.data_ptr(), V.data_ptr()
    use_norm, use_rope = qn is not None, cos is not None
    if use_norm:
        e.qn_w, e.qn_b, e.kn_w, e.kn_b = qn[0].data_ptr(), qn[1].data_ptr(), kn[0].data_ptr(), kn[1].data_ptr()
    if use_rope:
        e.cos, e.sin = cos.data_ptr(), sin.data_ptr()
    e.N, e.NH, e.tokens_per_frame, e.patch_start, e.Wp = N, NH, tokens_per_frame, patch_start, Wp
    e.use_norm, e.use_rope, e.eps = int(use_norm), int(use_rope), eps
    lib = _lib.load()
    if _TUNE_WS is None:
        gemm_tune_setup()
    launch = (lambda: check(lib.vq3_gemm_vit_qkv(d, e, _stream()), "vq3_gemm_vit_qkv")) if ln_fold is None else \
             (lambda: check(lib.vq3_gemm_vit_qkv_ln(d, e, ln_fold, _stream()), "vq3_gemm_vit_qkv_ln"))
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        GEMM_PROFILE.append((2.0 * T * d.N * K, 2.0 * T * K + 2.0 * d.N * K + 2.0 * T * d.N, e0, e1, (T, d.N, K, 1)))
    else:
        launch()
    return Q, Kt, V


def flash_attn(Q, K, V, out: Optional[torch.Tensor] = None, q_rows: Optional[int] = None, score_bound: Optional[float] = None) -> torch.Tensor:
    """softmax(Q K^T / 8) V for Q,K,V bf16 [G, NH, N, 64] -> token-major [G*N, NH*64]; with q_rows only the first q_rows queries of
    every group are computed (all N keys) -> [G*q_rows, NH*64]. score_bound: the caller's promise |q . k| / 8 <= score_bound for every
    pair (vq3_flash_attn_fwd_bounded: a small bound selects the kernels without a running maximum; None = no promise)."""
    _req(Q, BF16, "flash Q"); _req(K, BF16, "flash K"); _req(V, BF16, "flash V")
    assert Q.is_contiguous() and K.is_contiguous() and V.is_contiguous() and K.shape == Q.shape and V.shape == Q.shape
    G, NH, N, D = Q.shape
    nq = N if q_rows is None else int(q_rows)
    if out is None:
        out = torch.empty((G * nq, NH * D), device=Q.device, dtype=BF16)
    assert out.shape[0] >= G * nq
    if score_bound is not None:
        check(_lib.load().vq3_flash_attn_fwd_bounded(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), G, NH, N, nq, D,
                                                     out.stride(0), D ** -0.5, float(score_bound), _stream()), "vq3_flash_attn_fwd_bounded")
    elif q_rows is None:
        check(_lib.load().vq3_flash_attn_fwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), G, NH, N, D,
                                             out.stride(0), D ** -0.5, _stream()), "vq3_flash_attn_fwd")
    else:
        check(_lib.load().vq3_flash_attn_fwd_rows(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), G, NH, N, nq, D,
                                                  out.stride(0), D ** -0.5, _stream()), "vq3_flash_attn_fwd_rows")
    return out


# ---------------------------------------------------------------------------------------------- decoding
def skinny_linear(x: torch.Tensor, w: torch.Tensor, *, residual: Optional[torch.Tensor] = None, n: Optional[int] = None,
                  out: Optional[torch.Tensor] = None, out_dtype=BF16, ln_w: Optional[torch.Tensor] = None, eps: float = 0.0,
                  swiglu: bool = False) -> torch.Tensor:
    """y = f(x) w^T (+ residual) for 1..8 rows: streams the weight once (vq3_skinny_gemm_bf16). `n` limits the output
    features to the first n rows of w (tied lm_head: the vocabulary rows of the padded embedding). ln_w: RMSNorm with
    that weight is applied to x on the fly; swiglu: x is [M, 2K] = gate | up and f = silu(gate) * up."""
    _req(x, BF16, "skinny x"); _req(w, BF16, "skinny w")
    assert x.dim() == 2 and w.dim() == 2 and x.is_contiguous() and w.is_contiguous()
    assert not (swiglu and ln_w is not None)
    M, K = x.shape[0], w.shape[1]
    assert x.shape[1] == (2 * K if swiglu else K)
    N = w.shape[0] if n is None else n
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=out_dtype)
    if residual is not None:
        _req(residual, BF16, "skinny residual"); assert residual.shape == (M, N) and residual.is_contiguous()
    xmode = 2 if swiglu else (1 if ln_w is not None else 0)
    check(_lib.load().vq3_skinny_gemm_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), _p(residual), _p(ln_w), eps, xmode,
                                           M, N, K, x.shape[1], K, out.stride(0), N, 1 if out.dtype == F32 else 0,
                                           _stream()), "vq3_skinny_gemm_bf16")
    return out


def qwen_decode_qkprep(qkv, q_w, k_w, cos, sin, lens, Kc, Vc, B, Hq, Hkv, D, Lmax, eps) -> torch.Tensor:
    _req(qkv, BF16, "decode qkv"); _req(lens, torch.int32, "decode lens"); _req(Kc, BF16, "K cache"); _req(Vc, BF16, "V cache")
    assert qkv.is_contiguous() and qkv.shape == (B, (Hq + 2 * Hkv) * D)
    assert Kc.is_contiguous() and Vc.is_contiguous() and Kc.shape == (B, Hkv, Lmax, D) and Vc.shape == Kc.shape
    assert cos.shape[-2] >= Lmax and cos.shape[-1] == D and cos.is_contiguous() and sin.is_contiguous()
    Q = torch.empty((B, Hq * D), device=qkv.device, dtype=BF16)
    check(_lib.load().vq3_qwen_decode_qkprep(qkv.data_ptr(), q_w.data_ptr(), k_w.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                             lens.data_ptr(), Q.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), B, Hq, Hkv, D,
                                             Lmax, eps, _stream()), "vq3_qwen_decode_qkprep")
    return Q


def qwen_decode_attn(Q, Kc, Vc, lens, B, Hq, Hkv, D, Lmax, scale) -> torch.Tensor:
    _req(Q, BF16, "decode Q"); assert Q.is_contiguous() and Q.shape == (B, Hq * D)
    O = torch.empty_like(Q)
    check(_lib.load().vq3_qwen_decode_attn(Q.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), lens.data_ptr(), O.data_ptr(), B, Hq,
                                           Hkv, D, Lmax, scale, _stream()), "vq3_qwen_decode_attn")
    return O


def greedy_pick(logits, work, generated, step, finished, penalty, ngram, eos_ids, pad_id, next_ids, V) -> None:
    _req(logits, BF16, "pick logits"); _req(work, F32, "pick work"); _req(generated, torch.int64, "pick generated")
    _req(step, torch.int32, "pick step"); _req(finished, torch.int32, "pick finished"); _req(next_ids, torch.int32, "pick next")
    B = logits.shape[0]
    assert logits.stride(1) == 1 and work.numel() >= B * 128 + 1 and generated.is_contiguous() and generated.shape[0] == B
    n_eos = 0 if eos_ids is None else int(eos_ids.numel())
    check(_lib.load().vq3_greedy_pick(logits.data_ptr(), logits.stride(0), work.data_ptr(), B, V, generated.data_ptr(),
                                      generated.shape[1], step.data_ptr(), finished.data_ptr(), float(penalty), int(ngram),
                                      _p(eos_ids), n_eos, int(pad_id), next_ids.data_ptr(), _stream()), "vq3_greedy_pick")


def decode_advance(lens: Optional[torch.Tensor], B: int, step: Optional[torch.Tensor]) -> None:
    check(_lib.load().vq3_decode_advance(_p(lens), B, _p(step), _stream()), "vq3_decode_advance")


def decode_layers_supported(hidden: int, intermediate: int, Hq: int, Hkv: int, head_dim: int, Lmax: int) -> bool:
    """True when vq3_qwen_decode_layers (all decoder layers of a B = 1 decode step in one persistent launch) takes this shape on this device."""
    return bool(_lib.load().vq3_qwen_decode_layers_supported(hidden, intermediate, Hq, Hkv, head_dim, Lmax))


def decode_layers_workspace(device) -> torch.Tensor:
    """The rows that travel between the persistent decode kernel's workgroups (vq3_qwen_decode_layers_workspace_bytes)."""
    return torch.zeros(int(_lib.load().vq3_qwen_decode_layers_workspace_bytes()) // 2, device=device, dtype=BF16)


def decode_layers(wtab: torch.Tensor, h: torch.Tensor, workspace: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, lens: torch.Tensor,
                  K: torch.Tensor, V: torch.Tensor, barrier: torch.Tensor, status: torch.Tensor, hidden: int, intermediate: int, Hq: int,
                  Hkv: int, eps: float, scale: float) -> None:
    """h [1, hidden] <- the decoder stack applied to it for the token at position lens[0] (vq3_qwen_decode_layers). wtab: int64
    [layers, 8] device pointers (qkv, o, gate|up, down, ln1, ln2, q_norm, k_norm); K / V: [layers, 1, Hkv, Lmax, 128] caches; `workspace`
    from decode_layers_workspace(); `barrier` (int32 [1024]) is zeroed here on the stream, `status` (int32 [1]) is sticky - see
    decode_layers_status()."""
    for t, nm in ((h, "h"), (cos, "cos"), (sin, "sin"), (K, "K"), (V, "V")):
        _req(t, BF16, "decode_layers " + nm)
        assert t.is_contiguous(), nm
    _req(wtab, torch.int64, "decode_layers wtab"); _req(lens, torch.int32, "decode_layers lens"); _req(workspace, BF16, "decode_layers workspace")
    _req(barrier, torch.int32, "decode_layers barrier"); _req(status, torch.int32, "decode_layers status")
    nl = wtab.shape[0]
    assert wtab.is_contiguous() and wtab.shape[1] == 8 and K.dim() == 5 and K.shape[0] == nl and K.shape[1] == 1 and K.shape == V.shape
    Lmax, D = K.shape[3], K.shape[4]
    assert h.numel() == hidden and K.shape[2] == Hkv and workspace.is_contiguous()
    assert workspace.numel() * 2 >= int(_lib.load().vq3_qwen_decode_layers_workspace_bytes())
    assert barrier.numel() >= 1024 and barrier.is_contiguous() and workspace.data_ptr() % 16 == 0
    assert cos.shape[0] >= Lmax and cos.shape[1] == D and sin.shape == cos.shape
    import ctypes as C
    barrier.zero_()
    d = _lib.DecodeLayersDesc(wtab.data_ptr(), h.data_ptr(), workspace.data_ptr(), cos.data_ptr(), sin.data_ptr(), lens.data_ptr(),
                              K.data_ptr(), V.data_ptr(), K.stride(0), barrier.data_ptr(), status.data_ptr(), nl, hidden, intermediate,
                              Hq, Hkv, D, Lmax, float(eps), float(scale))
    check(_lib.load().vq3_qwen_decode_layers(C.byref(d), _stream()), "vq3_qwen_decode_layers")


def decode_layers_status(status: torch.Tensor) -> None:
    """Raise if a persistent decode launch reported a failure (synchronises)."""
    st = int(status.item())
    if st & 1:
        raise RuntimeError("vq3_qwen_decode_layers: a grid-barrier wait ran out (the 256 workgroups were not co-resident - is another "
                           "kernel holding CUs?); set VQ3_DECODE_PERSISTENT=0 to decode with one launch per projection")
    if st & 2:
        raise RuntimeError("vq3_qwen_decode_layers: KV cache full")


# ---------------------------------------------------------------------------------------------- fp8 forward (config C5)
def quant_fp8_rows(x: torch.Tensor):
    """bf16 [rows, K] -> (uint8 e4m3 [rows, K], f32 scale [rows]); x ~ q * scale[:, None]."""
    _req(x, BF16, "quant_fp8 x")
    assert x.dim() == 2 and x.stride(1) == 1
    rows, K = x.shape
    q = torch.empty((rows, K), device=x.device, dtype=torch.uint8)
    s = torch.empty((rows,), device=x.device, dtype=F32)
    check(_lib.load().vq3_quant_fp8_rows(x.data_ptr(), x.stride(0), rows, K, q.data_ptr(), K, s.data_ptr(), _stream()),
          "vq3_quant_fp8_rows")
    return q, s


def gemm_fp8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, *, residual: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [M, N] = (xs[:, None] * ws[None, :]) * (xq @ wq^T) (+ residual); xq [M, K], wq [N, K] e4m3 bytes."""
    _req(xq, torch.uint8, "gemm_fp8 xq"); _req(wq, torch.uint8, "gemm_fp8 wq"); _req(xs, F32, "xs"); _req(ws, F32, "ws")
    assert xq.dim() == 2 and wq.dim() == 2 and xq.is_contiguous() and wq.is_contiguous() and xq.shape[1] == wq.shape[1]
    M, K = xq.shape
    N = wq.shape[0]
    assert xs.numel() == M and ws.numel() == N
    if out is None:
        out = torch.empty((M, N), device=xq.device, dtype=BF16)
    if residual is not None:
        _req(residual, BF16, "gemm_fp8 residual"); assert residual.shape == (M, N) and residual.is_contiguous()
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.load().vq3_gemm_fp8_nt(xq.data_ptr(), xs.data_ptr(), wq.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                      _p(residual), M, N, K, K, K, out.stride(0), N, _stream()), "vq3_gemm_fp8_nt")
    if prof is not None:
        e1.record()
        prof.append((2.0 * M * N * K, 1.0 * M * K + 1.0 * N * K + 2.0 * M * N * (2 if residual is not None else 1), e0, e1, ("e4m3", M, N, K)))
    return out


def linear_fp8(x: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, *, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    xq, xs = quant_fp8_rows(x)
    return gemm_fp8(xq, xs, wq, ws, residual=residual)


def quant_fp8_rows_scaled(x: torch.Tensor, colmul: torch.Tensor):
    """(q uint8 [rows, K], scale f32 [rows]) of x * colmul[None, :] (colmul f32 [K]): dY with the weight's per-output-channel scales folded in."""
    _req(x, BF16, "quant_fp8 x"); _req(colmul, F32, "quant_fp8 colmul")
    assert x.dim() == 2 and x.stride(1) == 1 and colmul.numel() == x.shape[1]
    rows, K = x.shape
    q = torch.empty((rows, K), device=x.device, dtype=torch.uint8)
    s = torch.empty((rows,), device=x.device, dtype=F32)
    check(_lib.load().vq3_quant_fp8_rows_scaled(x.data_ptr(), x.stride(0), rows, K, colmul.data_ptr(), q.data_ptr(), K, s.data_ptr(), _stream()),
          "vq3_quant_fp8_rows_scaled")
    return q, s


def transpose_u8(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R, C] uint8 -> [C, R] (the e4m3 W^T copies of the dgrad GEMMs)."""
    _req(src, torch.uint8, "transpose_u8"); assert src.dim() == 2 and src.is_contiguous()
    R, Cc = src.shape
    if out is None:
        out = torch.empty((Cc, R), device=src.device, dtype=torch.uint8)
    check(_lib.load().vq3_transpose_u8(src.data_ptr(), out.data_ptr(), R, Cc, Cc, R, _stream()), "vq3_transpose_u8")
    return out


def gemm_fp8_ex(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: Optional[torch.Tensor], *, mode: int = 0,
                residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, gu: Optional[torch.Tensor] = None,
                keep_gu: bool = True):
    """vq3_gemm_fp8_ex. mode 0: out [M, N] (+ residual); mode 1 (SwiGLU forward): wq = gate|up weight [2 I, K] -> (gu [M, 2 I] or None, act
    [M, I] = out); mode 2 (SwiGLU backward): gu = saved gate|up [M, 2 N] -> dgu [M, 2 N]. ws None: the column scales were folded into xq."""
    _req(xq, torch.uint8, "gemm_fp8 xq"); _req(wq, torch.uint8, "gemm_fp8 wq"); _req(xs, F32, "xs")
    assert xq.dim() == 2 and wq.dim() == 2 and xq.is_contiguous() and wq.is_contiguous() and xq.shape[1] == wq.shape[1]
    M, K = xq.shape
    N = wq.shape[0]
    d = _lib.GemmFp8Desc()
    d.Xq, d.x_scale, d.Wq, d.w_scale = xq.data_ptr(), xs.data_ptr(), wq.data_ptr(), _p(ws)
    d.M, d.N, d.K, d.ldx, d.ldw, d.mode = M, N, K, K, K, mode
    dev = xq.device
    if mode == 0:
        out = _out2d(out, M, N, xq, "gemm_fp8_ex")
        d.C, d.ldc = out.data_ptr(), out.stride(0)
        if residual is not None:
            _req(residual, BF16, "gemm_fp8 residual"); assert residual.shape == (M, N) and residual.stride(1) == 1
            d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
        ret = out
    elif mode == 1:
        inter = N // 2
        if out is None:
            out = torch.empty((M, inter), device=dev, dtype=BF16)
        assert out.shape == (M, inter) and out.stride(1) == 1 and out.dtype == BF16
        g = None
        if keep_gu:
            g = gu if gu is not None else torch.empty((M, N), device=dev, dtype=BF16)
            assert g.shape == (M, N) and g.is_contiguous()
        d.C, d.ldc, d.gu = out.data_ptr(), out.stride(0), _p(g)
        ret = (g, out)
    else:
        _req(gu, BF16, "gemm_fp8 gu"); assert gu.shape == (M, 2 * N) and gu.is_contiguous()
        dgu = out if out is not None else torch.empty((M, 2 * N), device=dev, dtype=BF16)
        assert dgu.shape == (M, 2 * N) and dgu.is_contiguous()
        d.gu, d.dgu = gu.data_ptr(), dgu.data_ptr()
        ret = dgu
    import ctypes as C
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.load().vq3_gemm_fp8_ex(C.byref(d), _stream()), "vq3_gemm_fp8_ex")
    if prof is not None:
        e1.record()
        prof.append((2.0 * M * N * K, 1.0 * M * K + 1.0 * N * K + 2.0 * M * N, e0, e1, ("e4m3 mode %d" % mode, M, N, K)))
    return ret


# ---------------------------------------------------------------------------------------------- fused Qwen3 attention
def qwen_flash_fwd(Q, K, V, keymask, B, L, Hq, Hkv, D, scale, out: Optional[torch.Tensor] = None):
    """-> (O bf16 [B*L, Hq*D] token-major, LSE f32 [B, Hq, L])."""
    for t in (Q, K, V):
        _req(t, BF16, "qwen_flash"); assert t.is_contiguous()
    _req(keymask, torch.uint8, "qwen_flash keymask")
    assert Q.shape == (B, Hq, L, D) and K.shape == (B, Hkv, L, D) and V.shape == K.shape and keymask.shape == (B, L)
    O = _out2d(out, B * L, Hq * D, Q, "qwen_flash_fwd")
    lse = torch.empty(B * Hq * L + 4, device=Q.device, dtype=F32)[: B * Hq * L].view(B, Hq, L)   # 16 B of slack
    check(_lib.load().vq3_qwen_flash_fwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), keymask.data_ptr(), O.data_ptr(),
                                         lse.data_ptr(), B, L, Hq, Hkv, D, Hq * D, scale, _stream()),
          "vq3_qwen_flash_fwd")
    return O, lse


def qwen_flash_bwd(Q, K, V, keymask, O, dO, lse, B, L, Hq, Hkv, D, scale, kv_parts: Optional[int] = None):
    """O, dO: bf16 [B*L, Hq*D] token-major. -> dQ [B,Hq,L,D]; dK, dV bf16 [kv_parts, B,Hkv,L,D] partial slabs (kv_parts = 1 gives
    plain [B,Hkv,L,D], the default); qwen_qkprep_bwd adds the slabs. Worth it when most key tiles are padding (Qwen3ForCausalLM.forward_hidden decides)."""
    _req(O, BF16, "flash_bwd O"); _req(dO, BF16, "flash_bwd dO"); _req(lse, F32, "flash_bwd lse")
    assert O.shape == (B * L, Hq * D) and dO.shape == O.shape and O.stride(1) == 1 and dO.stride(1) == 1
    dev = Q.device
    if kv_parts is None:
        kv_parts = 1
    dQ = torch.empty((B, Hq, L, D), device=dev, dtype=BF16)
    kv_shape = (B, Hkv, L, D) if kv_parts == 1 else (kv_parts, B, Hkv, L, D)
    dK = torch.empty(kv_shape, device=dev, dtype=BF16)
    dV = torch.empty(kv_shape, device=dev, dtype=BF16)
    delta = torch.empty(B * Hq * L + 4, device=dev, dtype=F32)
    check(_lib.load().vq3_qwen_flash_bwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), keymask.data_ptr(), O.data_ptr(),
                                         dO.data_ptr(), lse.data_ptr(), delta.data_ptr(), dQ.data_ptr(), dK.data_ptr(),
                                         dV.data_ptr(), kv_parts, B, L, Hq, Hkv, D, O.stride(0), dO.stride(0), scale,
                                         _stream()), "vq3_qwen_flash_bwd")
    return dQ, dK, dV


def skinny_linear_fp8(x: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, *, residual: Optional[torch.Tensor] = None,
                      ln_w: Optional[torch.Tensor] = None, eps: float = 0.0, swiglu: bool = False) -> torch.Tensor:
    """Decode-time form of linear_fp8 for 1-2 rows: e4m3 weight rows streamed once, the activation row prepared (RMSNorm /
    SwiGLU) and quantised per token in registers (vq3_skinny_gemm_fp8)."""
    _req(x, BF16, "skinny_fp8 x"); _req(wq, torch.uint8, "skinny_fp8 wq"); _req(ws, F32, "skinny_fp8 ws")
    assert x.dim() == 2 and wq.dim() == 2 and x.is_contiguous() and wq.is_contiguous() and not (swiglu and ln_w is not None)
    M, (N, K) = x.shape[0], wq.shape
    assert x.shape[1] == (2 * K if swiglu else K) and ws.numel() == N
    out = torch.empty((M, N), device=x.device, dtype=BF16)
    if residual is not None:
        _req(residual, BF16, "skinny_fp8 residual"); assert residual.shape == (M, N) and residual.is_contiguous()
    xmode = 2 if swiglu else (1 if ln_w is not None else 0)
    check(_lib.load().vq3_skinny_gemm_fp8(x.data_ptr(), wq.data_ptr(), ws.data_ptr(), out.data_ptr(), _p(residual), _p(ln_w),
                                          eps, xmode, M, N, K, x.shape[1], K, N, N, _stream()), "vq3_skinny_gemm_fp8")
    return out


# ---------------------------------------------------------------------------------------------- benchmarking hook
def gemm_split_plan(M: int, N: int, K: int, ncu: int = 256):
    """(full, rem, slices) of the 256x256 kernel's last-round K split for an M x N x K product (host-only); slices == 0: no split."""
    import ctypes as C
    full, rem, sl = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    check(_lib.load().vq3_gemm_split_plan(M, N, K, ncu, C.addressof(full), C.addressof(rem), C.addressof(sl)), "vq3_gemm_split_plan")
    return full.value, rem.value, sl.value


def gemm_split_gave_up() -> bool:
    """Synchronises the current stream, then reads AND clears the process-wide sticky error word of the split GEMM launches: True if a
    reducer's bounded wait expired (on any stream) since the last clearing read - that launch's output is incomplete."""
    import ctypes as C
    flag = C.c_int32(0)
    check(_lib.load().vq3_gemm_split_status(_stream(), C.addressof(flag)), "vq3_gemm_split_status")
    return bool(flag.value)


def gemm_split_poll(clear: bool = True) -> bool:
    """The same word read from the host with NO synchronisation (it lives in host-mapped memory): what the launches completed so far
    have reported. Cheap enough for once per optimiser step (Stage1Trainer.check_kernels)."""
    import ctypes as C
    flag = C.c_int32(0)
    check(_lib.load().vq3_gemm_split_poll(C.addressof(flag), 1 if clear else 0), "vq3_gemm_split_poll")
    return bool(flag.value)


def gemm_split_debug_spin_bound(polls: int = 0) -> None:
    """Tests only: bound of the split reducer's wait in polls (0 = production, 1 << 23)."""
    check(_lib.load().vq3_gemm_split_debug_spin_bound(int(polls)), "vq3_gemm_split_debug_spin_bound")


def gemm_force_config(cfg: int = -3) -> None:
    """Force vq3_gemm_bf16_nt's tile configuration for NT shapes (-3 = automatic). Benchmarks and tests only."""
    check(_lib.load().vq3_gemm_force_config(cfg), "vq3_gemm_force_config")
