"""ctypes binding of libvq3hip.so (C ABI declared in include/vq3_hip.h).

The library is the product: there is no PyTorch/CPU fallback. If it is missing, ``load()`` raises.
PyTorch is only used for device memory and streams; tensors cross the boundary as raw device pointers.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libvq3hip.so"
TUNE_TABLE = _PKG / "gemm_tune_gfx950.txt"

c_p = C.c_void_p
i32, i64, f32 = C.c_int32, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_p), ("B", c_p), ("C", c_p), ("bias", c_p), ("colscale", c_p), ("R", c_p),
        ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32), ("ldr", i32),
        ("sA1", i64), ("sA2", i64), ("sB1", i64), ("sB2", i64),
        ("sC1", i64), ("sC2", i64), ("sR1", i64), ("sR2", i64),
        ("nb1", i32), ("nb2", i32), ("b2divB", i32),
        ("act", i32), ("out_f32", i32), ("accumulate", i32),
        ("alpha", f32),
        ("transA", i32), ("transB", i32), ("ksplit", i32),
    ]


class VitQkvEpilogue(C.Structure):
    _fields_ = [("Q", c_p), ("K", c_p), ("V", c_p), ("qn_w", c_p), ("qn_b", c_p), ("kn_w", c_p), ("kn_b", c_p),
                ("cos", c_p), ("sin", c_p), ("N", i32), ("NH", i32), ("tokens_per_frame", i32), ("patch_start", i32),
                ("Wp", i32), ("use_norm", i32), ("use_rope", i32), ("eps", f32)]


class GemmLnFold(C.Structure):
    _fields_ = [("stats_in", c_p), ("parts_in", i32), ("eps", f32), ("colsum", c_p), ("stats_out", c_p)]


class GemmFp8Desc(C.Structure):
    _fields_ = [("Xq", c_p), ("x_scale", c_p), ("Wq", c_p), ("w_scale", c_p), ("C", c_p), ("residual", c_p),
                ("M", i32), ("N", i32), ("K", i32), ("ldx", i64), ("ldw", i64), ("ldc", i64), ("ldr", i64),
                ("mode", i32), ("gu", c_p), ("dgu", c_p)]


class DecodeLayersDesc(C.Structure):
    _fields_ = [("weights", c_p), ("h", c_p), ("workspace", c_p), ("cos", c_p), ("sin", c_p),
                ("lens", c_p), ("Kcache", c_p), ("Vcache", c_p), ("cache_layer_stride", i64), ("barrier", c_p), ("status", c_p),
                ("layers", i32), ("hidden", i32), ("intermediate", i32), ("Hq", i32), ("Hkv", i32), ("head_dim", i32), ("Lmax", i32),
                ("eps", f32), ("scale", f32)]


class ColsumJob(C.Structure):
    _fields_ = [("part", c_p), ("out_bf16", c_p), ("nrows", i32), ("cols", i32), ("accumulate", i32)]


class ImageDesc(C.Structure):
    _fields_ = [("src", c_p), ("h", i32), ("w", i32), ("pitch", i32), ("ksize_h", i32), ("ksize_v", i32),
                ("kh_off", i32), ("kv_off", i32), ("bh_off", i32), ("bv_off", i32), ("crop_x", i32), ("crop_y", i32)]


# name -> argtypes (restype is int unless listed in _RESTYPES). Must match include/vq3_hip.h.
SIGNATURES = {
    "vq3_abi_version": [],
    "vq3_last_error": [],
    "vq3_target_arch": [],
    "vq3_gemm_bf16_nt": [C.POINTER(GemmDesc), c_p],
    "vq3_gemm_vit_qkv": [C.POINTER(GemmDesc), C.POINTER(VitQkvEpilogue), c_p],
    "vq3_gemm_swiglu_bwd": [C.POINTER(GemmDesc), c_p, c_p, c_p],
    "vq3_gemm_swiglu_fwd": [C.POINTER(GemmDesc), c_p, c_p],
    "vq3_gemm_bf16_nt_ln": [C.POINTER(GemmDesc), C.POINTER(GemmLnFold), c_p],
    "vq3_gemm_vit_qkv_ln": [C.POINTER(GemmDesc), C.POINTER(VitQkvEpilogue), C.POINTER(GemmLnFold), c_p],
    "vq3_rowstats128": [c_p, c_p, i64, i32, c_p],
    "vq3_rmsnorm_fwd": [c_p, c_p, c_p, c_p, i64, i32, i64, i64, f32, c_p],
    "vq3_rmsnorm_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i32, f32, c_p],
    "vq3_rmsnorm_bwd_rows": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i32, i32, c_p],
    "vq3_colsum_f32_to_bf16": [c_p, i32, i32, c_p, i32, c_p],
    "vq3_colsum_multi": [C.POINTER(ColsumJob), i32, c_p],
    "vq3_layernorm_fwd": [c_p, c_p, i32, c_p, c_p, c_p, c_p, i64, i32, f32, c_p],
    "vq3_silu_mul_fwd": [c_p, c_p, i64, i32, c_p],
    "vq3_silu_mul_bwd": [c_p, c_p, c_p, i64, i32, c_p],
    "vq3_transpose_bf16": [c_p, c_p, i32, i32, i32, i64, i64, i32, i32, i32, i64, i64, i64, i64, i64, i64, c_p],
    "vq3_cast": [c_p, c_p, i64, i32, c_p],
    "vq3_f32_to_bf16_acc": [c_p, c_p, i64, i32, c_p],
    "vq3_qwen_qkprep_fwd": [c_p] * 10 + [i32, i32, i32, i32, i32, f32, c_p],
    "vq3_qwen_qkprep_bwd": [c_p] * 13 + [i32, i32, i32, i32, i32, i32, c_p],
    "vq3_softmax_fwd": [c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, i32, c_p],
    "vq3_softmax_bwd": [c_p, c_p, c_p, i32, i32, i32, i32, i32, f32, c_p],
    "vq3_embed_splice_fwd": [c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, c_p],
    "vq3_embed_splice_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, c_p],
    "vq3_gather_rows": [c_p, c_p, c_p, i32, i32, i32, c_p],
    "vq3_scatter_rows": [c_p, c_p, c_p, i32, i32, i32, c_p],
    "vq3_cross_entropy_fwd_bwd": [c_p, c_p, c_p, i32, i32, i32, f32, c_p],
    "vq3_cross_entropy_rows": [c_p, c_p, c_p, c_p, i32, i32, i32, c_p],
    "vq3_im2col_norm": [c_p, c_p, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), c_p],
    "vq3_vit_qkprep": [c_p] * 10 + [i64, i32, i32, i32, i32, i32, i32, i32, i32, f32, c_p],
    "vq3_flash_attn_fwd": [c_p, c_p, c_p, c_p, i32, i32, i32, i32, i64, f32, c_p],
    "vq3_flash_attn_fwd_rows": [c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i64, f32, c_p],
    "vq3_flash_attn_fwd_bounded": [c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i64, f32, f32, c_p],
    "vq3_adamw_step": [c_p, c_p, c_p, c_p, c_p, i64, f32, f32, f32, f32, f32, i32, f32, c_p, f32, c_p],
    "vq3_sumsq": [c_p, i32, i64, c_p, c_p, c_p],
    "vq3_dropout": [c_p, i32, i64, f32, C.c_uint64, C.c_uint64, c_p],
    "vq3_perceiver_xattn_fwd": [c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i64, i64, i64, i64, i32, f32, f32,
                                C.c_uint64, C.c_uint64, c_p],
    "vq3_resample_ksize": [i32, i32],
    "vq3_resample_plan": [i32, i32, c_p, c_p],
    "vq3_resize_crop_u8": [c_p, i32, c_p, c_p, c_p, i32, i32, i32, c_p],
    "vq3_skinny_gemm_bf16": [c_p, c_p, c_p, c_p, c_p, f32, i32, i32, i32, i32, i64, i64, i64, i64, i32, c_p],
    "vq3_skinny_gemm_fp8": [c_p, c_p, c_p, c_p, c_p, c_p, f32, i32, i32, i32, i32, i64, i64, i64, i64, c_p],
    "vq3_qwen_decode_qkprep": [c_p] * 9 + [i32, i32, i32, i32, i32, f32, c_p],
    "vq3_qwen_decode_attn": [c_p] * 5 + [i32, i32, i32, i32, i32, f32, c_p],
    "vq3_greedy_pick": [c_p, i64, c_p, i32, i32, c_p, i32, c_p, c_p, f32, i32, c_p, i32, i64, c_p, c_p],
    "vq3_decode_advance": [c_p, i32, c_p, c_p],
    "vq3_qwen_decode_layers_supported": [i32] * 6,
    "vq3_qwen_decode_layers_workspace_bytes": [],
    "vq3_qwen_decode_layers": [C.POINTER(DecodeLayersDesc), c_p],
    "vq3_quant_fp8_rows": [c_p, i64, i64, i32, c_p, i64, c_p, c_p],
    "vq3_gemm_fp8_nt": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i64, i64, i64, i64, c_p],
    "vq3_quant_fp8_rows_scaled": [c_p, i64, i64, i32, c_p, c_p, i64, c_p, c_p],
    "vq3_transpose_u8": [c_p, c_p, i32, i32, i64, i64, c_p],
    "vq3_gemm_fp8_ex": [c_p, c_p],
    "vq3_qwen_flash_fwd": [c_p] * 6 + [i32, i32, i32, i32, i32, i64, f32, c_p],
    "vq3_qwen_flash_bwd": [c_p] * 11 + [i32, i32, i32, i32, i32, i32, i64, i64, f32, c_p],
    "vq3_gemm_force_config": [i32],
    "vq3_layernorm_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, i32, f32, c_p],
    "vq3_colsum_f32": [c_p, i32, i32, c_p, i32, c_p],
    "vq3_gelu_bwd": [c_p, c_p, c_p, i64, c_p],
    "vq3_gelu_fwd": [c_p, c_p, i64, c_p],
    "vq3_gemm_tile_order": [i32, i32, i32, i32, i32, c_p, c_p, c_p],
    "vq3_gemm_tune_table_load": [C.c_char_p, c_p],
    "vq3_gemm_tune_workspace": [c_p, i64],
    "vq3_gemm_autotune_hold": [i32],
    "vq3_gemm_split_plan": [i32, i32, i32, i32, c_p, c_p, c_p],
    "vq3_gemm_split_status": [c_p, c_p],
    "vq3_gemm_workspace_provider": [c_p],
    "vq3_gemm_split_poll": [c_p, i32],
    "vq3_gemm_split_debug_spin_bound": [i64],
    "vq3_pack_tokens": [c_p, c_p, c_p, c_p, i32, i32, i32, i64, c_p, c_p, c_p, c_p],
}
_RESTYPES = {"vq3_last_error": C.c_char_p, "vq3_target_arch": C.c_char_p, "vq3_qwen_decode_layers_workspace_bytes": i64}

_lib = None


class Vq3Error(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once). Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("VQ3_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise Vq3Error(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The HIP library is required; there is no CPU/PyTorch fallback."
        )
    # torch ships its own libamdhip64; import it first so that this library binds to the SAME HIP runtime instance
    # (device pointers and streams are shared with torch). Loading ours first gives two runtimes in one process.
    import torch  # noqa: F401
    lib = C.CDLL(str(path))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if lib.vq3_abi_version() != 1:
        raise Vq3Error(f"ABI version mismatch: library {lib.vq3_abi_version()} != binding 1")
    # the committed kernel-choice table of the Stage-1 shapes (tools/make_tune_table.sh); VQ3_GEMM_TUNE_TABLE=0 skips it, a path replaces it
    table = os.environ.get("VQ3_GEMM_TUNE_TABLE", str(TUNE_TABLE))
    if table != "0" and Path(table).exists():
        if lib.vq3_gemm_tune_table_load(table.encode(), None) != 0:
            raise Vq3Error(f"cannot read the GEMM tune table {table}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vq3_last_error()
        raise Vq3Error(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
