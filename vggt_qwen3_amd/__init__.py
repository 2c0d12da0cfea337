"""MI355X-native (gfx950) hot path of Sycamorers/vggt-qwen3: VGGT -> Perceiver -> Qwen3 forward/backward.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/vq3_hip.h) -> libvq3hip.so
  _lib.py      ctypes binding (fails loudly when the library is missing; no CPU fallback)
  ops.py       tensor-level wrappers (torch is used for device memory and streams only)
"""
__all__ = ["ops"]
