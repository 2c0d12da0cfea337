"""fp8 vs bf16 GEMM on the Qwen3-4B forward shapes (cold weights: a different weight buffer every launch)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vggt_qwen3_amd import ops  # noqa: E402

SHAPES = [("qkv", 1200, 6144, 2560), ("o", 1200, 2560, 4096), ("gate_up", 1200, 19456, 2560), ("down", 1200, 2560, 9728),
          ("qkv x8", 9600, 6144, 2560), ("o x8", 9600, 2560, 4096), ("gate_up x8", 9600, 19456, 2560), ("down x8", 9600, 2560, 9728),
          ("dgrad qkv x8", 9600, 2560, 6144), ("dgrad gu x8", 9600, 2560, 19456), ("dgrad down x8", 9600, 9728, 2560),
          ("square4096", 4096, 4096, 4096), ("square8192", 8192, 8192, 8192)]
if len(sys.argv) > 1:
    SHAPES = [s_ for s_ in SHAPES if any(a in s_[0] for a in sys.argv[1:])]


def timeit(fn, n):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    for name, M, N, K in SHAPES:
        nb = max(2, min(16, int(2e9 // (N * K * 2))))
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nb)]
        q = [ops.quant_fp8_rows(w) for w in ws]
        xq, xs = ops.quant_fp8_rows(x)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t_bf = timeit(lambda i: ops.linear(x, ws[i % nb]), 40)
        t_f8 = timeit(lambda i: ops.gemm_fp8(xq, xs, q[i % nb][0], q[i % nb][1], out=out), 40)
        t_q = timeit(lambda i: ops.quant_fp8_rows(x), 40)
        fl = 2.0 * M * N * K
        print(f"{name:12s} M={M:5d} N={N:6d} K={K:5d}  bf16 {t_bf*1e6:7.1f} us {fl/t_bf/1e12:7.1f} TF/s | fp8 {t_f8*1e6:7.1f} us "
              f"{fl/t_f8/1e12:7.1f} TF/s | act quant {t_q*1e6:6.1f} us")


if __name__ == "__main__":
    main()
