"""Micro-benchmark of vq3_gemm_bf16_nt on the shapes of the Stage-1 step (M = B*L = 1200 Qwen3 rows, 768 Perceiver
rows, 6174 VGGT rows). Prints TFLOP/s per shape; HIP events on torch's current stream."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops

SHAPES = [
    ("qwen qkv", 1200, 6144, 2560), ("qwen o", 1200, 2560, 4096), ("qwen gate_up", 1200, 19456, 2560),
    ("qwen down", 1200, 2560, 9728), ("qwen wgrad gu", 19456, 2560, 1216), ("qwen wgrad down", 2560, 9728, 1216),
    ("qwen dgrad gu", 1200, 2560, 19456), ("qwen dgrad down", 1200, 9728, 2560), ("qwen dgrad o", 1200, 4096, 2560),
    ("perc ffn1", 768, 16384, 4096), ("perc ffn2", 768, 4096, 16384), ("perc kv", 768, 8192, 4096),
    ("vggt qkv", 6174, 3072, 1024), ("vggt proj", 6174, 1024, 1024), ("vggt fc1", 6174, 4096, 1024),
    ("vggt fc2", 6174, 1024, 4096), ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192),
]


COLD = "--cold" in sys.argv


def main():
    torch.manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        # cold mode: rotate over enough weight copies (> 600 MB) that W always comes from HBM, as inside a training step
        ncopy = max(1, int(6e8 // (N * K * 2)) + 1) if COLD else 1
        ws = [w] + [w.clone() for _ in range(ncopy - 1)]
        for i in range(3):
            ops.linear(a, ws[i % ncopy], out=out)
        torch.cuda.synchronize()
        it = max(20, ncopy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it):
            ops.linear(a, ws[i % ncopy], out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        del ws
        tf = 2.0 * M * N * K / ms / 1e9
        # torch (hipBLASLt) for reference only
        for _ in range(3):
            torch.matmul(a, w.t(), out=out)
        e0.record()
        for _ in range(it):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / it
        print(f"{name:18s} M={M:6d} N={N:6d} K={K:6d}  {ms*1e3:9.1f} us  {tf:7.1f} TF/s   (hipBLASLt {ms2*1e3:8.1f} us {2.0*M*N*K/ms2/1e9:7.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
