"""Cost of the fused epilogues on the short-K VGGT GEMM shapes (M=6174 tokens): plain / bias / bias+GELU /
bias+LayerScale+residual, weights rotated so they are cold."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vggt_qwen3_amd import ops  # noqa: E402

SHAPES = [("vggt qkv", 6174, 3072, 1024), ("vggt fc1", 6174, 4096, 1024), ("vggt proj", 6174, 1024, 1024), ("vggt fc2", 6174, 1024, 4096)]


def timeit(fn, n=30):
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
        nb = 8
        xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(nb)]
        ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nb)]
        bias = torch.randn(N, device="cuda")
        cs = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        fl = 2.0 * M * N * K
        out = []
        for label, kw in [("plain", {}), ("bias", dict(bias=bias)), ("bias+gelu", dict(bias=bias, act=ops.ACT_GELU)),
                          ("bias+ls+res", dict(bias=bias, colscale=cs, residual=res))]:
            t = timeit(lambda i: ops.linear(xs[i % nb], ws[i % nb], **kw))
            out.append(f"{label} {t*1e6:6.1f} us {fl/t/1e12:6.1f} TF/s")
        print(f"{name:10s} " + " | ".join(out))


if __name__ == "__main__":
    main()
