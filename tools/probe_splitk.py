"""Does split-K (f32 atomics + zero + cast) pay on the under-filled dgrad shapes (200 tiles on 256 CUs)?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vggt_qwen3_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M, H = 1200, 2560
    for name, N in [("dgrad gate|up (K=19456)", 19456), ("dgrad qkv (K=6144)", 6144), ("dgrad o... fwd-like (K=4096)", 4096),
                    ("dgrad down->H? (K=9728)", 9728)]:
        nb = 4
        ws = [(torch.randn(N, H, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nb)]
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        dx = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
        dx32 = torch.empty(M, H, device="cuda", dtype=torch.float32)
        base = timeit(lambda i: ops.gemm_raw(dy, ws[i % nb], dx, M, H, N, N, H, H, transB=True))
        out = [f"{name}: plain {base:.0f} us"]
        for ks in (2, 3, 5):
            def f(i):
                dx32.zero_()
                ops.gemm_raw(dy, ws[i % nb], dx32, M, H, N, N, H, H, transB=True, ksplit=ks)
                ops.cast(dx32, torch.bfloat16)
            out.append(f"split {ks}: {timeit(f):.0f}")
        print(" | ".join(out))


if __name__ == "__main__":
    main()
