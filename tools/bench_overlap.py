"""Do two independent GEMMs on two HIP streams share the chip? dgrad gate|up (200 tiles: 78 % of the CUs) beside wgrad
gate|up (3040 tiles). Prints serial vs two-stream time."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vggt_qwen3_amd import ops  # noqa: E402


def main():
    M, H, I2 = 1200, 2560, 19456
    dgu = torch.randn(M, I2, device="cuda").to(torch.bfloat16)
    xn = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    W = (torch.randn(I2, H, device="cuda") * 0.02).to(torch.bfloat16)
    dx = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    dW = torch.empty(I2, H, device="cuda", dtype=torch.bfloat16)
    s2 = torch.cuda.Stream()

    def dgrad():   # dX = dY . W   (W k-major B operand)
        ops.gemm_raw(dgu, W, dx, M, H, I2, I2, H, H, transB=True)

    def wgrad():   # dW = dY^T . X
        ops.gemm_raw(dgu, xn, dW, I2, H, M, I2, H, H, transA=True, transB=True)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def both_serial():
        dgrad(); wgrad()

    def both_streams():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            wgrad()
        dgrad()
        torch.cuda.current_stream().wait_stream(s2)

    print(f"dgrad {timeit(dgrad):.1f} us  wgrad {timeit(wgrad):.1f} us  serial {timeit(both_serial):.1f} us  two streams {timeit(both_streams):.1f} us")


if __name__ == "__main__":
    main()
