"""The two wide-N GEMMs of a merged text pass (12 000 token rows) with their fused SwiGLU epilogues, operands cold (rotating copies larger
than the 256 MB Infinity Cache), HIP events: gate|up forward [12000 x 19456 x 2560] (vq3_gemm_swiglu_fwd: writes gate|up and act) and
the down-projection dgrad [12000 x 9728 x 2560] with the SwiGLU backward in its epilogue (vq3_gemm_swiglu_bwd: reads gate|up, writes
d(gate|up)). Used by tools/sweep_wide_order.sh with VQ3_GEMM_XM / VQ3_GEMM_BAND (the tile order is read once per process)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops

BF16 = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
if len(sys.argv) > 2:
    ops.gemm_force_config(int(sys.argv[2]))      # (e.g. 25: the last round's tiles split along K)
H, I = 2560, 9728
NCOPY = 4


def timed(fn, n=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.manual_seed(0)
    xs = [torch.randn(M, H, device="cuda").to(BF16) for _ in range(NCOPY)]
    wgu = [(torch.randn(2 * I, H, device="cuda") * 0.02).to(BF16) for _ in range(NCOPY)]
    wdT = [(torch.randn(I, H, device="cuda") * 0.02).to(BF16) for _ in range(NCOPY)]       # W_down^T [I, H]: the NT dgrad operand
    gus = [torch.empty(M, 2 * I, device="cuda", dtype=BF16) for _ in range(NCOPY)]
    acts = [torch.empty(M, I, device="cuda", dtype=BF16) for _ in range(NCOPY)]
    dgus = [torch.empty(M, 2 * I, device="cuda", dtype=BF16) for _ in range(NCOPY)]
    us = timed(lambda i: ops.gemm_swiglu_fwd(xs[i % NCOPY], wgu[i % NCOPY], gu_out=gus[i % NCOPY], act_out=acts[i % NCOPY]))
    print(f"swiglu_fwd gate|up {M}x{2 * I}x{H}: {us:8.1f} us {2.0 * M * 2 * I * H / us / 1e6:7.1f} TF/s", flush=True)
    us = timed(lambda i: ops.gemm_swiglu_bwd(xs[i % NCOPY], wdT[i % NCOPY], gus[i % NCOPY], transB=False, out=dgus[i % NCOPY]))
    print(f"swiglu_bwd dgrad   {M}x{I}x{H}: {us:8.1f} us {2.0 * M * I * H / us / 1e6:7.1f} TF/s", flush=True)
    o = torch.empty(M, H, device="cuda", dtype=BF16)
    wd = [(torch.randn(H, I, device="cuda") * 0.02).to(BF16) for _ in range(NCOPY)]
    us = timed(lambda i: ops.linear(acts[i % NCOPY], wd[i % NCOPY], out=o))
    print(f"down fwd           {M}x{H}x{I}: {us:8.1f} us {2.0 * M * I * H / us / 1e6:7.1f} TF/s", flush=True)
    qkvw = [(torch.randn(6144, H, device="cuda") * 0.02).to(BF16) for _ in range(NCOPY)]
    oq = torch.empty(M, 6144, device="cuda", dtype=BF16)
    us = timed(lambda i: ops.linear(xs[i % NCOPY], qkvw[i % NCOPY], out=oq))
    print(f"qkv fwd            {M}x6144x{H}: {us:8.1f} us {2.0 * M * 6144 * H / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
