"""Weight-gradient GEMM dW[N,K] (+)= dY^T . X (both operands k-major, as stored) at the Qwen3-4B shapes: one micro-batch's 1200 token
rows with the bf16 read-modify-write of the gradient (what every micro-batch of an accumulation window pays today) against the same
product over n micro-batches' rows at once (contraction n x 1200). Prints us per micro-batch's worth of rows and TF/s."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

def timeit(fn, it=10, rounds=3):
    ts = []
    for _ in range(rounds):
        for i in range(2): fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it): fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it * 1e3)
    return sorted(ts)[len(ts) // 2]

shapes = [("gate_up", 19456, 2560), ("down", 2560, 9728), ("qkv", 6144, 2560), ("o", 2560, 4096)]
tot = {}
for n in (1, 2, 4, 8, 32):
    rows = 1200 * n
    line = f"n={n:2d} (K={rows:5d}):"
    tsum = 0.0
    for name, N, K in shapes:
        ncopy = 3
        dY = [torch.randn(rows, N, device="cuda").to(torch.bfloat16) for _ in range(ncopy)]
        X = [torch.randn(rows, K, device="cuda").to(torch.bfloat16) for _ in range(ncopy)]
        g = [torch.zeros(N, K, device="cuda", dtype=torch.bfloat16) for _ in range(ncopy)]
        def run(i):
            j = i % ncopy
            ops.gemm_raw(dY[j], X[j], g[j], N, K, rows, N, K, K, accumulate=True, transA=True, transB=True)
        t = timeit(run, it=max(3, 12 // n))
        tsum += t / n
        line += f"  {name} {t / n:6.1f} us/mb {2.0 * rows * N * K / t / 1e6:6.1f} TF/s |"
        del dY, X, g
    print(line + f"  sum {tsum:6.1f} us per layer and micro-batch", flush=True)
