"""What would running two micro-batches of 6 in one pass buy? Qwen3 layer GEMMs (fwd, dgrad, wgrad; cold weights) at M = 1200 twice vs M = 2400 once."""
import sys; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops
def timeit(fn, n=20):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
H, I, Q, QO = 2560, 9728, 6144, 4096
tot = {1200: 0.0, 2400: 0.0}
for M in (1200, 2400):
    rep = 2400 // M
    rows = []
    for name, N, K in [("qkv", Q, H), ("o", H, QO), ("gu", 2 * I, H), ("down", H, I)]:
        nb = 6
        ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nb)]
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        dw = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
        f = timeit(lambda i: ops.linear(x, ws[i % nb])) * rep
        d = timeit(lambda i: ops.gemm_raw(dy, ws[i % nb], dx, M, K, N, N, K, K, transB=True)) * rep
        w = timeit(lambda i: ops.gemm_raw(dy, x, dw, N, K, M, N, K, K, transA=True, transB=True)) * rep
        rows.append((name, f, d, w)); tot[M] += f + d + w
    print("M", M, " ".join(f"{n}: fwd {f:.0f} dgrad {d:.0f} wgrad {w:.0f}" for n, f, d, w in rows), "| sum per 2400 rows", round(tot[M]), "us")
print("fused/unfused", tot[2400] / tot[1200])
