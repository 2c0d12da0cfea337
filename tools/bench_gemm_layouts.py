"""v3 GEMM (k-major operands) vs the NT kernel on the Qwen3 dgrad / wgrad shapes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

def bench(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

for name, M, N, K in [("wgrad gu", 19456, 2560, 1200), ("wgrad down", 2560, 9728, 1200), ("wgrad qkv", 6144, 2560, 1200),
                      ("dgrad gu", 1200, 2560, 19456), ("dgrad down", 1200, 9728, 2560), ("dgrad qkv", 1200, 2560, 6144),
                      ("dgrad o", 1200, 4096, 2560)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); Bm = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    At, Bt = A.t().contiguous(), Bm.t().contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    Kp = (K + 63) // 64 * 64
    Ap = torch.zeros(M, Kp, device="cuda", dtype=torch.bfloat16); Ap[:, :K] = A
    Bp = torch.zeros(N, Kp, device="cuda", dtype=torch.bfloat16); Bp[:, :K] = Bm
    fl = 2.0 * M * N * K
    t_nt = bench(lambda: ops.gemm_raw(Ap, Bp, out, M, N, Kp, Kp, Kp, N))
    t_nn = bench(lambda: ops.gemm_raw(A, Bt, out, M, N, K, K, N, N, transB=True))
    t_tt = bench(lambda: ops.gemm_raw(At, Bt, out, M, N, K, M, N, N, transA=True, transB=True))
    print(f"{name:12s} M={M:6d} N={N:6d} K={K:6d}  NT {fl/t_nt/1e9:7.1f}  NN(transB) {fl/t_nn/1e9:7.1f}  TT {fl/t_tt/1e9:7.1f} TF/s")

print("--- padded leading dimension for the k-major B (partition-camping probe)")
for name, M, N, K in [("dgrad gu", 1200, 2560, 19456), ("dgrad qkv", 1200, 2560, 6144), ("dgrad down", 1200, 9728, 2560)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for pad in (0, 8, 64, 136):
        Btp = torch.randn(K, N + pad, device="cuda").to(torch.bfloat16)
        t = bench(lambda: ops.gemm_raw(A, Btp, out, M, N, K, K, N + pad, N, transB=True))
        print(f"{name:12s} ldb = N+{pad:3d}: {fl/t/1e9:7.1f} TF/s")
