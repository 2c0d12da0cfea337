"""Cold-weight (HBM-resident W, rotated over > 600 MB of copies) timing of the dgrad / wgrad GEMMs in the layouts the
training step uses: dgrad = NN (W k-major), wgrad = TT (dY, X k-major)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

def bench(fn, n):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for name, M, N, K in [("dgrad gu", 1200, 2560, 19456), ("dgrad down", 1200, 9728, 2560), ("dgrad qkv", 1200, 2560, 6144),
                      ("dgrad o", 1200, 4096, 2560)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ncopy = int(6e8 // (N * K * 2)) + 2
    Ws = [torch.randn(K, N, device="cuda").to(torch.bfloat16) for _ in range(ncopy)]     # k-major B = W as stored [K rows, N cols]
    Wts = [w.t().contiguous() for w in Ws]
    outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(8)]
    fl = 2.0 * M * N * K
    t_nn = bench(lambda i: ops.gemm_raw(A, Ws[i % ncopy], outs[i % 8], M, N, K, K, N, N, transB=True), max(20, ncopy))
    t_nt = bench(lambda i: ops.gemm_raw(A, Wts[i % ncopy], outs[i % 8], M, N, K, K, K, N), max(20, ncopy))
    print(f"{name:12s} M={M} N={N} K={K}: cold NN {fl/t_nn/1e9:7.1f} TF/s   cold NT {fl/t_nt/1e9:7.1f} TF/s")
    del Ws, Wts
