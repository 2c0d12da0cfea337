"""Split-K as a batched GEMM (batch = K slice, f32 partial slabs) + one reduce pass, for the M=1200, N=2560 long-K shapes whose
200 tiles of 128x128 under-fill 256 CUs: does 2 x (256x128 tiles) beat the 128x128 ring kernel? Cold weights."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

def timeit(fn, it=20):
    for _ in range(3): fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

for M, N, K in ((1200, 2560, 19456), (1200, 2560, 9728), (1200, 2560, 6144), (1200, 2560, 4096), (6174, 1024, 4096), (768, 4096, 16384)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ncopy = max(2, int(6e8 // (N * K * 2)) + 1)
    ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(ncopy)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = a.float() @ ws[0].float().t()
    ops.gemm_force_config(-3)
    t_auto = timeit(lambda i: ops.linear(a, ws[i % ncopy], out=out))
    line = f"M={M} N={N} K={K}: auto {t_auto:6.1f} us |"
    for S in (2, 3, 4):
        if (K // S) % 64: continue
        P = torch.empty(S, M, N, device="cuda", dtype=torch.float32)
        for cfg in (11, 20, 13):
            ops.gemm_force_config(cfg)
            def run(i):
                w = ws[i % ncopy]
                ops.gemm_raw(a, w, P, M, N, K // S, K, K, N, nb1=S, sA=(K // S, 0), sB=(K // S, 0), sC=(M * N, 0))
            t = timeit(run)
            run(0)
            err = ((P.sum(0) - ref).norm() / ref.norm()).item()
            line += f" S={S} cfg{cfg}: {t:6.1f} us (err {err:.0e})"
    ops.gemm_force_config(-3)
    P = torch.empty(2, M, N, device="cuda", dtype=torch.float32)
    t_red = timeit(lambda i: torch.sum(P, dim=0).to(torch.bfloat16))
    print(line + f" | torch reduce(2) {t_red:5.1f} us", flush=True)
