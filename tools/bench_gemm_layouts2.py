"""A/B of the any-layout kernel's schedules (102: 128x128 two-stage, 103: 128x128 loader ring, 105: 256x128 loader ring) on the
k-major shapes of the Stage-1 backward (wgrad: both operands k-major; down_proj dgrad: k-major B), interleaved rounds."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

SHAPES = [("wgrad gate_up", 19456, 2560, 1200, True, True), ("wgrad down", 2560, 9728, 1200, True, True),
          ("wgrad qkv", 6144, 2560, 1200, True, True), ("wgrad o", 2560, 4096, 1200, True, True),
          ("dgrad down", 1200, 9728, 2560, False, True)]
for name, M, N, K, tA, tB in SHAPES:
    nc = 6
    As = [torch.randn((K, M) if tA else (M, K), device="cuda").to(torch.bfloat16) for _ in range(nc)]
    Bs = [torch.randn((K, N) if tB else (N, K), device="cuda").to(torch.bfloat16) for _ in range(nc)]
    C = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
    res = {c: [] for c in (102, 103, 105)}
    for _ in range(5):
        for cfg in res:
            ops.gemm_force_config(cfg)
            def run(i):
                ops.gemm_raw(As[i % nc], Bs[i % nc], C, M, N, K, M if tA else K, N if tB else K, N, transA=tA, transB=tB, accumulate=tA)
            for i in range(2): run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12): run(i)
            e1.record(); torch.cuda.synchronize()
            res[cfg].append(e0.elapsed_time(e1) / 12)
    ops.gemm_force_config(-3)
    line = f"{name:14s} M={M:6d} N={N:6d} K={K:5d} "
    for cfg, t in res.items():
        m = sorted(t)[len(t) // 2]
        line += f"| {cfg}: {m * 1e3:7.1f} us {2.0 * M * N * K / m / 1e9:7.1f} TF/s "
    print(line, flush=True)
