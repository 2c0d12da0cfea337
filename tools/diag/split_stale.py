"""cfg 25 (last-round K split) against cfg 20 and an fp32 product on FRESH data every launch, element by element in bf16 ulps.
Usage: python tools/diag/split_stale.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from vggt_qwen3_amd import ops

torch.manual_seed(0)
bad = 0
for (M, N, K) in [(1200, 2560, 9728), (1200, 2560, 19456), (9600, 2560, 6144), (1200, 2560, 2560)]:
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    for it in range(4):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16) * (1.0 + it)
        ops.gemm_force_config(20)
        ref = ops.linear(A, W).float()
        ops.gemm_force_config(25)
        out = ops.linear(A, W).float()
        truth = A.float() @ W.float().t()
        ulp = (ref.abs().clamp_min(1e-30)).log2().floor().exp2() * 2.0 ** -7
        d = (out - ref).abs() / ulp
        dt20 = ((ref - truth).abs() / ulp)
        dt25 = ((out - truth).abs() / ulp)
        nbad = int((d > 1.01).sum())
        rows = torch.nonzero((d > 1.01).any(1)).flatten()[:8].tolist()
        print(M, N, K, it, "max ulp diff 25 vs 20: %.2f" % d.max().item(), "n>1ulp:", nbad, "rows", rows,
              "| vs fp32: cfg20 %.3f cfg25 %.3f (max ulps), mean %.4f %.4f" % (dt20.max().item(), dt25.max().item(), dt20.mean().item(), dt25.mean().item()), flush=True)
        bad += nbad > 0
ops.gemm_force_config(-3)
print("gave up:", ops.gemm_split_gave_up())
sys.exit(1 if bad else 0)
