"""q/k-prep kernels of the Qwen3 attention at a merged pass (48 samples x 200 tokens, 32 + 8 + 8 heads x 128): forward and backward,
median of 20 launches each behind a 320 MB flush. usage: python tools/bench_qkprep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

B, L, Hq, Hkv, D = 48, 200, 32, 8, 128
BF16 = torch.bfloat16
torch.manual_seed(0)
qkv = torch.randn(B * L, (Hq + 2 * Hkv) * D, device="cuda").to(BF16)
qw = torch.ones(D, device="cuda", dtype=BF16); kw = torch.ones(D, device="cuda", dtype=BF16)
cos = torch.randn(L, D, device="cuda").to(BF16); sin = torch.randn(L, D, device="cuda").to(BF16)
flush = torch.empty(320 * 2 ** 20, dtype=torch.uint8, device="cuda")


def med(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


Q, K, V, qr, kr = ops.qwen_qkprep_fwd(qkv, qw, kw, cos, sin, B, L, Hq, Hkv, D, 1e-6)
dQ, dK, dV = torch.randn_like(Q), torch.randn_like(K), torch.randn_like(V)
gq = torch.zeros(D, device="cuda", dtype=BF16); gk = torch.zeros(D, device="cuda", dtype=BF16)
tf = med(lambda: ops.qwen_qkprep_fwd(qkv, qw, kw, cos, sin, B, L, Hq, Hkv, D, 1e-6))
tb = med(lambda: ops.qwen_qkprep_bwd(dQ, dK, dV, qkv, qw, kw, cos, sin, qr, kr, gq, gk, False, B, L, Hq, Hkv, D))
mbf = 2 * qkv.numel() * 2 / 1e6
mbb = (dQ.numel() + dK.numel() + dV.numel() + 2 * qkv.numel()) * 2 / 1e6
print(f"qkprep fwd {tf:6.1f} us ({mbf / tf:4.2f} TB/s of {mbf:.0f} MB)   bwd (+ two column sums) {tb:6.1f} us ({mbb / tb:4.2f} TB/s of {mbb:.0f} MB)", flush=True)
