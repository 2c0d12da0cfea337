"""The VGGT tower's K = 1024 GEMMs (60 samples x 1029 tokens per pass) with and without the last-round K split (cfg 25 vs cfg 20):
fc1 + LayerNorm fold + GELU (61740 x 4096 x 1024: 3872 tiles = 15 rounds + 32 tiles) and fused q|k|v + LayerNorm fold + RoPE
(61740 x 3072 x 1024: 2904 tiles = 11 rounds + 88 tiles). Usage: python tools/bench_tower_split.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops

BF16, F32 = torch.bfloat16, torch.float32
torch.manual_seed(0)
S, P = 60, 1029
M, C = S * P, 1024
x = (torch.randn(M, C, device="cuda") * 0.5).to(BF16)
st = ops.rowstats128(x)
flush = torch.empty(320 << 20, device="cuda", dtype=torch.uint8)


def timed(fn, n=12):
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


# fc1
W1 = (torch.randn(4096, C, device="cuda") * 0.03).to(BF16)
cs1 = W1.float().sum(1).contiguous(); b1 = torch.randn(4096, device="cuda") * 0.1
fc1 = lambda: ops.linear(x, W1, bias=b1, act=ops.ACT_GELU, ln_fold=ops.ln_fold(stats_in=st, eps=1e-5, colsum=cs1))
# q|k|v
NH, Wp, ps = 16, 32, 5
Wq = (torch.randn(3 * NH * 64, C, device="cuda") * 0.03).to(BF16)
bq = torch.randn(3 * NH * 64, device="cuda") * 0.1
csq = Wq.float().sum(1).contiguous()
qn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1)
ang = torch.rand(40, 16, device="cuda") * 3.0
emb = torch.cat([ang, ang], -1)
cos, sin = emb.cos().to(BF16).contiguous(), emb.sin().to(BF16).contiguous()
qkv = lambda: ops.linear_vit_qkv(x, Wq, bq, P, NH, qn=qn, kn=qn, cos=cos, sin=sin, tokens_per_frame=P, patch_start=ps, Wp=Wp, eps=1e-5,
                                 ln_fold=ops.ln_fold(stats_in=st, eps=1e-5, colsum=csq))
for name, fn, N in (("fc1", fc1, 4096), ("qkv", qkv, 3072)):
    res = {}
    for cfg in (20, 25, 20, 25):
        ops.gemm_force_config(cfg)
        fn(); torch.cuda.synchronize()
        res.setdefault(cfg, []).append(timed(fn))
    ops.gemm_force_config(-3)
    fl = 2.0 * M * N * C
    print(name, "plan", ops.gemm_split_plan(M, N, C), {k: ["%.1f us = %.0f TF/s" % (t, fl / t / 1e6) for t in v] for k, v in res.items()})
print("gave up:", ops.gemm_split_gave_up())
