"""Perceiver cross-attention at the reference's sizes (configs/perceiver_small.yaml: 8 heads x 512, 128 latents, 128 context rows):
the fused launch (csrc/perceiver_attn.hip) against the batched GEMM -> softmax -> batched GEMM route, eval and with dropout.
usage: python tools/bench_perceiver_xattn.py [samples=48]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vggt_qwen3_amd import ops  # noqa: E402
from vggt_qwen3_amd.perceiver import PerceiverConfig, PerceiverProjector  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H, N, T, hd = 8, 128, 128, 512
D = H * hd
torch.manual_seed(0)
q = (torch.randn(B * N, D, device="cuda") * 1.0).to(torch.bfloat16)
kv = (torch.randn(B * T, 2 * D, device="cuda") * 1.0).to(torch.bfloat16)
proj = PerceiverProjector(PerceiverConfig(), 2048, 2560).cuda()
flush = torch.empty(320 * 2 ** 20, dtype=torch.uint8, device="cuda")


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


flop = 4.0 * B * H * N * T * hd
for p in (0.0, 0.1):
    def drop(t):
        if p > 0:
            ops.dropout_(t, p, 1, 0)
        return t
    tf = timed(lambda: ops.perceiver_xattn(q, kv, B, H, N, T, hd, 128, p, 1, 0))
    tk = timed(lambda: ops.perceiver_xattn(q, kv, B, H, N, T, hd, 128, p, 1, 0, keep_p=True))
    t3 = timed(lambda: proj._xattn_three_launches(q, kv, B, T, 128, drop))
    print(f"p_drop {p}: fused {tf:7.1f} us ({flop / tf / 1e6:6.1f} TF/s)   fused + kept P {tk:7.1f} us   three launches {t3:7.1f} us", flush=True)
