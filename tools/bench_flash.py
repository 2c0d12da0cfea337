"""Micro-benchmark of vq3_flash_attn_fwd at the VGGT shapes (frame attention 1029 tokens, global attention 8232 tokens)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

for G, NH, N in [(6, 16, 1029), (48, 16, 1029), (1, 16, 8232), (6, 16, 8232)]:
    Q = torch.randn(G, NH, N, 64, device="cuda").to(torch.bfloat16)
    K = torch.randn_like(Q); V = torch.randn_like(Q)
    for _ in range(3): ops.flash_attn(Q, K, V)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 10
    e0.record()
    for _ in range(it): ops.flash_attn(Q, K, V)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = 4.0 * G * NH * N * N * 64
    print(f"G={G} NH={NH} N={N}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s (V read as stored)", flush=True)
