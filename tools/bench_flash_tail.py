"""VGGT flash attention at 48 x 16 pairs x 1029 tokens: median of 40 launches (each behind a 320 MB flush) for the tail mode in
VQ3_FLASH_TAIL_MODE (0 fifth block / 1 side-stream launch / 2 in-launch key-split workgroup)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

G, NH, N = 48, 16, 1029
Q = torch.randn(G, NH, N, 64, device="cuda").to(torch.bfloat16)
K = torch.randn_like(Q); V = torch.randn_like(Q)
flush = torch.empty(320 * 2 ** 20, dtype=torch.uint8, device="cuda")
for _ in range(5):
    ops.flash_attn(Q, K, V)
ts = []
for _ in range(40):
    flush.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.flash_attn(Q, K, V); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"mode {os.environ.get('VQ3_FLASH_TAIL_MODE', 'default')}: median {ts[20]:.1f} us  min {ts[0]:.1f}  p90 {ts[36]:.1f}", flush=True)
