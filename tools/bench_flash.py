"""Micro-benchmark of vq3_flash_attn_fwd at the VGGT shapes: frame attention (1029 tokens per frame; 60 x 16 pairs = one merged pass of
10 micro-batches) and global attention at 8 views (8232 tokens; 48 x 16 pairs = the C4 pass). Two operand sets alternate so that a
launch's rows do not come out of the Infinity Cache the previous launch of the SAME rows filled. VQ3_FLASH_XCD=0/1 selects the workgroup
placement (read once per process). Every shape is timed twice: the general kernels, and with a score bound promised (the kernels
without a running maximum, vq3_flash_attn_fwd_bounded; the operands are unit normal: |q . k| / 8 stays far below the 24 promised).
    python tools/bench_flash.py [small]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os
import torch
from vggt_qwen3_amd import ops
SAME = os.environ.get('FLASH_BENCH_SAME') == '1'      # one operand set only (rows stay in the caches): how much of the time is memory latency

shapes = [(6, 16, 1029), (60, 16, 1029), (1, 16, 8232), (6, 16, 8232), (48, 16, 8232)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    shapes = shapes[:4]
for G, NH, N in shapes:
    sets = []
    for _ in range(2):
        Q = torch.randn(G, NH, N, 64, device="cuda").to(torch.bfloat16)
        sets.append((Q, torch.randn_like(Q), torch.randn_like(Q)))
    out = torch.empty(G * N, NH * 64, device="cuda", dtype=torch.bfloat16)
    fl = 4.0 * G * NH * N * N * 64
    line = f"G={G} NH={NH} N={N}:"
    for bound in (None, 24.0):
        for i in range(3):
            ops.flash_attn(*sets[0 if SAME else i % 2], out=out, score_bound=bound)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10 if N < 4096 or G < 48 else 4
        e0.record()
        for i in range(it):
            ops.flash_attn(*sets[0 if SAME else i % 2], out=out, score_bound=bound)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        line += f" {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s" + ("   | bounded:" if bound is None else "")
    print(line, flush=True)
    del sets, out
