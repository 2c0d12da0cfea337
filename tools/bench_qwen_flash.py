"""Micro-benchmark of the fused Qwen3 attention kernels at the Stage-1 shape (B=6, 32 q / 8 kv heads x 128, L=200):
forward and backward (dQ pass + dK/dV pass) per layer, operands rotated over several copies so they do not sit in L2."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from vggt_qwen3_amd import ops

B, L, Hq, Hkv, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 6), 200, 32, 8, 128     # 48 = a merged pass of 8 micro-batches
NC = 12 if B <= 8 else 3
torch.manual_seed(0)
sets = []
for _ in range(NC):
    Q = torch.randn(B, Hq, L, D, device="cuda").to(torch.bfloat16)
    K = torch.randn(B, Hkv, L, D, device="cuda").to(torch.bfloat16)
    V = torch.randn(B, Hkv, L, D, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, Hq * D, device="cuda").to(torch.bfloat16)
    sets.append((Q, K, V, dO))
km = torch.ones(B, L, device="cuda", dtype=torch.uint8)
km[:, 40:] = 0            # the synthetic batches attend to ~20-40 text positions; causal structure is what costs
outs = [ops.qwen_flash_fwd(Q, K, V, km, B, L, Hq, Hkv, D, D ** -0.5) for Q, K, V, _ in sets]
def timeit(fn, it=48):
    for i in range(6): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, mask in (("padded (40 of 200 keys attended)", km), ("dense", torch.ones(B, L, device="cuda", dtype=torch.uint8))):
    for ml in (0, int(mask[0].view(-1).nonzero().max().item()) // 32 + 1):
        tf = timeit(lambda i: ops.qwen_flash_fwd(*sets[i % NC][:3], mask, B, L, Hq, Hkv, D, D ** -0.5, max_live_tiles=ml))
        for parts in (1, 2):
            tb = timeit(lambda i: ops.qwen_flash_bwd(*sets[i % NC][:3], mask, outs[i % NC][0], sets[i % NC][3], outs[i % NC][1], B, L, Hq, Hkv, D, D ** -0.5, kv_parts=parts, max_live_tiles=ml))
            print(f"{name}, max_live_tiles={ml}: fwd {tf:6.1f} us   bwd (dQ + dK/dV, kv_parts={parts}) {tb:6.1f} us   total {tf + tb:6.1f} us/layer", flush=True)
