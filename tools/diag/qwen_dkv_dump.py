"""Writes dQ / dK / dV of vq3_qwen_flash_bwd for fixed inputs to argv[1] (run once per VQ3_QWEN_DKV_LDS setting and compare)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from vggt_qwen3_amd import ops
out = {}
for (B, L, pad) in ((3, 200, "right"), (2, 77, "left"), (2, 256, "none"), (1, 520, "right")):
    torch.manual_seed(L)
    Hq, Hkv, D = 32, 8, 128
    Q = torch.randn(B, Hq, L, D, device="cuda").to(torch.bfloat16)
    K = torch.randn(B, Hkv, L, D, device="cuda").to(torch.bfloat16)
    V = torch.randn(B, Hkv, L, D, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, Hq * D, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, L, dtype=torch.uint8, device="cuda")
    if pad == "right":
        for b in range(B): mask[b, 30 + 17 * b:] = 0
    elif pad == "left":
        for b in range(B): mask[b, : 2 + 5 * b] = 0
    O, lse = ops.qwen_flash_fwd(Q, K, V, mask, B, L, Hq, Hkv, D, D ** -0.5)
    for parts in (1, 2, 3):
        dQ, dK, dV = ops.qwen_flash_bwd(Q, K, V, mask, O, dO, lse, B, L, Hq, Hkv, D, D ** -0.5, kv_parts=parts)
        out[f"{B}_{L}_{pad}_{parts}"] = (dQ.cpu(), dK.cpu(), dV.cpu())
torch.save(out, sys.argv[1])
print("saved", len(out))
