import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from vggt_qwen3_amd import ops
BF16 = torch.bfloat16
N = 2200
g = torch.Generator().manual_seed(3)
Q = torch.randn(1, 1, N, 64, generator=g); K = torch.randn(1, 1, N, 64, generator=g); V = torch.randn(1, 1, N, 64, generator=g)
spikes = [(70, 5, 3.0), (150, 5, 3.0), (290, 40, 3.0), (10, 100, 3.0), (64 * 9 + 17, 7, 3.0), (64 * 9 + 49, 7, 4.0), (64 * 20 + 40, 300, 10.0), (64 * 21 + 3, 300, 3.0), (N - 3, 1500, 5.0)]
for j, q, f in spikes:
    K[0, 0, j] = Q[0, 0, q] * f
Q, K, V = Q.to(BF16).cuda(), K.to(BF16).cuda(), V.to(BF16).cuda()
out = ops.flash_attn(Q, K, V).view(1, N, 1, 64).transpose(1, 2).float()
ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float())
err = (out - ref).abs()[0, 0]
rows = err.max(dim=1).values
top = torch.topk(rows, 8)
print("worst rows:", [(int(i), round(float(v), 4)) for v, i in zip(top.values, top.indices)])
print("rows of interest:", {q: round(float(rows[q]), 4) for q in (5, 7, 40, 100, 300, 1500)})
