"""Does the FIRST full-depth tower evaluation of a process differ from the second (same input)? Optionally after running some test files
in the same process first:   python tools/diag/first_call.py [pytest args ...]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
os.chdir(ROOT)
import torch

if len(sys.argv) > 1:
    import pytest
    print("pytest rc", pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", *sys.argv[1:]]), flush=True)

from vggt_qwen3_amd.vggt import VGGT

model = VGGT(img_size=518, patch_size=14, embed_dim=1024, device="cuda", seed=2)
g = torch.Generator().manual_seed(1234)
img = torch.rand(2, 1, 3, 448, 448, generator=g).cuda()
outs = [model.aggregator.forward_head(img, 128).float().cpu() for _ in range(5)]
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("vs last:", " ".join("%.5f" % rel(o, outs[-1]) for o in outs[:-1]), " finite", bool(torch.isfinite(outs[0]).all()), flush=True)
d = (outs[0] - outs[-1]).abs()
if d.max() > 0:
    rows = (d.amax(-1) > 0)
    print("differing token rows per sample:", rows.sum(-1).tolist(), "of", d.shape[1], "| frame half differs:", bool(d[..., :1024].max() > 0),
          "| global half differs:", bool(d[..., 1024:].max() > 0), "| first differing rows:", [r.nonzero().flatten()[:8].tolist() for r in rows], flush=True)
full = [model.aggregator(img)[0][-1].float().cpu() for _ in range(3)]
print("full forward vs last:", " ".join("%.5f" % rel(o, full[-1]) for o in full[:-1]), flush=True)
