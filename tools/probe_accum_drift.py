"""How far does bf16 gradient accumulation (flat_g, the default) drift from an fp32 accumulator over an accumulation window?
DeepSpeed-bf16, the reference's default engine, accumulates in fp32. Full-size model, N DIFFERENT synthetic micro-batches:
(a) flat_g accumulated in bf16 by the backward itself (what the trainer does), (b) the same N per-micro-batch gradients, each
produced in bf16 with accumulate=False, summed in fp32 on the side. Prints the relative L2 distance per tensor class."""
import importlib.util
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import yaml
from vggt_qwen3_amd.perceiver import PerceiverConfig
from vggt_qwen3_amd.qwen3 import Qwen3Config
from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig

spec = importlib.util.spec_from_file_location("b", ROOT / "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
model = VGGTQwen3VLM(VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=0,
                                          projector_cfg=pcfg, text_config=Qwen3Config.qwen3_4b(), device="cuda", seed=0)).train()
model.projector.eval()
tm = model.text_model
dev = torch.device("cuda")
batches = [bench.synthetic_batch(6, 1, 200, 448, 151936, model.image_id, 151643, 198, 100 + i, dev, False) for i in range(N)]
ref = torch.zeros_like(tm.flat_g, dtype=torch.float32)
for b in batches:
    st = model.forward_state(b["pixel_values"], None, b["input_ids"], b["attention_mask"], b["labels"], True)
    model._backward_text(st, 1.0 / N, accumulate=False)
    ref += tm.flat_g.float()
for i, b in enumerate(batches):
    st = model.forward_state(b["pixel_values"], None, b["input_ids"], b["attention_mask"], b["labels"], True)
    model._backward_text(st, 1.0 / N, accumulate=i > 0)
got = tm.flat_g.float()
def rel(a, b): return ((a - b).norm() / (b.norm() + 1e-30)).item()
print(f"N={N} micro-batches: whole buffer rel L2 {rel(got, ref):.3e}")
for name in ("embed", "l0.qkv", "l0.down", "l17.gu", "l35.o", "l35.ln1", "norm"):
    o, s = tm.table[name]
    n = 1
    for d in s: n *= d
    print(f"  {name:8s} {rel(got[o:o + n], ref[o:o + n]):.3e}")
