"""Dense C2 forward loss of the full-size model under several GEMM configurations (diagnostic for cfg 25)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import yaml

from vggt_qwen3_amd import ops
from vggt_qwen3_amd.perceiver import PerceiverConfig
from vggt_qwen3_amd.qwen3 import Qwen3Config
from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
import importlib.util
spec = importlib.util.spec_from_file_location("vq3_bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
cfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=8,
                           projector_cfg=pcfg, text_config=Qwen3Config.qwen3_4b(), device="cuda", seed=0)
model = VGGTQwen3VLM(cfg).train()
model.projector.eval()
b = bench.synthetic_batch(6, 1, 200, 448, 151936, model.image_id, 151643, 198, 1234, torch.device("cuda"), True)


def loss(need_grad=False):
    st = model.forward_state(b["pixel_values"], b.get("geom_token"), b["input_ids"], b["attention_mask"], b["labels"], need_grad=need_grad)
    return st["loss"].item()


for cfgid in (-3, 20, 25, 13, -3):
    ops.gemm_force_config(cfgid)
    print("cfg", cfgid, "loss", loss(), loss(True), flush=True)
ops.gemm_force_config(-3)
model.trim_padding = True
print("trimmed", loss(), loss(True))
print("gave up:", ops.gemm_split_gave_up())
