"""Full-depth tower parity at config C4's shape (one sample, 8 views, 448 x 448: the 8 232-token global attention through all 24 + 24 + 24
blocks) - too slow on the CPU for the test suite (2-4 minutes per oracle evaluation), run once per round and kept as
profiles/r5_depth_parity_c4.json. Same three-way comparison as tests/test_fulldepth_gpu.py: HIP vs the oracle in bf16 (as the reference's
CPU forward runs) vs the oracle in fp32, on the tower tokens the reference consumes and on the Perceiver's visual tokens."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import yaml

from tests.test_fulldepth_gpu import _oracle_vision, relerr
from vggt_qwen3_amd.perceiver import PerceiverConfig
from vggt_qwen3_amd.qwen3 import Qwen3Config
from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig

BF16, F32 = torch.bfloat16, torch.float32
pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
q = Qwen3Config.qwen3_4b(); q.num_hidden_layers = 1            # (the text model is not part of this run)
cfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=0, projector_cfg=pcfg,
                           text_config=q, device="cuda", seed=0)
model = VGGTQwen3VLM(cfg).eval()
g = torch.Generator().manual_seed(4321)
images = torch.rand(1, 8, 3, 448, 448, generator=g)
with torch.no_grad():
    tok = model._vision_tokens(images.cuda()).float().cpu()
    vis = model.encode_images(images.cuda()).float().cpu()
    full = model.vision_model.aggregator(images.cuda())[0][-1].float().cpu()          # every row of the last iterate [1, 8, 1029, 2048]
vsd = {n: t.detach().cpu() for n, t in model.vision_model.aggregator.named_tensors().items()}
psd = {k: v.detach().float().cpu() for k, v in model.projector.state_dict().items()}
emb = model.text_model._w["embed"].detach().cpu()
ids = torch.full((1, 200), 7, dtype=torch.long); ids[0, 3] = model.image_id
rep = {"shape": {"B": 1, "V": 8, "tokens_per_sample": 8232, "tower_blocks": 72}, "cpu_threads": torch.get_num_threads()}
out = {}
for name, td in (("ref16", BF16), ("ref32", F32)):
    t0 = time.perf_counter()
    out[name] = _oracle_vision(images, ids, vsd, psd, emb, model.image_id, pcfg.num_heads, pcfg.num_layers, td, F32, BF16 if td == BF16 else F32)
    rep.setdefault("cpu_seconds", {})[name] = round(time.perf_counter() - t0, 1)
    print(name, rep["cpu_seconds"][name], "s", flush=True)
for key, hip in (("tower_tokens", tok), ("vis_tokens", vis)):
    rep[key] = {"hip_vs_ref16": relerr(hip, out["ref16"][key]), "hip_vs_ref32": relerr(hip, out["ref32"][key]),
                "ref16_vs_ref32": relerr(out["ref16"][key], out["ref32"][key])}
print(json.dumps(rep, indent=1))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r5_depth_parity_c4.json").write_text(json.dumps(rep, indent=1))
ok = all(rep[k]["hip_vs_ref32"] <= 1.25 * rep[k]["ref16_vs_ref32"] + 1e-3 for k in ("tower_tokens", "vis_tokens"))
sys.exit(0 if ok else 1)
