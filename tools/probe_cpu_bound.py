"""Is the Stage-1 step CPU-bound? Times the host loop of N micro-steps (returns when everything is ENQUEUED) against the same loop
followed by a device synchronise. enqueue ~= total -> the GPU waits for the host's launches."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, yaml
import bench
from vggt_qwen3_amd.perceiver import PerceiverConfig
from vggt_qwen3_amd.qwen3 import Qwen3Config
from vggt_qwen3_amd.trainer import Stage1Trainer
from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig

dev = torch.device("cuda", 0)
qcfg = Qwen3Config.qwen3_4b()
pcfg = PerceiverConfig(**yaml.safe_load((bench.ROOT / "configs" / "perceiver_small.yaml").read_text()))
vcfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=0, projector_cfg=pcfg,
                            text_config=qcfg, device=str(dev), seed=0)
model = VGGTQwen3VLM(vcfg).train()
N = 16
tr = Stage1Trainer(model, grad_accum=N, max_steps=30000)
batch = bench.synthetic_batch(6, 1, 200, 448, 151936, model.image_id, 151643, 198, 1234, dev, False)
for _ in range(N): tr.micro_step(batch)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(N): tr.micro_step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / N:.2f} ms/micro-step, total {1e3 * (t2 - t0) / N:.2f} ms/micro-step", flush=True)
# pieces: forward only / backward only host time
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(4): tr.micro_step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(25)
