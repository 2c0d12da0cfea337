"""Soak: several hundred Stage-1 micro-steps on one batch; reports throughput per window and allocator high-water marks
(a leak or fragmentation drift shows as a rising `reserved`)."""
import importlib.util
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import yaml
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.trainer import Stage1Trainer
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
    spec = importlib.util.spec_from_file_location("vq3_bench", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
    model = VGGTQwen3VLM(VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128,
                                              projector_cfg=pcfg, text_config=Qwen3Config.qwen3_4b(), device="cuda", seed=0)).train()
    tr = Stage1Trainer(model, grad_accum=32, max_steps=30000)
    batches = [bench.synthetic_batch(6, 1, 200, 448, 151936, model.image_id, 151643, 198, 1234 + i, torch.device("cuda"), False)
               for i in range(4)]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    t0 = time.perf_counter()
    for i in range(n):
        loss = tr.micro_step(batches[i % 4])
        if (i + 1) % 64 == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"steps {i + 1 - 64:4d}-{i + 1:4d}: {64 * 6 / dt:6.1f} samples/s  loss {loss.item():.4f}  allocated "
                  f"{torch.cuda.memory_allocated() / 2**30:6.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB", flush=True)
            t0 = time.perf_counter()


if __name__ == "__main__":
    main()
