"""Race detector at PRODUCTION shapes: the bench's merged pass (10 micro-batches = 60 samples, 12 000 token rows, Qwen3-4B 36 layers,
VGGT-1B, Perceiver; the shipped kernel-choice table) is run N times from the same inputs and weights; the tower tokens, the visual
tokens, the loss vector and the WHOLE flat gradient must come out bit-identical every time (the pipeline sums in fixed orders: any
difference is a race or an uninitialised read).   python tools/soak_repro.py [N] [--c4] [--mb=K] [--fp8] [--trim] [--train-projector]"""
import importlib.util
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import yaml

from vggt_qwen3_amd import ops
from vggt_qwen3_amd.perceiver import PerceiverConfig
from vggt_qwen3_amd.qwen3 import Qwen3Config
from vggt_qwen3_amd.trainer import Stage1Trainer
from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
C4 = "--c4" in sys.argv
spec = importlib.util.spec_from_file_location("vq3_bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
cfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=8 if C4 else 0, projector_cfg=pcfg,
                           text_config=Qwen3Config.qwen3_4b(), device="cuda", seed=0)
cfg.fp8_text_forward = "--fp8" in sys.argv                 # config C5's arithmetic
cfg.trim_padding = "--trim" in sys.argv                    # the exact padding shortcut
cfg.train_projector = "--train-projector" in sys.argv      # the "corrected" mode: the Perceiver's backward runs too
model = VGGTQwen3VLM(cfg).train()
model.projector.eval()                       # (dropout offsets advance per call: the same mask every time needs eval; the kernels are the same)
tr = Stage1Trainer(model, grad_accum=20, max_steps=1000)
nmb = 4 if C4 else 10
for a in sys.argv:
    if a.startswith("--mb="):
        nmb = int(a[5:])
dev = torch.device("cuda")
mbs = [bench.synthetic_batch(6, 8 if C4 else 1, 200, 448, 151936, model.image_id, 151643, 198, 1234 + i, dev, C4) for i in range(nmb)]
big, sizes = tr._merge(mbs)
ref = None
bad = 0
for it in range(N):
    t0 = time.perf_counter()
    with torch.no_grad():
        tok = model._vision_tokens(big["pixel_values"]).clone()
    st = model.forward_state(big["pixel_values"], big.get("geom_token"), big["input_ids"], big["attention_mask"], big["labels"], need_grad=True,
                             loss_groups=sizes)
    model._backward_text(st, 1.0, accumulate=False)
    model.text_model.join_wgrad_stream()
    torch.cuda.synchronize()
    cur = [tok, st["emb"].clone(), st["h_last"].clone(), model.text_model.flat_g.clone()]
    if cfg.train_projector:
        cur.append(torch.cat([p.grad.reshape(-1) for p in model.projector.parameters() if p.grad is not None]).clone())
        for p in model.projector.parameters():
            p.grad = None
    loss = st["loss"].float().cpu()
    del st
    if ref is None:
        ref, loss0 = cur, loss
        assert all(torch.isfinite(t.float()).all() for t in cur), "non-finite values"
        print(f"pass 0: {time.perf_counter() - t0:.2f} s, losses {loss.tolist()[:3]}..., |g| {float(cur[3].float().norm()):.4e}", flush=True)
        continue
    names = ("tower tokens", "inputs_embeds", "h_last", "flat gradient", "projector gradient")
    diffs = [(n, int((a != b).sum()), float((a.float() - b.float()).abs().max())) for n, a, b in zip(names, cur, ref) if not torch.equal(a, b)]
    bad += bool(diffs)
    print(f"pass {it}: {time.perf_counter() - t0:.2f} s, {'IDENTICAL' if not diffs else diffs}, loss max diff {float((loss - loss0).abs().max()):.2e}", flush=True)
print("split-K gave up:", ops.gemm_split_gave_up())
sys.exit(1 if bad else 0)
