"""Diagnostic (needs the library built with `make -C vggt_qwen3_amd/csrc EXTRA=-DVQ3_DL_STAMPS` after touching decode_layers.hip): wall-clock
stamps of layer 1's phases (workgroups 0 and 37) and of every layer's end. Usage: python tools/diag/decode_layers_stamps.py [layers]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from vggt_qwen3_amd import _lib
from vggt_qwen3_amd.qwen3 import Qwen3Config, Qwen3ForCausalLM

cfg = Qwen3Config.qwen3_4b()
cfg.num_hidden_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tm = Qwen3ForCausalLM(cfg, device="cuda", seed=0)
emb = (torch.randn(1, 200, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
mask = torch.ones(1, 200, dtype=torch.long, device="cuda")
tm.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=6, use_graph=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
lib = _lib.load()
lib.vq3_debug_decode_stamps.argtypes = [C.c_void_p]
assert lib.vq3_debug_decode_stamps(buf) == 0
names = ["wait h", "sync+xprep0", "stream0", "epi0", "arrive+wait G0", "attention", "arrive+wait G1", "xprep2", "stream2", "epi2", "wait G2",
         "xprep3", "stream3", "epi3", "wait G3", "xprep4", "stream4", "epi4", "arrive"]
for g in range(2):
    t = [buf[g * 32 + i] for i in range(20)]
    print("workgroup", (0, 37)[g], "layer time us:", (t[19] - t[0]) / 100.0)
    for i, n in enumerate(names):
        print(f"  {n:16s} {(t[i + 1] - t[i]) / 100.0:7.2f} us")
    a = [buf[g * 32 + i] for i in range(20, 26)]
    print("  attention: q/k/v prep %.2f, scores %.2f, exp %.2f, P.V %.2f, store %.2f" % tuple((a[i + 1] - a[i]) / 100.0 for i in range(5)))

nl = cfg.num_hidden_layers
le = [buf[64 + i] for i in range(nl + 1)]
print("kernel: %.1f us for %d layers; per layer:" % ((le[nl] - le[0]) / 100.0, nl), " ".join("%.1f" % ((le[i + 1] - le[i]) / 100.0) for i in range(nl)))
