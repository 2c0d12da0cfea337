"""Decode throughput of Qwen3-4B (random weights) through text_model.generate(): tokens/s and the fraction of HBM
bandwidth the weight stream achieves. Usage: python tools/bench_decode.py [--batch 1] [--prompt 200] [--new 64]"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch

from vggt_qwen3_amd.qwen3 import Qwen3Config, Qwen3ForCausalLM


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prompt", type=int, default=200)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--layers", type=int, default=36)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    cfg = Qwen3Config.qwen3_4b()
    cfg.num_hidden_layers = a.layers
    tm = Qwen3ForCausalLM(cfg, device="cuda", seed=0)
    emb = (torch.randn(a.batch, a.prompt, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16)
    mask = torch.ones(a.batch, a.prompt, dtype=torch.long, device="cuda")
    kw = dict(inputs_embeds=emb, attention_mask=mask, repetition_penalty=1.1, no_repeat_ngram_size=4,
              use_graph=not a.no_graph)
    tm.generate(max_new_tokens=4, **kw)                       # warm-up (library load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tm.generate(max_new_tokens=2, **kw); torch.cuda.synchronize(); t_short = time.perf_counter() - t0
    t0 = time.perf_counter(); out, stats = tm.generate(max_new_tokens=a.new, return_stats=True, **kw); torch.cuda.synchronize(); t_long = time.perf_counter() - t0
    per_tok = (t_long - t_short) / (a.new - 2)
    wbytes = 2 * sum(p.numel() for n, p in tm.named_parameters() if n != "lm_head.weight")   # every weight once (embedding = lm_head)
    print(json.dumps({"batch": a.batch, "prompt": a.prompt, "new_tokens": int(out.shape[1]), "graph": not a.no_graph,
                      "persistent_layers_kernel": stats["persistent"], "ms_per_token": per_tok * 1e3, "tokens_per_s": a.batch / per_tok,
                      "prefill_plus_2_ms": t_short * 1e3, "weight_GB_per_token": wbytes / 1e9,
                      "weight_stream_GBps": wbytes / per_tok / 1e9, "hbm_peak_GBps": 8000,
                      "frac_of_peak": wbytes / per_tok / 8e12}))


if __name__ == "__main__":
    main()
