"""Writes tests/golden/generate_tiny.npz: transformers' own greedy generate() on the tiny Qwen3 of qwen3_tiny.npz
(same weights), called the way the reference's inference scripts call it. Run in the build container only."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests.golden_io import load, meta, weights  # noqa: E402
from tools.make_golden import bf16_bits, tiny_qwen_cfg  # noqa: E402


def main():
    from transformers import Qwen3ForCausalLM
    z = load("qwen3_tiny.npz")
    cfg = tiny_qwen_cfg(320)
    model = Qwen3ForCausalLM(cfg).to(torch.bfloat16).eval()
    sd = weights(z)
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    model.load_state_dict(sd)
    model.generation_config.pad_token_id = None
    H = cfg.hidden_size
    arrays, cases = {}, []
    g = torch.Generator().manual_seed(77)

    def run(name, B, L, pad_left, **kw):
        emb = (torch.randn(B, L, H, generator=g) * 0.5).to(torch.bfloat16)
        mask = torch.ones(B, L, dtype=torch.long)
        for b, p in enumerate(pad_left):
            mask[b, :p] = 0
        with torch.no_grad():
            out = model.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, num_beams=1, **kw)
        arrays[f"{name}:embeds"] = bf16_bits(emb)
        arrays[f"{name}:mask"] = mask.numpy()
        arrays[f"{name}:out"] = out.numpy()
        cases.append({"name": name, "kw": kw})
        print(name, out.tolist())
        return out

    # qa_inference.py:207-216 (repetition_penalty 1.1), no eos hit expected
    free = run("qa", 1, 24, [0], max_new_tokens=24, repetition_penalty=1.1, eos_token_id=319, pad_token_id=318)
    # arkit_inference.py:274-284 (+ no_repeat_ngram_size=4); strong penalty off so n-grams do repeat without the ban
    run("arkit", 1, 17, [0], max_new_tokens=40, repetition_penalty=1.0, no_repeat_ngram_size=2, eos_token_id=319, pad_token_id=318)
    run("arkit4", 1, 17, [0], max_new_tokens=40, repetition_penalty=1.1, no_repeat_ngram_size=4, eos_token_id=319, pad_token_id=318)
    # two rows, left padding; eos = what row 0 emits at step 12 of an unconstrained run -> row 0 finishes and pads
    # while row 1 runs on; then eos = row 1's step-3 token with a fresh prompt pair for the early-stop-of-all case
    gstate = g.get_state()
    free = run("batch_free", 2, 20, [0, 6], max_new_tokens=16, repetition_penalty=1.1, eos_token_id=319, pad_token_id=317)
    g.set_state(gstate)
    run("batch", 2, 20, [0, 6], max_new_tokens=16, repetition_penalty=1.1, eos_token_id=int(free[0, 12]), pad_token_id=317)
    g.set_state(gstate)
    run("batch_all", 2, 20, [0, 6], max_new_tokens=16, repetition_penalty=1.1,
        eos_token_id=[int(free[0, 12]), int(free[1, 3])], pad_token_id=317)
    # input_ids path: the prompt takes part in the penalty and is returned in front
    ids = torch.randint(0, 300, (2, 12), generator=g)
    mask = torch.ones(2, 12, dtype=torch.long)
    with torch.no_grad():
        out = model.generate(input_ids=ids, attention_mask=mask, do_sample=False, num_beams=1, max_new_tokens=12,
                             repetition_penalty=1.3, eos_token_id=319, pad_token_id=318)
    arrays["ids:input_ids"] = ids.numpy()
    arrays["ids:mask"] = mask.numpy()
    arrays["ids:out"] = out.numpy()
    cases.append({"name": "ids", "kw": dict(max_new_tokens=12, repetition_penalty=1.3, eos_token_id=319, pad_token_id=318)})
    print("ids", out.tolist())
    import transformers
    arrays["meta"] = np.frombuffer(json.dumps({"cases": cases, "transformers": transformers.__version__}).encode(), np.uint8)
    np.savez_compressed(ROOT / "tests" / "golden" / "generate_tiny.npz", **arrays)


if __name__ == "__main__":
    main()
