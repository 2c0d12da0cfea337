"""Text side of MultiViewCollator restated (collate_multiview.py:46-79): prompt "{q}\\n<image>\\n", labels -100 on
the prompt, truncation to max_length, padding to max(longest, num_vis+geom+64), mask = ids != pad. Integer work:
must be bit-exact against the reference's collator output (tests/golden/vlm_tiny.npz)."""
from __future__ import annotations

import json
from typing import Dict, List

import torch


def collate_text(tokenizer, questions: List[str], answers: List, max_length: int, num_vis_tokens: int = 128,
                 geom_tokens: int = 8) -> Dict[str, torch.Tensor]:
    pad_id = tokenizer.pad_token_id
    min_text_length = num_vis_tokens + geom_tokens + 64
    ids_l, lab_l, max_len = [], [], 0
    for q, a in zip(questions, answers):
        if not isinstance(a, str):
            a = json.dumps(a, ensure_ascii=False)
        p_ids = tokenizer(f"{q}\n<image>\n", add_special_tokens=False)["input_ids"]
        a_ids = tokenizer(a, add_special_tokens=False)["input_ids"]
        ids = (p_ids + a_ids)[:max_length]
        lab = ([-100] * len(p_ids) + a_ids)[:max_length]
        max_len = max(max_len, len(ids))
        ids_l.append(ids)
        lab_l.append(lab)
    max_len = max(max_len, min_text_length)
    for ids, lab in zip(ids_l, lab_l):
        n = max_len - len(ids)
        if n > 0:
            ids += [pad_id] * n
            lab += [-100] * n
    input_ids = torch.tensor(ids_l, dtype=torch.long)
    return {"input_ids": input_ids, "attention_mask": (input_ids != pad_id).long(),
            "labels": torch.tensor(lab_l, dtype=torch.long)}
