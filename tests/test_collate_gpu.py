"""Device-side batch builder against the oracle and the reference-generated goldens.
Byte / integer work: every comparison is bit-exact."""
import json

import numpy as np
import pytest
import torch

from tests.golden_io import GOLDEN, load, meta

pytestmark = pytest.mark.gpu


def _rand_img(rng, h, w):
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[h // 4: h // 2, w // 3: w // 2] = 255          # hard edges: bicubic overshoot must clip like Pillow's
    a[h // 2:, : w // 5] = 0
    return a


def test_resize_crop_matches_pillow_golden():
    from vggt_qwen3_amd.collate import preprocess_images
    z = load("preprocess_tiny.npz")
    for i, (h, w, S) in enumerate(meta(z)["cases"]):
        out = preprocess_images([z[f"in{i}"]], S)
        assert out.shape == (1, 3, S, S) and out.dtype == torch.float32
        assert np.array_equal(out[0].cpu().numpy(), z[f"out{i}"]), (i, h, w, S)


@pytest.mark.parametrize("S", [56, 448])
def test_resize_crop_mixed_batch_vs_oracle(S):
    """One launch over images of different sizes: portrait, landscape, square, already-at-size (identity taps),
    upscaled, odd crop offsets (round-half-even), strong shrink (tile height drops to fit LDS)."""
    from oracle import preprocess as opre
    from vggt_qwen3_amd.collate import preprocess_images
    rng = np.random.default_rng(S)
    sizes = [(S, S), (S, S + 1), (S + 3, S), (2 * S + 5, 3 * S + 1), (S // 2 + 1, S // 3 + 2), (S + 2, 5 * S),
             (9 * S + 4, 7 * S + 3), (S, 2 * S + 3)]
    if S == 56:
        sizes.append((40 * S, 33 * S + 7))
    imgs = [_rand_img(rng, h, w) for h, w in sizes]
    out = preprocess_images(imgs, S).cpu().numpy()
    for i, a in enumerate(imgs):
        assert np.array_equal(out[i], opre.transform(a, S)), sizes[i]


def test_resize_crop_rejects_bad_input():
    from vggt_qwen3_amd import _lib
    from vggt_qwen3_amd.collate import preprocess_images
    with pytest.raises(ValueError):
        preprocess_images([], 16)
    with pytest.raises(ValueError):
        preprocess_images([np.zeros((4, 4), np.uint8)], 16)
    with pytest.raises(ValueError):
        preprocess_images([np.zeros((4, 4, 3), np.float32)], 16)
    lib = _lib.load()
    assert lib.vq3_resize_crop_u8(None, 1, None, None, None, 16, 16, 8, None) != 0
    assert b"null" in lib.vq3_last_error()


def test_pack_tokens_vs_oracle_and_reference_golden():
    from transformers import AutoTokenizer
    from oracle import collate as ocollate
    from vggt_qwen3_amd.collate import pack_tokens
    z = load("vlm_tiny.npz")
    m = meta(z)
    tok = AutoTokenizer.from_pretrained(str(GOLDEN / "tiny_tokenizer"))
    tok.add_tokens(["<image>"])
    qs = json.loads(bytes(z["questions"]).decode())
    ans = json.loads(bytes(z["answers"]).decode())
    p = [tok(f"{q}\n<image>\n", add_special_tokens=False)["input_ids"] for q in qs]
    a = [tok(x if isinstance(x, str) else json.dumps(x, ensure_ascii=False), add_special_tokens=False)["input_ids"] for x in ans]
    out = pack_tokens(p, a, m["max_length"], m["num_vis_tokens"] + m["geom_tokens"] + 64, tok.pad_token_id)
    for k in ("input_ids", "attention_mask", "labels"):
        assert out[k].dtype == torch.int64 and np.array_equal(out[k].cpu().numpy(), z[k]), k
    # ragged / edge rows against the oracle: empty answer, empty prompt, truncation inside prompt and inside answer,
    # a pad id occurring inside the text (mask follows the id, like the reference)
    class Tok:
        pad_token_id = 3
        def __call__(self, s, add_special_tokens=False):
            return {"input_ids": [int(t) for t in s.split()]}
    tk = Tok()
    rows_q = ["5 6 7", "", "1 2 3 4 5 6 7 8 9 10 11 12", "4 3 4", "9"]
    rows_a = ["", "8 8", "13 14", "3 7", "1 2 3 4 5 6 7 8 9 10 11 12 13 14"]
    for max_len, floor in [(10, 4), (10, 16), (64, 1)]:
        ref = ocollate_text_raw(tk, rows_q, rows_a, max_len, floor)
        pq = [tk(q)["input_ids"] for q in rows_q]
        pa = [tk(x)["input_ids"] for x in rows_a]
        got = pack_tokens(pq, pa, max_len, floor, tk.pad_token_id)
        for k in ref:
            assert np.array_equal(got[k].cpu().numpy(), ref[k]), (k, max_len, floor)


def ocollate_text_raw(tok, prompts, answers, max_length, floor):
    """collate_multiview.py:56-79 on already-built prompt strings (pure-Python restatement for tiny cases)."""
    ids_l, lab_l, L = [], [], 0
    for p, a in zip(prompts, answers):
        pi, ai = tok(p)["input_ids"], tok(a)["input_ids"]
        ids = (pi + ai)[:max_length]
        lab = ([-100] * len(pi) + ai)[:max_length]
        L = max(L, len(ids))
        ids_l.append(ids)
        lab_l.append(lab)
    L = max(L, floor)
    for ids, lab in zip(ids_l, lab_l):
        n = L - len(ids)
        ids += [tok.pad_token_id] * n
        lab += [-100] * n
    ids = np.array(ids_l, dtype=np.int64).reshape(len(ids_l), L)
    return {"input_ids": ids, "attention_mask": (ids != tok.pad_token_id).astype(np.int64),
            "labels": np.array(lab_l, dtype=np.int64).reshape(len(lab_l), L)}


def test_multiview_collator_end_to_end():
    """The reference's collator call, on PIL images: same dict keys, shapes, dtypes and values as
    (oracle image transform, reference-golden token layout, reference geometry stacking rules)."""
    Image = pytest.importorskip("PIL.Image")
    from transformers import AutoTokenizer
    from oracle import preprocess as opre
    from src.dataio.collate_multiview import MultiViewCollator
    z = load("vlm_tiny.npz")
    m = meta(z)
    tok = AutoTokenizer.from_pretrained(str(GOLDEN / "tiny_tokenizer"))
    tok.add_tokens(["<image>"])
    qs = json.loads(bytes(z["questions"]).decode())
    ans = json.loads(bytes(z["answers"]).decode())
    rng = np.random.default_rng(3)
    S, V = 28, 2
    batch, raw = [], []
    for i, (q, a) in enumerate(zip(qs, ans)):
        views = [_rand_img(rng, 30 + 7 * i + v, 41 + 3 * v) for v in range(V)]
        raw.append(views)
        g = None if i == 1 else {"R": np.eye(3).reshape(1, 9).repeat(V, 0).tolist(), "t": [[0.1 * i, 0, 1]] * V}
        batch.append({"images": [Image.fromarray(x, "RGB") for x in views], "question": q, "answer": a, "geom_token": g})
    col = MultiViewCollator(S, tok, m["max_length"], num_vis_tokens=m["num_vis_tokens"], geom_tokens=m["geom_tokens"])
    out = col(batch)
    assert set(out) == {"pixel_values", "geom_token", "input_ids", "attention_mask", "labels"}
    B = len(batch)
    assert out["pixel_values"].shape == (B, V, 3, S, S) and out["pixel_values"].is_cuda
    for b in range(B):
        for v in range(V):
            assert np.array_equal(out["pixel_values"][b, v].cpu().numpy(), opre.transform(raw[b][v], S))
    for k in ("input_ids", "attention_mask", "labels"):
        assert np.array_equal(out[k].cpu().numpy(), z[k]), k
    gt = out["geom_token"]
    assert gt["mask"].tolist() == [i != 1 for i in range(B)]
    assert gt["R"].shape == (B, V, 9) and gt["t"].shape == (B, V, 3) and gt["R"].dtype == torch.float32
    assert torch.all(gt["R"][1] == 0) and torch.all(gt["t"][1] == 0)
    assert abs(gt["t"][2, 0, 0].item() - 0.2) < 1e-7
    # no geometry anywhere -> None, like the reference
    for s in batch:
        s["geom_token"] = None
    assert col(batch)["geom_token"] is None
    # the transform alone keeps the reference's per-image contract
    one = col.transform(batch[0]["images"][0])
    assert one.shape == (3, S, S) and np.array_equal(one.cpu().numpy(), opre.transform(raw[0][0], S))
