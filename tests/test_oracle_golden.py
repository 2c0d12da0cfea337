"""Pins the CPU oracle to the golden vectors produced by the reference itself (tools/make_golden.py)."""
import json

import numpy as np
import pytest
import torch

from oracle import collate as ocollate
from oracle import perceiver as operc
from oracle import qwen3 as oq
from oracle import vlm as ovlm
from tests.golden_io import GOLDEN, bf16, load, meta, weights


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _qcfg(m):
    return oq.Qwen3Cfg(hidden_size=m["hidden_size"], num_hidden_layers=m["num_hidden_layers"],
                       num_attention_heads=m["num_attention_heads"], num_key_value_heads=m["num_key_value_heads"],
                       head_dim=m["head_dim"], intermediate_size=m["intermediate_size"],
                       vocab_size=m.get("vocab_size", m.get("vocab")), rms_norm_eps=m["rms_norm_eps"],
                       rope_theta=m["rope_theta"])


def test_qwen3_forward_and_grads_match_hf():
    z = load("qwen3_tiny.npz")
    cfg = _qcfg(meta(z, "config"))
    sd = {k: v.clone().requires_grad_(True) for k, v in weights(z).items()}
    emb = bf16(z["inputs_embeds"]).clone().requires_grad_(True)
    mask, labels = torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["labels"])
    hs = []
    loss, logits = oq.causal_lm(emb, mask, labels, sd, cfg, collect=hs)
    assert abs(loss.item() - float(z["loss"])) < 2e-3 * abs(float(z["loss"]))
    valid = mask.bool()
    assert relerr(logits[valid], bf16(z["logits"])[valid]) < 1e-2
    # per-layer hidden states (HF hidden_states[i+1] is the output of layer i; the last one is post final norm)
    assert relerr(hs[0][valid], bf16(z["hidden_1"])[valid]) < 1e-2
    loss.backward()
    assert relerr(emb.grad, bf16(z["d_inputs_embeds"])) < 2e-2
    for name in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
                 "model.layers.0.self_attn.k_norm.weight", "model.layers.1.input_layernorm.weight",
                 "model.norm.weight", "model.embed_tokens.weight"):
        assert relerr(sd[name].grad, bf16(z["g:" + name])) < 3e-2, name


def test_perceiver_matches_reference():
    z = load("perceiver_tiny.npz")
    m = meta(z, "config")
    sd = {k: v.float() for k, v in weights(z).items()}
    out = operc.projector(torch.from_numpy(z["tokens"]), sd, m["num_heads"], m["num_layers"])
    assert relerr(out, torch.from_numpy(z["out"])) < 1e-5


def test_collate_token_indices_bit_exact():
    from transformers import AutoTokenizer
    z = load("vlm_tiny.npz")
    m = meta(z)
    tok = AutoTokenizer.from_pretrained(str(GOLDEN / "tiny_tokenizer"))
    tok.add_tokens(["<image>"])
    assert tok.convert_tokens_to_ids("<image>") == m["image_id"]
    qs = json.loads(bytes(z["questions"]).decode()); ans = json.loads(bytes(z["answers"]).decode())
    out = ocollate.collate_text(tok, qs, ans, m["max_length"], m["num_vis_tokens"], m["geom_tokens"])
    assert np.array_equal(out["input_ids"].numpy(), z["input_ids"])
    assert np.array_equal(out["attention_mask"].numpy(), z["attention_mask"])
    assert np.array_equal(out["labels"].numpy(), z["labels"])


def test_vlm_forward_matches_reference():
    z = load("vlm_tiny.npz")
    m = meta(z)
    sd = weights(z)
    geom = {k: torch.from_numpy(z["geom:" + k]) for k in ("R", "t", "K", "depth_hist")}
    ids = torch.from_numpy(z["input_ids"])
    out = ovlm.forward(bf16(z["agg"]), geom, ids, torch.from_numpy(z["attention_mask"]),
                       torch.from_numpy(z["labels"]), sd, _qcfg(m), heads=m["num_heads"], num_layers=m["num_layers"],
                       num_vis_tokens=m["num_vis_tokens"], geom_tokens=m["geom_tokens"], image_id=m["image_id"])
    assert relerr(out["vis_tokens"], torch.from_numpy(z["vis_tokens"])) < 1e-5
    assert relerr(out["features"][:, :m["geom_tokens"]], torch.from_numpy(z["geom_feats"])) < 1e-5
    assert torch.equal(out["inputs_embeds"], bf16(z["inputs_embeds"]))
    assert abs(out["loss"].item() - float(z["loss"])) < 2e-3 * abs(float(z["loss"]))
    # integer image of the splice loop
    sm = ovlm.splice_srcmap(ids, m["num_vis_tokens"] + m["geom_tokens"], m["image_id"])
    pos = (ids == m["image_id"]).nonzero()
    for b, p in pos.tolist():
        assert sm[b, p] == 0 and sm[b, p + m["num_vis_tokens"] + m["geom_tokens"] - 1] == m["num_vis_tokens"] + m["geom_tokens"] - 1
    assert (sm >= 0).sum().item() == len(pos) * (m["num_vis_tokens"] + m["geom_tokens"])


def test_splice_overrun_raises_like_reference():
    ids = torch.tensor([[1, 2, 7, 3]])
    with pytest.raises(RuntimeError):
        ovlm.splice(torch.zeros(1, 4, 8), ids, torch.ones(1, 3, 8), image_id=7)
    with pytest.raises(RuntimeError):
        ovlm.splice_srcmap(ids, 3, 7)


def test_preprocess_oracle_matches_pillow_golden_and_live():
    """oracle.preprocess (restated Pillow resample + torchvision size/crop rules) against the committed Pillow outputs
    and, when Pillow is importable, against Pillow itself on fresh random sizes. uint8 / float32: bit-exact."""
    from oracle import preprocess as opre
    z = load("preprocess_tiny.npz")
    for i, (h, w, S) in enumerate(meta(z)["cases"]):
        img = z[f"in{i}"]
        nh, nw = opre.resized_size(h, w, S)
        assert (nh, nw) == z[f"resized{i}"].shape[:2]
        assert np.array_equal(opre.pil_resize_bicubic(img, nh, nw), z[f"resized{i}"]), i
        got = opre.transform(img, S)
        assert got.dtype == np.float32 and np.array_equal(got, z[f"out{i}"]), i
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for _ in range(6):
        h, w, S = int(rng.integers(9, 120)), int(rng.integers(9, 120)), int(rng.integers(8, 64))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nh, nw = opre.resized_size(h, w, S)
        ref = np.asarray(Image.fromarray(a, "RGB").resize((nw, nh), Image.BICUBIC))
        assert np.array_equal(opre.pil_resize_bicubic(a, nh, nw), ref), (h, w, S)


def test_library_resample_plan_matches_oracle():
    """vq3_resample_plan is a host routine (no GPU): its taps and bounds equal the restated Pillow plan bit for bit."""
    from oracle import preprocess as opre
    from vggt_qwen3_amd import _lib
    lib = _lib.load()
    for i, o in [(53, 16), (600, 448), (33, 40), (1000, 448), (448, 448), (4032, 448), (7, 300)]:
        ks = lib.vq3_resample_ksize(i, o)
        b = np.empty((o, 2), np.int32)
        c = np.empty((o, ks), np.int32)
        assert lib.vq3_resample_plan(i, o, b.ctypes.data, c.ctypes.data) == 0
        k2, b2, c2 = opre.precompute_coeffs(i, o)
        assert ks == k2 and np.array_equal(b, b2) and np.array_equal(c, c2), (i, o)
    assert lib.vq3_resample_ksize(0, 5) == -1
    assert lib.vq3_resample_plan(5, 0, b.ctypes.data, c.ctypes.data) != 0


def test_generate_oracle_matches_transformers_golden():
    """oracle.generate (cache-free greedy loop + restated logits processors) reproduces transformers' generate() ids
    exactly on every golden case: single row, n-gram ban, left-padded batch with early finish, multi-eos, input_ids."""
    from oracle import generate as ogen
    z = load("qwen3_tiny.npz")
    sd = weights(z)
    cfg = oq.Qwen3Cfg(**meta(z, "config"))
    g = load("generate_tiny.npz")
    for case in meta(g)["cases"]:
        n, kw = case["name"], dict(case["kw"])
        mask = torch.from_numpy(g[f"{n}:mask"])
        if n == "ids":
            out = ogen.greedy_generate(sd, cfg, None, mask, input_ids=torch.from_numpy(g["ids:input_ids"]), **kw)
        else:
            out = ogen.greedy_generate(sd, cfg, bf16(g[f"{n}:embeds"]), mask, **kw)
        assert np.array_equal(out.numpy(), g[f"{n}:out"]), n


def test_fp8_oracle_known_answers():
    """oracle.fp8 against the OCP e4m3 (fn) encoding itself - the only external fixed point config C5 has (the reference
    holds no fp8 code): known byte patterns, per-row scale rule, round-to-nearest-even, zero rows."""
    from oracle import fp8 as ofp8
    x = torch.tensor([[448.0, 1.0, -2.0, 2.0 ** -6, 2.0 ** -9, 0.0, 17.0, 19.0]])      # amax 448 -> scale exactly 1
    q, s = ofp8.quant_rows(x)
    assert s.tolist() == [1.0]
    assert q.view(torch.uint8)[0].tolist() == [0x7E, 0x38, 0xC0, 0x08, 0x01, 0x00, 0x58, 0x5A]   # 17 -> 16 (tie to even), 19 -> 20
    y = torch.tensor([[0.0, 0.0], [3.0, -1.5]])
    q, s = ofp8.quant_rows(y)
    assert s[0].item() == 1.0 and s[1].item() == np.float32(3.0) / np.float32(448.0)
    assert q.float().tolist() == [[0.0, 0.0], [448.0, -224.0]]
    w = torch.eye(4)[:, [1, 0, 3, 2]] * 2.0
    out = ofp8.linear(torch.arange(8.0).view(2, 4).to(torch.bfloat16), w.to(torch.bfloat16))
    assert torch.allclose(out.float(), torch.tensor([[2.0, 0.0, 6.0, 4.0], [10.0, 8.0, 14.0, 12.0]]), rtol=0.07)


def test_vggt_dino_backbone_matches_transformers_golden():
    """The DINOv2-with-registers stage of oracle.vggt (VGGT's `patch_embed`, one third of the aggregator's FLOPs) against
    transformers' Dinov2WithRegistersModel on the same weights: native grid and an interpolated non-square grid.
    (The frame/global alternating stage has no independent implementation here and stays unpinned.)"""
    from oracle import vggt as ovg
    z = load("dinov2_tiny.npz")
    m = meta(z)
    sd = {k: v.float() for k, v in weights(z).items()}
    mean = torch.tensor(ovg.MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ovg.STD).view(1, 3, 1, 1)
    for name in ("native", "interp"):
        x = (torch.from_numpy(z[f"{name}:images"]) - mean) / std
        t = ovg.dino_tokens(x, sd, patch_size=m["patch"], num_heads=m["num_heads"], dino_depth=m["depth"],
                            num_register_tokens=m["registers"])
        ref = torch.from_numpy(z[f"{name}:tokens"])
        assert t.shape == ref.shape and relerr(t, ref) < 1e-5, name
