"""Checkpoint compatibility on the resident HIP model: the reference's key space in both
directions, the reference loader's own call sequence, and trainer resume."""
import numpy as np
import pytest
import torch

from tests.golden_io import load, meta, weights
from tests.test_parity_gpu import _build_vlm

pytestmark = pytest.mark.gpu


def _batch(z):
    geom = {k: torch.from_numpy(z["geom:" + k]).cuda() for k in ("R", "t", "K", "depth_hist")}
    return {"pixel_values": torch.from_numpy(z["pixel_values"].astype(np.float32)).cuda(), "geom_token": geom,
            "input_ids": torch.from_numpy(z["input_ids"]).cuda(),
            "attention_mask": torch.from_numpy(z["attention_mask"]).cuda(),
            "labels": torch.from_numpy(z["labels"]).cuda()}


def _loss(model, b):
    model.eval()
    with torch.no_grad():
        return model(images=b["pixel_values"], geom_token=b["geom_token"], input_ids=b["input_ids"],
                     attention_mask=b["attention_mask"], labels=b["labels"]).item()


def test_reference_written_checkpoint_loads_by_name(tmp_path):
    """A flat .bin with the REFERENCE's state_dict (names + fp32/bf16 tensors as the golden run saved them) goes through
    the reference's search order into a differently initialised model and reproduces the reference's loss."""
    from vggt_qwen3_amd.checkpoint import load_checkpoint_if_available
    z = load("vlm_tiny.npz")
    m = meta(z)
    sd = {k: v.float() for k, v in weights(z).items()}
    torch.save(sd, tmp_path / "pytorch_model.bin")
    model = _build_vlm(z, m)
    with torch.no_grad():                       # scramble, then restore from the file
        model.text_model.flat_w.mul_(0.5)
        for p in model.projector.parameters():
            p.add_(0.1)
    model.to("cpu")                             # what the reference loader does first: a no-op here
    assert model.text_model.flat_w.is_cuda
    rep = load_checkpoint_if_available(model, str(tmp_path), verbose=False)
    model.to("cuda")
    assert rep is not None and not [u for u in rep["unexpected"] if not u.startswith("vision_model")]
    assert not [k for k in rep["missing"] if not k.startswith("vision_model")], rep["missing"][:5]
    ref = float(z["loss"])
    assert abs(_loss(model, _batch(z)) - ref) < 5e-3 * abs(ref)


def test_sharded_round_trip_and_torch_loader(tmp_path):
    """save_model -> (a) our streaming loader, (b) plain `load_state_dict(torch.load(shard), strict=False)` per shard,
    which is what transformers' load_sharded_checkpoint does for the reference (arkit_inference.py:93)."""
    import json
    from vggt_qwen3_amd.checkpoint import INDEX_NAME, MERGED_DIR, load_checkpoint_if_available, save_model
    z = load("vlm_tiny.npz")
    m = meta(z)
    src = _build_vlm(z, m)
    wm = save_model(src, tmp_path, max_shard_bytes=64 << 10)
    root = tmp_path / MERGED_DIR
    idx = json.loads((root / INDEX_NAME).read_text())
    assert idx["weight_map"] == wm and len(set(wm.values())) > 1
    assert "text_model.lm_head.weight" not in wm and "text_model.model.embed_tokens.weight" in wm
    want = {k: v.clone() for k, v in src.state_dict().items() if not k.startswith("vision_model")}
    for how in ("ours", "torch"):
        dst = _build_vlm(z, m)
        with torch.no_grad():
            dst.text_model.flat_w.zero_()
            for p in list(dst.projector.parameters()) + list(dst.geom_head.parameters()):
                p.zero_()
        if how == "ours":
            assert load_checkpoint_if_available(dst, str(tmp_path), verbose=False) is not None
        else:
            for f in sorted(set(wm.values())):
                dst.load_state_dict(torch.load(root / f, map_location="cpu"), strict=False)
        got = dst.state_dict()
        for k, v in want.items():
            assert torch.equal(got[k], v), (how, k)   # bf16 -> fp32 -> bf16 and fp32 -> fp32 are both exact
        # tied head follows the embedding
        assert got["text_model.lm_head.weight"].data_ptr() == got["text_model.model.embed_tokens.weight"].data_ptr()


def test_trainer_resume(tmp_path):
    """State restore is exact; the continued run follows the uninterrupted one (the backward uses f32 atomics in the
    split-K and embedding reductions, so two runs agree to rounding, not bit for bit)."""
    from vggt_qwen3_amd.checkpoint import load_trainer_state, save_trainer_state
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z = load("vlm_tiny.npz")
    m = meta(z)
    b = _batch(z)
    # eps well above the bf16 noise of the gradients, so that the two runs' updates can be compared element by element
    kw = dict(lr=1e-3, proj_lr=1e-3, weight_decay=0.1, warmup_ratio=0.1, max_steps=20, grad_accum=2, eps=1e-3)
    a = _build_vlm(z, m).train()
    ta = Stage1Trainer(a, **kw)
    for _ in range(4):
        ta.micro_step(b)
    save_trainer_state(ta, tmp_path)
    snap = {k: getattr(ta, k).clone() for k in ("master", "m", "v", "geom_master", "geom_m", "geom_v")}
    w_at_save = a.text_model.flat_w.clone()
    c = _build_vlm(z, m).train()
    tc = Stage1Trainer(c, **kw)
    load_trainer_state(tc, tmp_path)
    assert (tc.micro, tc.opt_step) == (4, 2)
    for k, v in snap.items():
        assert torch.equal(getattr(tc, k), v), k
    assert torch.equal(c.text_model.flat_w, w_at_save)
    for p, q in zip(a.geom_head.parameters(), c.geom_head.parameters()):
        assert torch.equal(p, q)
    for _ in range(2):
        ta.micro_step(b)
        tc.micro_step(b)
    da = ta.master - snap["master"]
    dc = tc.master - snap["master"]
    assert da.abs().max() > 0
    assert ((da - dc).norm() / da.norm()).item() < 2e-2
    ga, gc = a.text_model.flat_g.float(), c.text_model.flat_g.float()
    assert ((ga - gc).norm() / ga.norm()).item() < 2e-2
    assert tc.lrs() == ta.lrs()
    # mid-window state refuses to load
    ta.micro_step(b)
    save_trainer_state(ta, tmp_path / "mid")
    with pytest.raises(RuntimeError, match="accumulation"):
        load_trainer_state(tc, tmp_path / "mid")


def test_constructor_from_local_hf_directory(tmp_path):
    """The reference's own constructor route (vggt_qwen3_vlm.py:29-45): `text_model_name` is a LOCAL directory with
    config.json + *.safetensors + tokenizer files, `vision_ckpt_dir` has no checkpoint (warning + random init), only the
    reference's dataclass fields are used. The tokenizer gains <image>, the embedding is resized, and the loaded text
    weights reproduce the golden logits."""
    import json
    import shutil
    from safetensors.torch import save_file
    from tests.golden_io import GOLDEN, bf16
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
    z = load("qwen3_tiny.npz")
    c = meta(z, "config")
    d = tmp_path / "tiny_qwen3"
    shutil.copytree(GOLDEN / "tiny_tokenizer", d)
    hf_cfg = dict(architectures=["Qwen3ForCausalLM"], model_type="qwen3", hidden_size=c["hidden_size"],
                  num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                  num_key_value_heads=c["num_key_value_heads"], head_dim=c["head_dim"],
                  intermediate_size=c["intermediate_size"], vocab_size=c["vocab_size"], rms_norm_eps=c["rms_norm_eps"],
                  rope_parameters={"rope_theta": c["rope_theta"], "rope_type": "default"}, tie_word_embeddings=True)
    (d / "config.json").write_text(json.dumps(hf_cfg))
    save_file({k: v.contiguous() for k, v in weights(z).items()}, str(d / "model.safetensors"))
    cfg = VisionLanguageConfig(text_model_name=str(d), vision_ckpt_dir=str(tmp_path / "no_ckpt_here"), num_vis_tokens=16,
                               geom_tokens=4, projector_cfg=PerceiverConfig(latent_dim=128, num_latents=16, num_heads=2,
                                                                            num_layers=1, ffn_dim=256),
                               freeze_vision=True, dtype="bfloat16")
    cfg.vision_config = dict(depth=1, embed_dim=128, dino_depth=1)        # extension field: keep the test's vision tower tiny
    model = VGGTQwen3VLM(cfg).to("cuda")
    tm = model.text_model
    vocab0 = c["vocab_size"]
    assert "<image>" in model.tokenizer.get_vocab() and model.image_id == model.tokenizer.convert_tokens_to_ids("<image>")
    assert tm.vocab == len(model.tokenizer)          # resize_token_embeddings(len(tokenizer)) as the reference does (:40-42)
    assert tm.get_input_embeddings().weight.shape[0] == tm.vocab and tm.config.hidden_size == c["hidden_size"]
    assert all(not p.requires_grad for p in model.vision_model.parameters())
    # the loaded text weights reproduce the golden forward
    emb = bf16(z["inputs_embeds"]).cuda()
    mask = torch.from_numpy(z["attention_mask"]).cuda()
    h, saved = tm.forward_hidden(emb, mask, save=False)
    B, L = mask.shape
    nv = min(vocab0, tm.vocab)
    logits = tm.logits_all(h).view(B, saved["L"], -1)[:, :L, :nv].float().cpu()
    gold = bf16(z["logits"]).float()[..., :nv]
    keep = mask.bool().cpu()
    assert ((logits[keep] - gold[keep]).norm() / gold[keep].norm()).item() < 1e-2
    with pytest.raises(FileNotFoundError, match="local directory"):
        VGGTQwen3VLM(VisionLanguageConfig(text_model_name="Qwen/Qwen3-4B-Instruct-2507", vision_ckpt_dir="x"))


def test_trainer_fit_loop_like_train_sft(tmp_path):
    """Stage1Trainer.fit = the reference's loop body (train_sft.py:208-255): max_steps counts micro-batches, periodic
    step_{n} checkpoints + a final one in the reference's layout, rank-0 log lines with both learning rates and samples/s,
    the iterable restarts when exhausted, and the loss goes down on a tiny fixed set."""
    from vggt_qwen3_amd.checkpoint import INDEX_NAME, MERGED_DIR, TRAINER_STATE, load_checkpoint_if_available
    from vggt_qwen3_amd.trainer import Stage1Trainer
    z = load("vlm_tiny.npz")
    m = meta(z)
    model = _build_vlm(z, m).train()
    from tests.test_trainer_gpu import _Tower
    model.vision_model = _Tower(model.vision_model.agg)      # a tower that follows its input's batch axis: fit() merges micro-batches
    b = _batch(z)
    half = {k: (v[:3] if torch.is_tensor(v) else v) for k, v in b.items()}
    half["geom_token"] = {k: v[:3] for k, v in b["geom_token"].items()}
    tr = Stage1Trainer(model, lr=2e-3, proj_lr=2e-3, weight_decay=0.0, warmup_ratio=0.1, max_steps=10, grad_accum=2)
    lines = []
    recs = tr.fit([b, half], log_every_steps=3, save_every_steps=4, output_dir=tmp_path / "run", log=lines.append)
    assert tr.micro == 10 and tr.opt_step == 5
    assert [r["step"] for r in recs] == [0, 3, 6, 9] and len(lines) == 4 and "samples/s" in lines[0] and "LR:" in lines[0]
    assert recs[-1]["loss"] < recs[0]["loss"]
    assert recs[1]["samples_per_s"] > 0 and recs[0]["lr"] == 0.0 and recs[-1]["lr"] > 0     # warm-up starts at zero
    for d in ("run/step_4", "run/step_8", "run"):
        assert (tmp_path / d / MERGED_DIR / INDEX_NAME).exists(), d
        assert (tmp_path / d / TRAINER_STATE).exists(), d                                  # 4, 8, 10: accumulation boundaries
    fresh = _build_vlm(z, m)
    assert load_checkpoint_if_available(fresh, str(tmp_path / "run"), verbose=False) is not None
    assert torch.equal(fresh.text_model.flat_w, model.text_model.flat_w)


def test_upstream_named_vision_checkpoint_key_space(tmp_path):
    """`vision_model.*` lives in the upstream key space (vggt_qwen3_vlm.py:87-101, qa_inference.py:84,103):
    (1) `_load_vggt` reads a `vggt_1B_commercial.pt`-shaped file (aggregator.* under a "model" wrapper, head tensors ignored);
    (2) state_dict() emits `vision_model.aggregator.<upstream dotted name>`; (3) a reference-written full checkpoint loads
    those tensors through load_checkpoint_if_available and through load_state_dict(strict=False) - none lands in `unexpected`;
    (4) save_model(include_vision=True) writes names the reference can read; (5) the loaded weights are the ones the tower
    computes with (the fp32 compute copies are rebuilt)."""
    from tests.golden_io import GOLDEN
    from vggt_qwen3_amd.checkpoint import load_checkpoint_if_available, save_model
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.vggt import VGGT
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
    vkw = dict(img_size=56, depth=1, embed_dim=128, dino_depth=1)
    donor = VGGT(patch_size=14, device="cuda", seed=123, **vkw)
    g = torch.Generator().manual_seed(1)
    upstream = {}
    for n, t in donor.aggregator.named_tensors().items():
        upstream["aggregator." + n] = (torch.randn(t.shape, generator=g) * 0.05).float()
    upstream["camera_head.trunk.0.weight"] = torch.zeros(4, 4)          # heads exist upstream, never used by the reference
    ck = tmp_path / "vggt"
    ck.mkdir()
    torch.save({"model": upstream}, ck / "vggt_1B_commercial.pt")
    qcfg = Qwen3Config(hidden_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                       intermediate_size=256, vocab_size=300)
    pcfg = PerceiverConfig(latent_dim=128, num_latents=16, num_heads=2, num_layers=1, ffn_dim=256)

    def build(ckpt_dir, seed):
        return VGGTQwen3VLM(VisionLanguageConfig(text_model_name=str(GOLDEN / "tiny_tokenizer"), vision_ckpt_dir=str(ckpt_dir),
                                                 num_vis_tokens=16, geom_tokens=0, projector_cfg=pcfg, text_config=qcfg,
                                                 vision_config=vkw, seed=seed))
    a = build(ck, 0)                                                    # (1) the reference's own vision-weight route
    agg = a.vision_model.aggregator
    for n in ("patch_embed.blocks.0.attn.qkv.weight", "frame_blocks.0.attn.q_norm.bias", "camera_token"):
        assert torch.equal(agg.p(n).cpu(), upstream["aggregator." + n].to(torch.bfloat16)), n
    sd = a.state_dict()                                                 # (2)
    vkeys = [k for k in sd if k.startswith("vision_model.")]
    assert vkeys and all("__" not in k for k in vkeys)
    assert set(vkeys) == {"vision_model." + k for k in upstream if k.startswith("aggregator.")}
    images = torch.rand(1, 2, 3, 56, 56, generator=g).cuda()
    out_a = agg(images)[0][-1].float().clone()
    save_model(a, tmp_path / "full", include_vision=True)               # (4)
    shard = torch.load(next((tmp_path / "full" / "pytorch_model_fp32").glob("*.bin")), map_location="cpu")
    assert "vision_model.aggregator.patch_embed.blocks.0.attn.qkv.weight" in shard
    b = build(tmp_path / "none", 5)                                     # (3) different random vision weights
    assert not torch.equal(b.vision_model.aggregator.p("camera_token"), agg.p("camera_token"))
    _ = b.vision_model.aggregator(images)                               # builds b's compute copies from the OLD weights
    rep = load_checkpoint_if_available(b, str(tmp_path / "full"), verbose=False)
    assert rep is not None and not rep["unexpected"], rep["unexpected"][:4]
    assert not [k for k in rep["missing"] if k.startswith("vision_model.")]
    out_b = b.vision_model.aggregator(images)[0][-1].float()
    assert torch.equal(out_a, out_b)                                    # (5)
    c = build(tmp_path / "none", 6)
    _ = c.vision_model.aggregator(images)
    inc = c.load_state_dict({k: v for k, v in shard.items()}, strict=False)
    assert not inc.unexpected_keys
    assert torch.equal(c.vision_model.aggregator(images)[0][-1].float(), out_a)
