"""Generate tests/golden/*.npz from the REFERENCE's own modules (run in the build container only).

    PYTHONPATH=/root/reference python tools/make_golden.py

What is imported from where:
  * /root/reference/src/models/projector_perceiver.py  (PerceiverProjector)          - in-repo reference code
  * /root/reference/src/models/vggt_qwen3_vlm.py       (VGGTQwen3VLM.forward)        - in-repo reference code
  * /root/reference/src/dataio/collate_multiview.py    (MultiViewCollator)           - in-repo reference code
  * transformers Qwen3ForCausalLM                                                     - the reference's pip dependency
The un-vendored `vggt` package is replaced by a stub module whose aggregator returns the fixture's own token tensor
(the reference only consumes `aggregator(images) -> (list, patch_start_idx)`, vggt_qwen3_vlm.py:144-148), and
torchvision (absent here) by a stub exposing the three transform names the collator composes. No reference source
is copied: only inputs, seeded weights (bf16-representable values) and the outputs the reference produced are
written, as data. All models run on CPU in eval mode (dropout off), without autocast: Perceiver/geom_head in fp32,
Qwen3 in bf16 - the regime of the reference's plain CPU forward.
"""
from __future__ import annotations

import json
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
sys.path.insert(0, str(REF))
sys.path.insert(0, str(ROOT))


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def snap_bf16_(module: torch.nn.Module) -> None:
    """Make every parameter value bf16-representable (keeps the parameter's own dtype)."""
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(p.to(torch.bfloat16).to(p.dtype))


def save(name: str, **arrays) -> None:
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / name, **arrays)
    print(f"wrote {name}: {sum(a.nbytes for a in arrays.values()) / 1e6:.2f} MB raw, "
          f"{(OUT / name).stat().st_size / 1e6:.2f} MB on disk")


# ----------------------------------------------------------------------------------------------- tokenizer
def build_tokenizer(dirpath: Path):
    """Small BPE trained on the reference's ScanQA/SQA3D test questions; saved as a HF fast tokenizer."""
    from tokenizers import Tokenizer, models, pre_tokenizers, trainers, decoders
    from transformers import PreTrainedTokenizerFast
    texts = []
    for rel in ("data/processed/scanqa/test_split.jsonl", "data/processed/sqa3d/test_split.jsonl"):
        p = REF / rel
        if p.exists():
            for line in p.read_text().splitlines()[:1500]:
                try:
                    row = json.loads(line)
                except Exception:
                    continue
                texts.append(str(row.get("question", "")))
                a = row.get("answer", "")
                texts.append(a if isinstance(a, str) else json.dumps(a))
    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=300, special_tokens=["<unk>", "<pad>", "<eos>"],
                             initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(texts, tr)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", pad_token="<pad>", eos_token="<eos>")
    fast.save_pretrained(str(dirpath))
    return fast


# ----------------------------------------------------------------------------------------------- Qwen3 (HF)
def tiny_qwen_cfg(vocab: int):
    from transformers import Qwen3Config
    return Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                       head_dim=128, intermediate_size=256, vocab_size=vocab, rms_norm_eps=1e-6,
                       tie_word_embeddings=True, max_position_embeddings=512,
                       rope_parameters={"rope_theta": 5_000_000.0, "rope_type": "default"})


def golden_qwen3():
    from transformers import Qwen3ForCausalLM
    torch.manual_seed(1234)
    cfg = tiny_qwen_cfg(320)
    model = Qwen3ForCausalLM(cfg).to(torch.bfloat16).eval()
    with torch.no_grad():  # make norms non-trivial
        for n, p in model.named_parameters():
            if "norm" in n:
                p.copy_((1.0 + 0.2 * torch.randn_like(p.float())).to(p.dtype))
    B, L = 3, 40
    emb = (torch.randn(B, L, cfg.hidden_size) * 0.5).to(torch.bfloat16)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[0, 25:] = 0
    mask[1, 33:] = 0
    labels = torch.full((B, L), -100, dtype=torch.long)
    labels[0, 20:25] = torch.randint(0, 320, (5,))
    labels[1, 30:33] = torch.randint(0, 320, (3,))
    labels[2, 36:40] = torch.randint(0, 320, (4,))
    emb_g = emb.clone().requires_grad_(True)
    out = model(inputs_embeds=emb_g, attention_mask=mask, labels=labels, output_hidden_states=True)
    out.loss.backward()
    arrays = {
        "config": np.frombuffer(json.dumps(dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=2,
                                                num_key_value_heads=1, head_dim=128, intermediate_size=256,
                                                vocab_size=320, rms_norm_eps=1e-6, rope_theta=5e6)).encode(), np.uint8),
        "inputs_embeds": bf16_bits(emb), "attention_mask": mask.numpy(), "labels": labels.numpy(),
        "loss": out.loss.detach().float().numpy(), "logits": bf16_bits(out.logits),
        "d_inputs_embeds": bf16_bits(emb_g.grad),
    }
    for i, h in enumerate(out.hidden_states):
        arrays[f"hidden_{i}"] = bf16_bits(h)
    for n, p in model.state_dict().items():
        if n == "lm_head.weight":
            continue
        arrays["w:" + n] = bf16_bits(p)
    for n, p in model.named_parameters():
        if p.grad is not None and n != "lm_head.weight":
            arrays["g:" + n] = bf16_bits(p.grad)
    save("qwen3_tiny.npz", **arrays)


# ----------------------------------------------------------------------------------------------- Perceiver (reference)
def golden_perceiver():
    from src.models.projector_perceiver import PerceiverConfig, PerceiverProjector
    torch.manual_seed(4321)
    cfg = PerceiverConfig(latent_dim=128, num_latents=16, num_heads=2, num_layers=2, ffn_dim=256, dropout=0.1)
    proj = PerceiverProjector(cfg, in_dim=128, out_dim=256).eval()
    with torch.no_grad():  # the reference zero-inits biases; make them non-trivial so bias paths are exercised
        for n, p in proj.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
            if "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
    snap_bf16_(proj)
    tokens = (torch.randn(2, 24, 128)).to(torch.bfloat16).float()
    with torch.no_grad():
        out = proj(tokens)
    arrays = {"tokens": tokens.numpy(), "out": out.numpy(),
              "config": np.frombuffer(json.dumps(dict(latent_dim=128, num_latents=16, num_heads=2, num_layers=2,
                                                      ffn_dim=256, in_dim=128, out_dim=256)).encode(), np.uint8)}
    for n, p in proj.state_dict().items():
        arrays["w:" + n] = bf16_bits(p)
    save("perceiver_tiny.npz", **arrays)


# ----------------------------------------------------------------------------------------------- collator (reference)
def install_torchvision_stub():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class _T:
        def __init__(self, *a, **k): pass
        def __call__(self, x): return x

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class InterpolationMode:
        BICUBIC = "bicubic"

    tr.Resize = tr.CenterCrop = tr.ToTensor = _T
    tr.Compose = Compose
    tr.InterpolationMode = InterpolationMode
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr


def load_rows(n_scan: int, n_sqa: int):
    rows = []
    for rel, n in (("data/processed/scanqa/test_split.jsonl", n_scan), ("data/processed/sqa3d/test_split.jsonl", n_sqa)):
        lines = (REF / rel).read_text().splitlines()
        for line in lines[:n]:
            r = json.loads(line)
            rows.append({"question": r["question"], "answer": r["answer"]})
    return rows


# ----------------------------------------------------------------------------------------------- full VLM (reference)
def golden_vlm_and_collate():
    install_torchvision_stub()
    from transformers import Qwen3ForCausalLM
    from src.dataio.collate_multiview import MultiViewCollator
    from src.models.projector_perceiver import PerceiverConfig
    num_vis, geom_tok = 16, 4
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        tok = build_tokenizer(td)
        # keep the tokenizer as a (data) fixture so the GPU box can re-tokenise the same strings
        tokdir = OUT / "tiny_tokenizer"
        tokdir.mkdir(parents=True, exist_ok=True)
        tok.save_pretrained(str(tokdir))
        base_vocab = len(tok)
        torch.manual_seed(777)
        cfg = tiny_qwen_cfg(base_vocab)
        lm = Qwen3ForCausalLM(cfg).to(torch.bfloat16)
        with torch.no_grad():
            for n, p in lm.named_parameters():
                if "norm" in n:
                    p.copy_((1.0 + 0.2 * torch.randn_like(p.float())).to(p.dtype))
        lm.save_pretrained(str(td), safe_serialization=True)

        # stub for the un-vendored vggt package: aggregator returns the fixture tensor
        holder = {}
        vg = types.ModuleType("vggt"); vgm = types.ModuleType("vggt.models"); vgv = types.ModuleType("vggt.models.vggt")

        class VGGT(torch.nn.Module):
            def __init__(self, **kw):
                super().__init__()
                self.dummy = torch.nn.Parameter(torch.zeros(1))

            def aggregator(self, images):
                return [holder["agg"].to(images.dtype)], 5

        vgv.VGGT = VGGT
        sys.modules.update({"vggt": vg, "vggt.models": vgm, "vggt.models.vggt": vgv})
        from src.models.vggt_qwen3_vlm import VGGTQwen3VLM, VisionLanguageConfig

        pcfg = PerceiverConfig(latent_dim=128, num_latents=num_vis, num_heads=2, num_layers=2, ffn_dim=256, dropout=0.1)
        vcfg = VisionLanguageConfig(text_model_name=str(td), vision_ckpt_dir=str(td), num_vis_tokens=num_vis,
                                    geom_tokens=geom_tok, projector_cfg=pcfg, freeze_vision=True, dtype="bfloat16")
        model = VGGTQwen3VLM(vcfg).eval()
        with torch.no_grad():
            for n, p in list(model.projector.named_parameters()) + list(model.geom_head.named_parameters()):
                if n.endswith("bias"):
                    p.copy_(0.1 * torch.randn_like(p))
        snap_bf16_(model.projector)
        snap_bf16_(model.geom_head)
        image_id = model.tokenizer.convert_tokens_to_ids("<image>")

        # ---- collator golden on real ScanQA / SQA3D strings (token indices must be bit-exact)
        ctok = model.tokenizer
        coll = MultiViewCollator(image_size=28, tokenizer=ctok, max_length=512, num_vis_tokens=num_vis,
                                 geom_tokens=geom_tok)
        rows = load_rows(4, 2)
        V = 2
        batch_in = []
        g = torch.Generator().manual_seed(99)
        for i, r in enumerate(rows):
            geom = None
            if i != 3:  # one sample without geometry -> zero-filled by the collator
                geom = {"R": torch.randn(V, 9, generator=g).tolist(), "t": torch.randn(V, 3, generator=g).tolist(),
                        "K": torch.randn(V, 9, generator=g).tolist(),
                        "depth_hist": torch.softmax(torch.randn(V, 16, generator=g), -1).tolist()}
            batch_in.append({"images": [torch.rand(3, 28, 28, generator=g) for _ in range(V)],
                             "question": r["question"], "answer": r["answer"], "geom_token": geom})
        batch = coll(batch_in)
        B, L = batch["input_ids"].shape
        P = 10
        agg = (torch.randn(B, V, P, 2048, generator=g) * 0.5).to(torch.bfloat16)
        holder["agg"] = agg

        geom_in = {k: v for k, v in batch["geom_token"].items()}
        loss = model(images=batch["pixel_values"], geom_token=geom_in, input_ids=batch["input_ids"],
                     attention_mask=batch["attention_mask"], labels=batch["labels"])
        loss.backward()
        with torch.no_grad():
            vis = model.encode_images(batch["pixel_values"])
            gfeat = model.encode_geom(geom_in)
            feats = torch.cat([gfeat, vis], dim=1)
            emb = model.text_model.get_input_embeddings()(batch["input_ids"])
            for b, pos in (batch["input_ids"] == image_id).nonzero(as_tuple=False):
                emb[b, pos:pos + feats.size(1), :] = feats[b]
            out = model.text_model(inputs_embeds=emb, attention_mask=batch["attention_mask"], labels=batch["labels"])

        arrays = {
            "meta": np.frombuffer(json.dumps(dict(num_vis_tokens=num_vis, geom_tokens=geom_tok, image_id=image_id,
                                                  pad_id=ctok.pad_token_id, vocab=len(ctok), latent_dim=128,
                                                  num_heads=2, num_layers=2, ffn_dim=256, hidden_size=256,
                                                  num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                                                  head_dim=128, intermediate_size=256, rms_norm_eps=1e-6,
                                                  rope_theta=5e6, min_text_length=coll.min_text_length,
                                                  max_length=512)).encode(), np.uint8),
            "questions": np.frombuffer(json.dumps([r["question"] for r in rows]).encode(), np.uint8),
            "answers": np.frombuffer(json.dumps([r["answer"] for r in rows]).encode(), np.uint8),
            "input_ids": batch["input_ids"].numpy(), "attention_mask": batch["attention_mask"].numpy(),
            "labels": batch["labels"].numpy(), "geom_mask": batch["geom_token"]["mask"].numpy(),
            "agg": bf16_bits(agg), "pixel_values": batch["pixel_values"].numpy().astype(np.float16),
            "vis_tokens": vis.float().numpy(), "geom_feats": gfeat.float().numpy(),
            "inputs_embeds": bf16_bits(emb), "loss": loss.detach().float().numpy(),
            "logits": bf16_bits(out.logits),
        }
        for k in ("R", "t", "K", "depth_hist"):
            arrays["geom:" + k] = batch["geom_token"][k].numpy()
        for n, p in model.state_dict().items():
            if n.startswith("vision_model") or n == "text_model.lm_head.weight":
                continue
            arrays["w:" + n] = bf16_bits(p)
        ngrad = 0
        for n, p in model.named_parameters():
            if p.grad is not None and n != "text_model.lm_head.weight":
                arrays["g:" + n] = bf16_bits(p.grad) if p.dtype == torch.bfloat16 else p.grad.float().numpy()
                ngrad += 1
        nograd = [n for n, p in model.named_parameters() if p.grad is None and p.requires_grad]
        arrays["params_without_grad"] = np.frombuffer(json.dumps(nograd).encode(), np.uint8)
        print(f"vlm golden: B={B} L={L} loss={loss.item():.4f} grads for {ngrad} params, none for {len(nograd)}")
        save("vlm_tiny.npz", **arrays)


if __name__ == "__main__":
    torch.set_num_threads(8)
    golden_qwen3()
    golden_perceiver()
    golden_vlm_and_collate()
