"""Writes tests/golden/dinov2_tiny.npz: the DINOv2-with-registers backbone (VGGT's `patch_embed`) as computed by
transformers' Dinov2WithRegistersModel - an implementation independent of both upstream VGGT and this repo - on seeded
random weights, stored under UPSTREAM VGGT parameter names (`patch_embed.*`). Two inputs: the native grid (no position
interpolation) and a non-square grid (bicubic antialiased interpolation of the position table). Build container only."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.make_golden import bf16_bits  # noqa: E402

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def main():
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    import transformers
    torch.manual_seed(4321)
    C, NH, depth, R, p, img = 128, 2, 2, 4, 14, 56
    cfg = Dinov2WithRegistersConfig(hidden_size=C, num_hidden_layers=depth, num_attention_heads=NH, image_size=img, patch_size=p,
                                    num_register_tokens=R, mlp_ratio=4, layerscale_value=1.0, hidden_act="gelu",
                                    layer_norm_eps=1e-6, qkv_bias=True)
    model = Dinov2WithRegistersModel(cfg).eval()
    with torch.no_grad():
        for n, q in model.named_parameters():          # non-trivial norms / layer scales / tokens, bf16-representable values
            if "norm" in n and n.endswith("weight") or "lambda1" in n:
                q.copy_(1.0 + 0.2 * torch.randn_like(q))
            elif q.dim() == 1 or "token" in n or "position" in n:
                q.copy_(0.1 * torch.randn_like(q))
            q.copy_(q.to(torch.bfloat16).float())
    hf = dict(model.named_parameters())
    sd = {"patch_embed.cls_token": hf["embeddings.cls_token"], "patch_embed.register_tokens": hf["embeddings.register_tokens"],
          "patch_embed.pos_embed": hf["embeddings.position_embeddings"],
          "patch_embed.patch_embed.proj.weight": hf["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.patch_embed.proj.bias": hf["embeddings.patch_embeddings.projection.bias"],
          "patch_embed.norm.weight": hf["layernorm.weight"], "patch_embed.norm.bias": hf["layernorm.bias"]}
    for i in range(depth):
        a, b = f"encoder.layer.{i}.", f"patch_embed.blocks.{i}."
        att = a + "attention.attention."
        sd[b + "attn.qkv.weight"] = torch.cat([hf[att + "query.weight"], hf[att + "key.weight"], hf[att + "value.weight"]], 0)
        sd[b + "attn.qkv.bias"] = torch.cat([hf[att + "query.bias"], hf[att + "key.bias"], hf[att + "value.bias"]], 0)
        sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"] = hf[a + "attention.output.dense.weight"], hf[a + "attention.output.dense.bias"]
        sd[b + "ls1.gamma"], sd[b + "ls2.gamma"] = hf[a + "layer_scale1.lambda1"], hf[a + "layer_scale2.lambda1"]
        for k in ("norm1", "norm2"):
            sd[b + k + ".weight"], sd[b + k + ".bias"] = hf[a + k + ".weight"], hf[a + k + ".bias"]
        for k in ("fc1", "fc2"):
            sd[b + f"mlp.{k}.weight"], sd[b + f"mlp.{k}.bias"] = hf[a + f"mlp.{k}.weight"], hf[a + f"mlp.{k}.bias"]
    arrays = {"w:" + k: bf16_bits(v.detach()) for k, v in sd.items()}
    mean, std = torch.tensor(MEAN).view(1, 3, 1, 1), torch.tensor(STD).view(1, 3, 1, 1)
    for name, (H, W) in {"native": (56, 56), "interp": (70, 42)}.items():
        images = torch.rand(2, 3, H, W)
        with torch.no_grad():
            out = model(pixel_values=(images - mean) / std, interpolate_pos_encoding=True).last_hidden_state
        arrays[f"{name}:images"] = images.numpy()
        arrays[f"{name}:tokens"] = out.float().numpy()
        print(name, tuple(out.shape), float(out.abs().mean()))
    arrays["meta"] = np.frombuffer(json.dumps({"embed_dim": C, "num_heads": NH, "depth": depth, "registers": R, "patch": p,
                                               "transformers": transformers.__version__}).encode(), np.uint8)
    np.savez_compressed(ROOT / "tests" / "golden" / "dinov2_tiny.npz", **arrays)


if __name__ == "__main__":
    main()
