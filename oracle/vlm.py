"""VGGTQwen3VLM.forward restated (vggt_qwen3_vlm.py:128-201) on top of oracle.qwen3 / oracle.perceiver."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import perceiver, qwen3


def select_tokens(agg_last: torch.Tensor, num_vis_tokens: int) -> torch.Tensor:
    """vggt_qwen3_vlm.py:148-156: last aggregator iterate, flatten views, keep the first num_vis_tokens tokens."""
    if agg_last.dim() == 3:
        return agg_last[:, :num_vis_tokens, :]
    B = agg_last.shape[0]
    return agg_last.reshape(B, -1, agg_last.shape[-1])[:, :num_vis_tokens, :]


def encode_geom(geom: Optional[Dict[str, torch.Tensor]], sd, geom_tokens: int):
    """vggt_qwen3_vlm.py:164-177: cat(R,t,K,depth_hist) -> mean over views -> Linear-SiLU-Linear -> expand."""
    if not geom or geom_tokens == 0:
        return None
    feats = torch.cat([geom["R"], geom["t"], geom["K"], geom["depth_hist"]], dim=-1).mean(dim=1)
    g = F.linear(F.silu(F.linear(feats, sd["geom_head.0.weight"], sd["geom_head.0.bias"])), sd["geom_head.2.weight"],
                 sd["geom_head.2.bias"])
    return g.unsqueeze(1).expand(-1, geom_tokens, -1)


def splice(inputs_embeds: torch.Tensor, input_ids: torch.Tensor, features: torch.Tensor, image_id: int):
    """vggt_qwen3_vlm.py:191-195: OVERWRITE rows pos..pos+S-1 (raises like the reference if the span overruns)."""
    out = inputs_embeds.clone()
    for b, pos in (input_ids == image_id).nonzero(as_tuple=False).tolist():
        span = features[b]
        out[b, pos:pos + span.size(0), :] = span
    return out


def splice_srcmap(input_ids: torch.Tensor, S: int, image_id: int) -> torch.Tensor:
    """Integer image of the splice loop: srcmap[b,l] = feature row written at (b,l) or -1. Bit-exact index work."""
    B, L = input_ids.shape
    m = torch.full((B, L), -1, dtype=torch.int32)
    for b, pos in (input_ids == image_id).nonzero(as_tuple=False).tolist():
        if pos + S > L:
            raise RuntimeError(f"visual span overruns the sequence: pos {pos} + {S} > {L}")
        m[b, pos:pos + S] = torch.arange(S, dtype=torch.int32)
    return m


def forward(agg_last, geom, input_ids, attention_mask, labels, sd, qcfg: qwen3.Qwen3Cfg, *, heads, num_layers,
            num_vis_tokens, geom_tokens, image_id, proj_dtype=torch.float32, text_dtype=torch.bfloat16, collect=None):
    """Returns dict(vis_tokens, features, inputs_embeds, loss, logits). sd holds `projector.*`, `geom_head.*`,
    `text_model.*` entries (the reference's checkpoint key space)."""
    psd = {k[len("projector."):]: v.to(proj_dtype) for k, v in sd.items() if k.startswith("projector.")}
    tsd = {k[len("text_model."):]: v.to(text_dtype) for k, v in sd.items() if k.startswith("text_model.")}
    gsd = {k: v.to(proj_dtype) for k, v in sd.items() if k.startswith("geom_head.")}
    tok = select_tokens(agg_last, num_vis_tokens).to(proj_dtype)
    vis = perceiver.projector(tok, psd, heads, num_layers)
    g = encode_geom({k: v.to(proj_dtype) for k, v in geom.items() if k != "mask"} if geom else None, gsd, geom_tokens)
    feats = vis if g is None else torch.cat([g, vis], dim=1)
    emb = F.embedding(input_ids, tsd["model.embed_tokens.weight"])
    emb = splice(emb, input_ids, feats.to(emb.dtype), image_id)
    loss, logits = qwen3.causal_lm(emb, attention_mask, labels, tsd, qcfg, collect)
    return {"vis_tokens": vis, "features": feats, "inputs_embeds": emb, "loss": loss, "logits": logits}
