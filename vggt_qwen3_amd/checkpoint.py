"""Checkpoint I/O in the reference's key space.

The reference trains under Accelerate/DeepSpeed and later reads the merged fp32 export: a directory
`pytorch_model_fp32/` holding `pytorch_model-0000i-of-0000n.bin` shards plus `pytorch_model.bin.index.json`, or a flat
`*.bin` / `*.safetensors` file, loaded with `strict=False` (src/inference/qa_inference.py:51-105,
arkit_inference.py:93). This module writes and reads exactly that layout with the reference's parameter names
(`text_model.model.layers.N...`, `projector...`, `geom_head...`, optionally `vision_model...`), so a checkpoint
moves between the two implementations in both directions.

Weights stream shard by shard straight into the resident HBM buffers - the model is never staged on the host the way
the reference's loader does (`model.to("cpu")` ... `model.to(device)`); VGGTQwen3VLM treats such moves as no-ops.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import torch

MERGED_DIR = "pytorch_model_fp32"
LEGACY_MERGED_DIR = "pytorch_model_fp32.bin"
INDEX_NAME = "pytorch_model.bin.index.json"
TRAINER_STATE = "trainer_state.pt"


def checkpoint_items(model, include_vision: bool = False) -> Iterable[Tuple[str, torch.Tensor]]:
    """(name, tensor) pairs in the reference's state_dict order. `text_model.lm_head.weight` is tied to
    `text_model.model.embed_tokens.weight` and written once, under the embedding's name."""
    for name, t in model.state_dict().items():
        if name == "text_model.lm_head.weight":
            continue
        if name.startswith("vision_model.") and not include_vision:
            continue
        yield name, t


def save_model(model, out_dir, max_shard_bytes: int = 5 << 30, include_vision: bool = False,
               dtype: torch.dtype = torch.float32) -> Dict[str, str]:
    """Write `out_dir/pytorch_model_fp32/{shards, index}`. Returns the weight map (name -> shard file)."""
    root = Path(out_dir) / MERGED_DIR
    root.mkdir(parents=True, exist_ok=True)
    if hasattr(model, "_pass_weights_gate"):
        model._pass_weights_gate()            # an optimiser step Stage1Trainer left running on its side stream finishes first
    shards: List[Dict[str, torch.Tensor]] = [{}]
    size, total = 0, 0
    for name, t in checkpoint_items(model, include_vision):
        nbytes = t.numel() * torch.empty((), dtype=dtype).element_size()
        if shards[-1] and size + nbytes > max_shard_bytes:
            shards.append({})
            size = 0
        shards[-1][name] = t.detach().to(device="cpu", dtype=dtype).contiguous()
        size += nbytes
        total += nbytes
    n = len(shards)
    weight_map: Dict[str, str] = {}
    for i, sd in enumerate(shards):
        fname = f"pytorch_model-{i + 1:05d}-of-{n:05d}.bin"
        torch.save(sd, root / fname)
        for k in sd:
            weight_map[k] = fname
    with (root / INDEX_NAME).open("w", encoding="utf-8") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
    return weight_map


def load_state_into(model, state: Dict[str, torch.Tensor]) -> Tuple[List[str], List[str]]:
    """`load_state_dict(strict=False)` semantics without leaving the device: copies matching names into the resident
    parameters/buffers, returns (matched, unexpected). Shape mismatches raise like torch's loader."""
    own = dict(model.state_dict())
    matched, unexpected = [], []
    with torch.no_grad():
        for k, v in state.items():
            if k not in own:
                unexpected.append(k)
                continue
            dst = own[k]
            if tuple(dst.shape) != tuple(v.shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, "
                                   f"the shape in current model is {tuple(dst.shape)}")
            dst.copy_(v.to(device=dst.device, dtype=dst.dtype, non_blocking=False))
            matched.append(k)
    refresh_after_weight_load(model)
    return matched, unexpected


def refresh_after_weight_load(model) -> None:
    """Everything derived FROM the weights must follow a load that bypasses nn.Module.load_state_dict's post-hooks: the
    text model's e4m3 / W^T copies, the projector's and the aggregator's compute copies, and the fp32 master weights of
    any Stage1Trainer built on this model (otherwise its next AdamW step would write the pre-load weights back)."""
    tm = getattr(model, "text_model", None)
    if tm is not None and hasattr(tm, "refresh_derived"):
        tm.refresh_derived()
    proj = getattr(model, "projector", None)
    if proj is not None and hasattr(proj, "_cc"):
        proj._cc = None
    agg = getattr(getattr(model, "vision_model", None), "aggregator", None)
    if agg is not None and hasattr(agg, "invalidate_compute_copies"):
        agg.invalidate_compute_copies()
    for ref in list(getattr(model, "_trainers", [])):
        tr = ref()
        if tr is not None:
            tr.resync_master_from_weights()


def _read_file(path: Path) -> Dict[str, torch.Tensor]:
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path))
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return sd


def load_checkpoint_if_available(model, ckpt_dir: Optional[str], verbose: bool = True) -> Optional[dict]:
    """Same search order as the reference (qa_inference.py:51-105): sharded `pytorch_model_fp32/` (or the legacy
    `pytorch_model_fp32.bin/`) with an index -> all its shards; else the first flat *.bin / *.safetensors found.
    Returns {"files", "matched", "missing", "unexpected"} or None when nothing was loaded (base weights kept)."""
    if not ckpt_dir:
        return None
    path = Path(ckpt_dir)
    say = print if verbose else (lambda *a, **k: None)
    if not path.exists():
        say(f"Checkpoint directory {path} does not exist; running with base weights.")
        return None
    files: List[Path] = []
    for cand in (path / MERGED_DIR, path / LEGACY_MERGED_DIR):
        index = cand / INDEX_NAME
        if cand.is_dir() and index.exists():
            with index.open("r", encoding="utf-8") as f:
                wm = json.load(f).get("weight_map", {})
            files = [cand / n for n in sorted(set(wm.values()))]
            break
    if not files:
        merged, legacy = path / MERGED_DIR, path / LEGACY_MERGED_DIR
        if merged.exists():
            cands = [merged] if merged.is_file() else sorted(merged.glob("*.bin"))
        elif legacy.is_dir():
            cands = sorted(legacy.glob("*.bin"))
        else:
            cands = list(path.glob("*.bin")) + list(path.glob("*.safetensors"))
        files = cands[:1]           # the reference reads only the first flat file
    if not files:
        say(f"No model weights found in {path}; using base weights.")
        return None
    matched: List[str] = []
    unexpected: List[str] = []
    for f in files:
        m, u = load_state_into(model, _read_file(f))
        matched += m
        unexpected += u
    have = set(matched)
    missing = [k for k in model.state_dict() if k not in have and
               not (k == "text_model.lm_head.weight" and "text_model.model.embed_tokens.weight" in have)]
    say(f"Loaded {len(matched)} tensors from {len(files)} file(s); missing {len(missing)}, unexpected {len(unexpected)}")
    return {"files": [str(f) for f in files], "matched": matched, "missing": missing, "unexpected": unexpected}


# ---------------------------------------------------------------------- trainer state (resume)
def save_trainer_state(trainer, out_dir) -> Path:
    """fp32 master weights, Adam moments, counters. (The reference cannot resume - SURVEY.md appendix A - so this
    format is ours; the model weights next to it stay in the reference's layout.)"""
    if getattr(trainer, "dp_mode", "allreduce") == "sharded" and not getattr(trainer, "_shards_gathered", True):
        #  the fp32 master weights and moments of the other ranks' shards are stale here: written out, a resume would put
        # the old weights back at the next AdamW step. Every rank has to call trainer.gather_sharded_state() first (a collective).
        raise RuntimeError("save_trainer_state: dp_mode='sharded' and the optimiser state has not been gathered since the last "
                           "optimiser step - call trainer.gather_sharded_state() on every rank first (Stage1Trainer._save does)")
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    st = {"micro": trainer.micro, "opt_step": trainer.opt_step, "grad_accum": trainer.grad_accum,
          "layout": {k: [int(o), list(s)] for k, (o, s) in trainer.tm.table.items()},
          "master": trainer.master.cpu(), "m": trainer.m.cpu(), "v": trainer.v.cpu(),
          "geom_master": trainer.geom_master.cpu(), "geom_m": trainer.geom_m.cpu(), "geom_v": trainer.geom_v.cpu()}
    if getattr(trainer, "proj_on", False):       # trained projector ("corrected" mode): its fp32 parameters live in the model file
        st["proj_m"], st["proj_v"] = trainer.proj_m.cpu(), trainer.proj_v.cpu()
    torch.save(st, out / TRAINER_STATE)
    return out / TRAINER_STATE


def load_trainer_state(trainer, ckpt_dir) -> None:
    st = torch.load(Path(ckpt_dir) / TRAINER_STATE, map_location="cpu", weights_only=True)
    layout = {k: [int(o), list(s)] for k, (o, s) in trainer.tm.table.items()}
    if st["layout"] != layout:
        raise RuntimeError("trainer state was written for a different flat parameter layout")
    if st["micro"] % trainer.grad_accum:
        raise RuntimeError("trainer state was saved in the middle of a gradient-accumulation window")
    trainer.micro, trainer.opt_step = st["micro"], st["opt_step"]
    for name in ("master", "m", "v", "geom_master", "geom_m", "geom_v"):
        getattr(trainer, name).copy_(st[name])
    if getattr(trainer, "proj_on", False) and "proj_m" in st:
        trainer.proj_m.copy_(st["proj_m"]); trainer.proj_v.copy_(st["proj_v"])
    from . import ops
    trainer.tm.flat_w.copy_(ops.cast(trainer.master, torch.bfloat16))
    trainer.tm.refresh_derived()      # e4m3 / W^T copies follow the restored weights (they were built at construction)
    off = 0
    with torch.no_grad():
        for p in trainer.geom_params:
            p.copy_(trainer.geom_master[off:off + p.numel()].view_as(p))
            off += p.numel()
