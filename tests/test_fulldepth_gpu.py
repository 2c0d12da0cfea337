"""The north star's own acceptance criterion at FULL DEPTH: "logits within 1e-2 rel. of the CPU reference" for the whole
VGGT (24 DINOv2 + 24 frame + 24 global blocks) -> Perceiver (128 latents, 6 layers) -> Qwen3-4B (36 layers, full vocabulary)
forward, BASELINE config C1/C2 shape (V = 1, 448 x 448, L = 200), seeded random weights at the exact shapes.

What is compared (reference: /root/reference/src/models/vggt_qwen3_vlm.py:128-201; transformers modeling_qwen3.py:367-508;
loss_utils.py:49-71; projector_perceiver.py:44-82):

  HIP  : vggt_qwen3_amd.VGGTQwen3VLM on the GPU (bf16 MFMA GEMMs, f32 accumulate, fused epilogues)
  ref16: the CPU oracle evaluated the way the reference's CPU forward runs (config C1: bf16 tower, fp32 projector, bf16 LLM, every op
         rounding its output to bf16)
  ref32: the same oracle on the same (bf16-valued) weights with every op in fp32 - the value both bf16 evaluations approximate

at every stage: the consumed tower tokens, the Perceiver's visual tokens, inputs_embeds, the hidden state after each of the 36
decoder layers, the logits on attended positions, the loss. The batch holds two samples: row 0 has the collator's layout (only the
visual rows under real text are attended: collate_multiview.py:63-76 does not extend the mask over the span), row 1 a 170-token text
so that the whole 128-row visual span is attended and the towers reach the logits through every row.

Tolerances are stated BEFORE the measurement and are of two kinds:
  (1) the north star's: logits rel <= 1e-2 and loss rel <= 5e-3 against ref16 - asserted when bf16 evaluation itself allows it, i.e.
      when ref16 is within 1e-2 / 2 of ref32 (otherwise two correct bf16 evaluations of the same network cannot be expected to agree
      to 1e-2 and the figure is reported, not asserted);
  (2) always: HIP is at least as close to the exact (fp32) value as the reference's own bf16 evaluation is, with 25 % slack:
      err(HIP, ref32) <= 1.25 * err(ref16, ref32) + 1e-3 for logits, hidden states and visual tokens; and HIP and ref16 sit within the
      sum of their distances to ref32 of each other (triangle bound, always true - kept as a check of the bookkeeping).
Greedy token choice (argmax of the logits on attended positions): HIP may differ from the exact evaluation only where the exact top-2
gap is inside the reference's OWN bf16 noise at that position (<= 6 x the RMS of ref16 - ref32 over the vocabulary there), and not at
more positions than 1.5 x the reference's bf16 evaluation does + 3.

Two weight regimes:
  A  end to end, the model's own initialisation (HF `initializer_range` 0.02 everywhere, the weights bench.py runs): 36 random layers
     amplify rounding - measured round 5: ref16 itself sits 3.4e-2 from ref32 on the logits, HIP 3.5e-2, HIP vs ref16 3.2e-2 (the curve
     crosses 1e-2 after layer 1); kind (2) is what can be asserted, kind (1) is reported.
  B  the 36-layer text model alone on the same inputs_embeds, with the residual-branch output projections (o_proj, down_proj) scaled by
     1 / sqrt(2 * 36) - the published depth-scaled initialisation (GPT-2 / Megatron "scaled init") under which, as in a trained
     network, each layer's update is small against the residual stream. Measured round 5: the error then grows LINEARLY with depth,
     5.5e-4 per layer - the 72 bf16 roundings of the residual stream itself (2^-9 / sqrt(3) each, summed in quadrature: 1e-2 for ANY
     36-layer model whose residual stream is bf16, the reference's included): ref16 sits 1.96e-2 from ref32 on the logits, HIP 1.94e-2,
     HIP vs ref16 2.2e-2; the literal 1e-2 holds through layer 10. No weight regime makes two bf16 evaluations of 36 bf16-residual layers
     agree to 1e-2; kind (2) is the statement that can be made at full depth, and it is asserted for every layer.
The error-vs-depth curves are written to gpurun_out/r5_depth_parity.json (tools/run_profiles_r5.sh copies it to profiles/)."""
import importlib.util
import json
import os
import time
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
BF16, F32 = torch.bfloat16, torch.float32


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _bench():
    spec = importlib.util.spec_from_file_location("vq3_bench", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _batch(model):
    """Row 0: bench.synthetic_batch's layout. Row 1: 170 text tokens, <image> at position 20: the 128-row span lies
    under attended text entirely; labels on the last 8 text positions."""
    dev = torch.device("cuda")
    b = _bench().synthetic_batch(2, 1, 200, 448, 151936, model.image_id, 151643, 198, 1234, dev, False)
    g = torch.Generator().manual_seed(99)
    ids, labels = b["input_ids"].cpu().clone(), b["labels"].cpu().clone()
    t = torch.randint(1000, 150000, (170,), generator=g)
    t[20] = model.image_id
    ids[1] = 151643
    ids[1, :170] = t
    labels[1] = -100
    labels[1, 162:170] = t[162:170]
    b["input_ids"], b["labels"] = ids.to(dev), labels.to(dev)
    b["attention_mask"] = (ids != 151643).long().to(dev)
    return b


def _oracle_vision(images, ids, vsd, psd, emb_table, image_id, heads, players, tower_dtype, proj_dtype, text_dtype):
    """Tower -> first 128 tokens -> Perceiver -> embedding + splice (vggt_qwen3_vlm.py:128-162,179-195)."""
    from oracle import perceiver as operc, vggt as ov, vlm as ovlm
    out = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        agg = ov.aggregator(images, vsd, num_heads=16, depth=24, dino_depth=24, dtype=tower_dtype)[-1]
        out["t_tower"] = time.perf_counter() - t0
        tok = ovlm.select_tokens(agg, 128)
        out["tower_tokens"] = tok.float()
        t1 = time.perf_counter()
        vis = operc.projector(tok.to(proj_dtype), {k: v.to(proj_dtype) for k, v in psd.items()}, heads, players)
        out["t_perceiver"] = time.perf_counter() - t1
        out["vis_tokens"] = vis.float()
        emb = torch.nn.functional.embedding(ids, emb_table.to(text_dtype))
        out["emb"] = ovlm.splice(emb, ids, vis.to(emb.dtype), image_id)
        out["inputs_embeds"] = out["emb"].float()
    return out


def _oracle_text(emb, mask, labels, tsd, text_dtype):
    """Qwen3ForCausalLM.forward (modeling_qwen3.py:367-508) + the shifted mean CE (loss_utils.py:49-71), hidden state after every layer."""
    from oracle import qwen3 as oq
    out, hs = {}, []
    t0 = time.perf_counter()
    with torch.no_grad():
        tsd = {k: v.to(text_dtype) for k, v in tsd.items()}
        loss, logits = oq.causal_lm(emb.to(text_dtype), mask, labels, tsd, oq.Qwen3Cfg(vocab_size=tsd["model.embed_tokens.weight"].shape[0]), hs)
    out["t_text"] = time.perf_counter() - t0
    out["hidden"] = [h.float() for h in hs]
    out["logits"] = logits.float()
    out["loss"] = float(loss)
    return out


def _hip_text(model, st, B, L, H):
    tm = model.text_model
    Lp = st["saved"]["L"]
    hid = [c["h_in"].view(B, Lp, H)[:, :L].float().cpu() for c in st["saved"]["layers"][1:]]
    hid.append(st["h_last"].view(B, Lp, H)[:, :L].float().cpu())
    logits = tm.logits_all(st["h_last"]).view(B, Lp, -1)[:, :L].float().cpu()
    return {"hidden": hid, "logits": logits, "loss": float(st["loss"].item())}


def _compare_text(hip, ref16, ref32, valid):
    """Error triples per layer / logits / loss + the greedy-choice statistics."""
    rep = {}
    curve = []
    for i in range(len(hip["hidden"])):
        a, r16, r32 = hip["hidden"][i][valid], ref16["hidden"][i][valid], ref32["hidden"][i][valid]
        curve.append({"layer": i, "hip_vs_ref16": relerr(a, r16), "hip_vs_ref32": relerr(a, r32), "ref16_vs_ref32": relerr(r16, r32)})
    rep["hidden_by_layer"] = curve
    lead = 0
    while lead < len(curve) and curve[lead]["hip_vs_ref16"] <= 1e-2:
        lead += 1
    rep["literal_1e-2_holds_through_layer"] = lead - 1        # (-1: not even after the first layer)
    rep["layer_where_ref16_itself_leaves_0.5e-2_of_ref32"] = next((c["layer"] for c in curve if c["ref16_vs_ref32"] > 0.5e-2), None)
    lh, l16, l32 = hip["logits"], ref16["logits"], ref32["logits"]
    rep["logits"] = {"hip_vs_ref16": relerr(lh[valid], l16[valid]), "hip_vs_ref32": relerr(lh[valid], l32[valid]),
                     "ref16_vs_ref32": relerr(l16[valid], l32[valid])}
    for r in range(valid.shape[0]):                            # per row: collator layout / fully attended span
        v = valid[r]
        rep["logits_row%d" % r] = {"hip_vs_ref16": relerr(lh[r][v], l16[r][v]), "hip_vs_ref32": relerr(lh[r][v], l32[r][v]),
                                   "ref16_vs_ref32": relerr(l16[r][v], l32[r][v])}
    am_hip, am16, am32 = lh[valid].argmax(-1), l16[valid].argmax(-1), l32[valid].argmax(-1)
    top2 = l32[valid].topk(2, -1).values
    gap = top2[:, 0] - top2[:, 1]
    noise = (l16[valid] - l32[valid]).pow(2).mean(-1).sqrt()    # the reference's own bf16 RMS logit error per position
    d = am_hip != am32
    rep["argmax"] = {"positions": int(valid.sum()), "hip_ne_ref32": int(d.sum()), "ref16_ne_ref32": int((am16 != am32).sum()),
                     "hip_ne_ref16": int((am_hip != am16).sum()),
                     "max_gap_over_ref16_noise_where_hip_differs": float((gap[d] / noise[d]).max()) if d.any() else 0.0}
    rep["loss"] = {"hip": hip["loss"], "ref16": ref16["loss"], "ref32": ref32["loss"],
                   "hip_vs_ref16_rel": abs(hip["loss"] - ref16["loss"]) / abs(ref16["loss"]),
                   "hip_vs_ref32_rel": abs(hip["loss"] - ref32["loss"]) / abs(ref32["loss"]),
                   "ref16_vs_ref32_rel": abs(ref16["loss"] - ref32["loss"]) / abs(ref32["loss"])}
    return rep


SLACK, FLOOR = 1.25, 1e-3


def _as_good_as_reference_bf16(name, r):
    assert r["hip_vs_ref32"] <= SLACK * r["ref16_vs_ref32"] + FLOOR, \
        f"{name}: HIP is further from the exact value ({r['hip_vs_ref32']:.4g}) than the reference's bf16 CPU evaluation ({r['ref16_vs_ref32']:.4g})"
    assert r["hip_vs_ref16"] <= r["hip_vs_ref32"] + r["ref16_vs_ref32"] + 1e-6, name


def _assert_text(tag, rep):
    for name in ("logits", "logits_row0", "logits_row1"):
        _as_good_as_reference_bf16(f"{tag} {name}", rep[name])
    for c in rep["hidden_by_layer"]:
        _as_good_as_reference_bf16(f"{tag} hidden after layer {c['layer']}", c)
    assert rep["loss"]["hip_vs_ref32_rel"] <= SLACK * rep["loss"]["ref16_vs_ref32_rel"] + 2e-3, (tag, rep["loss"])
    assert rep["loss"]["hip_vs_ref16_rel"] <= 5e-3, (tag, rep["loss"])                 # (held in both regimes)
    a = rep["argmax"]
    assert a["hip_ne_ref32"] <= 1.5 * a["ref16_ne_ref32"] + 3, (tag, a)
    assert a["max_gap_over_ref16_noise_where_hip_differs"] <= 6.0, (tag, a)
    # the north star's literal statement, asserted wherever bf16 evaluation of this network is itself reproducible to half of it: on the
    # logits, and layer by layer as long as the reference's own bf16 evaluation stays within 0.5e-2 of the exact one
    if rep["logits"]["ref16_vs_ref32"] <= 0.5e-2:
        assert rep["logits"]["hip_vs_ref16"] <= 1e-2, (tag, rep["logits"])
    for c in rep["hidden_by_layer"]:
        if c["ref16_vs_ref32"] <= 0.5e-2:
            assert c["hip_vs_ref16"] <= 1e-2, (tag, c)


def test_full_depth_forward_vs_cpu_oracle():
    import yaml
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
    pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
    cfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=0,
                               projector_cfg=pcfg, text_config=Qwen3Config.qwen3_4b(), device="cuda", seed=0)
    model = VGGTQwen3VLM(cfg).train()
    model.projector.eval()                                   # the reference's CPU forward is an eval-mode run (no dropout)
    tm = model.text_model
    assert tm.config.num_hidden_layers == 36 and model.vision_model.aggregator.depth == 24 and pcfg.num_layers == 6
    b = _batch(model)
    B, L, H = 2, 200, tm.config.hidden_size

    # ---------------- HIP path (through the C ABI), every intermediate kept
    with torch.no_grad():
        tok_hip = model._vision_tokens(b["pixel_values"]).float().cpu()
        vis_hip = model.encode_images(b["pixel_values"]).float().cpu()
    st = model.forward_state(b["pixel_values"], None, b["input_ids"], b["attention_mask"], b["labels"], need_grad=True)
    assert st["saved"]["L"] == L
    hipA = _hip_text(model, st, B, L, H)
    emb_dev = st["emb"].view(B, L, H).clone()
    emb_hip = emb_dev.float().cpu()
    del st

    # ---------------- host copies of the very same weights (bf16 values)
    vsd = {n: t.detach().cpu() for n, t in model.vision_model.aggregator.named_tensors().items()}
    psd = {k: v.detach().float().cpu() for k, v in model.projector.state_dict().items()}
    tsd = {n: p.detach().cpu() for n, p in tm.named_parameters() if n != "lm_head.weight"}
    images, ids = b["pixel_values"].cpu(), b["input_ids"].cpu()
    mask, labels = b["attention_mask"].cpu(), b["labels"].cpu()
    valid = mask.bool()
    torch.cuda.empty_cache()

    # ================= regime A: end to end, the model's own initialisation
    vcommon = (images, ids, vsd, psd, tsd["model.embed_tokens.weight"], model.image_id, pcfg.num_heads, pcfg.num_layers)
    v16 = _oracle_vision(*vcommon, BF16, F32, BF16)            # the reference's CPU forward (config C1)
    v32 = _oracle_vision(*vcommon, F32, F32, F32)              # exact evaluation of the same network
    t16 = _oracle_text(v16["emb"], mask, labels, tsd, BF16)
    t32 = _oracle_text(v32["emb"], mask, labels, tsd, F32)
    rep = {"shape": {"B": B, "V": 1, "L": L, "qwen_layers": 36, "tower_blocks": 72, "perceiver_layers": 6, "vocab": int(hipA["logits"].shape[-1])},
           "cpu_threads": torch.get_num_threads(),
           "cpu_seconds_forward_2_samples": {"ref16": {"tower": round(v16["t_tower"], 2), "perceiver": round(v16["t_perceiver"], 2), "text": round(t16["t_text"], 2)},
                                             "ref32": {"tower": round(v32["t_tower"], 2), "perceiver": round(v32["t_perceiver"], 2), "text": round(t32["t_text"], 2)}}}

    def three(name, hip, sel=None):
        a, r16, r32 = hip, v16[name], v32[name]
        if sel is not None:
            a, r16, r32 = a[sel], r16[sel], r32[sel]
        return {"hip_vs_ref16": relerr(a, r16), "hip_vs_ref32": relerr(a, r32), "ref16_vs_ref32": relerr(r16, r32)}

    A = {"tower_tokens": three("tower_tokens", tok_hip), "vis_tokens": three("vis_tokens", vis_hip),
         "inputs_embeds": three("inputs_embeds", emb_hip, valid)}
    A.update(_compare_text(hipA, t16, t32, valid))
    rep["A_end_to_end_hf_init"] = A
    del t16, t32, v16, v32

    # ================= regime B: the text model alone, depth-scaled residual projections, same inputs_embeds for all three
    scale = 1.0 / (2 * 36) ** 0.5
    with torch.no_grad():
        for i in range(36):
            tm._w[f"l{i}.o"].mul_(scale)
            tm._w[f"l{i}.down"].mul_(scale)
    tm.refresh_derived()
    h_last, saved = tm.forward_hidden(emb_dev, b["attention_mask"], save=True)
    lossB, _ = tm.loss_head(h_last, b["labels"], save=False, L=saved["L"])
    hipB = _hip_text(model, {"saved": saved, "h_last": h_last, "loss": lossB}, B, L, H)
    tsdB = {n: p.detach().cpu() for n, p in tm.named_parameters() if n != "lm_head.weight"}
    embB = emb_dev.cpu()
    del saved, h_last, tsd
    Bd = _compare_text(hipB, _oracle_text(embB, mask, labels, tsdB, BF16), _oracle_text(embB, mask, labels, tsdB, F32), valid)
    rep["B_text_model_depth_scaled_init"] = Bd

    out_dir = ROOT / "gpurun_out"
    try:
        out_dir.mkdir(exist_ok=True)
        (out_dir / "r5_depth_parity.json").write_text(json.dumps(rep, indent=1))
    except OSError:
        pass
    for tag in ("A_end_to_end_hf_init", "B_text_model_depth_scaled_init"):
        r = rep[tag]
        print(tag, json.dumps({k: v for k, v in r.items() if k != "hidden_by_layer"}))
        print(tag, "depth curve (layer: hip-ref16 / hip-ref32 / ref16-ref32):",
              " ".join("%d:%.4f/%.4f/%.4f" % (c["layer"], c["hip_vs_ref16"], c["hip_vs_ref32"], c["ref16_vs_ref32"]) for c in r["hidden_by_layer"]))

    # ---------------- assertions
    assert all(torch.isfinite(t).all() for t in (hipA["logits"], hipB["logits"], vis_hip, tok_hip))
    for name in ("tower_tokens", "vis_tokens", "inputs_embeds"):
        _as_good_as_reference_bf16(name, A[name])
    _assert_text("A", A)
    _assert_text("B", Bd)
    # the literal 1e-2 holds for as many layers as bf16 evaluation itself allows - and for at least the depths the goldens pin (2 layers)
    assert A["literal_1e-2_holds_through_layer"] >= 0 and Bd["literal_1e-2_holds_through_layer"] >= 6, \
        (A["literal_1e-2_holds_through_layer"], Bd["literal_1e-2_holds_through_layer"])
