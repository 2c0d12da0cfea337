"""OPT-IN (model.eval_graphs / VQ3_EVAL_GRAPH=1; measured slower than stream launches for this path, see vlm.py): eval-mode forwards
replay HIP graphs of their two static parts (vlm.py: _graph_call; round 5): the tower + projector for an image
shape, the decoder layers for a (B, L). The reference's unit is VGGTQwen3VLM.forward under no_grad / .eval()
(/root/reference/src/models/vggt_qwen3_vlm.py:179-201, as its evaluation callers use it). Checked here: graph replay gives the eager
forward's loss for every batch (inputs are copied into the graph's static buffers), a changed weight is seen, replaced derived tensors
(the projector's bf16 compute copies) drop the graph, training-mode forwards never take the path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM, VisionLanguageConfig
    qcfg = Qwen3Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                       intermediate_size=256, vocab_size=300)
    pcfg = PerceiverConfig(latent_dim=128, num_latents=16, num_heads=2, num_layers=2, ffn_dim=256, dropout=0.1)
    cfg = VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=16, geom_tokens=4, projector_cfg=pcfg,
                               text_config=qcfg, vision_config=dict(img_size=70, embed_dim=128, depth=2, dino_depth=2), device="cuda:0", seed=3)
    return VGGTQwen3VLM(cfg)


def _batch(seed, B=3, V=2, L=40):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, V, 3, 56, 56, generator=g)
    ids = torch.randint(1, 300, (B, L), generator=g)
    ids[:, 30:] = 0
    labels = torch.full((B, L), -100)
    for r in range(B):
        ids[r, 5 + r] = 300                                  # <image>
        labels[r, 27:30] = ids[r, 27:30]
    geom = {"R": torch.randn(B, V, 9, generator=g), "t": torch.randn(B, V, 3, generator=g), "K": torch.randn(B, V, 9, generator=g),
            "depth_hist": torch.rand(B, V, 16, generator=g)}
    return dict(images=images.cuda(), geom_token={k: v.cuda() for k, v in geom.items()}, input_ids=ids.cuda(),
                attention_mask=(ids != 0).long().cuda(), labels=labels.cuda())


def test_eval_forward_graph_replay_equals_eager():
    model = _model()
    assert model.image_id == 300
    model.eval()
    batches = [_batch(s) for s in (1, 2, 3)]
    with torch.no_grad():
        model.eval_graphs = False
        want = [float(model(**b)) for b in batches]
        assert not model._graphs
        model.eval_graphs = True
        got = []
        for rnd in range(3):                                  # round 0: first sight (eager), 1: capture + replay, 2: replay
            got.append([float(model(**b)) for b in batches])
    assert len(set(want)) == 3                                # (the batches do differ)
    for rnd in range(3):
        for a, b in zip(got[rnd], want):
            assert abs(a - b) <= 1e-5 * abs(b), (rnd, got, want)
    assert len(model._graphs) == 2 and all(e[1] is not None and e[0] >= 8 for e in model._graphs.values())
    # a weight changed in place is seen by the replay (the graph reads the resident buffers)
    with torch.no_grad():
        model.text_model._w["l0.o"].mul_(0.5)
        a = float(model(**batches[0]))
        model.eval_graphs = False
        b = float(model(**batches[0]))
        model.eval_graphs = True
    assert abs(a - b) <= 1e-5 * abs(b) and abs(a - want[0]) > 1e-4 * abs(want[0])
    # replaced derived tensors drop the graph: new projector weights -> new bf16 compute copies
    sd = {k: v.clone() for k, v in model.projector.state_dict().items()}
    sd["out_proj.weight"] = sd["out_proj.weight"] * 1.5
    model.projector.load_state_dict(sd)
    with torch.no_grad():
        c = float(model(**batches[0]))
        model.eval_graphs = False
        d = float(model(**batches[0]))
        model.eval_graphs = True
    assert abs(c - d) <= 1e-5 * abs(d) and abs(c - a) > 1e-5 * abs(a)
    # training-mode forwards (dropout in the projector, autograd) never replay
    model.train()
    n = {k: e[0] for k, e in model._graphs.items()}
    loss = model(**batches[1])
    loss.backward()
    assert {k: e[0] for k, e in model._graphs.items()} == n
