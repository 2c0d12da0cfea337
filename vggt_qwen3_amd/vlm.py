"""VGGTQwen3VLM on the HIP kernels - drop-in for the reference's src/models/vggt_qwen3_vlm.py:15-201.

Same dataclass fields, constructor, attribute names (.tokenizer .text_model .vision_model .projector .geom_head
.num_vis_tokens .geom_tokens), parameter names and `forward(images, geom_token, input_ids, attention_mask, labels)
-> loss` contract, including the reference's quirks: only the first `num_vis_tokens` aggregator
tokens are used, the visual span OVERWRITES the rows after <image>, encode_images runs without gradient, and a span
that overruns the sequence raises RuntimeError.

The loss returned by forward() is attached to autograd through one custom Function whose backward runs the
hand-written HIP backward of the whole text model (+ geom_head), so `accelerator.backward(loss)` /
`loss.backward()` in the reference trainer (train_sft.py:217) works unchanged.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .hostplan import PLAN
from .ops import BF16, F32
from .perceiver import PerceiverConfig, PerceiverProjector
from .qwen3 import Qwen3Config, Qwen3ForCausalLM

GEOM_FEATURE_DIM = 37  # R(9) + t(3) + K(9) + depth_hist(16)   (vggt_qwen3_vlm.py:51)


@dataclass
class VisionLanguageConfig:
    text_model_name: str
    vision_ckpt_dir: str
    num_vis_tokens: int = 64
    geom_tokens: int = 0
    projector_cfg: Optional[PerceiverConfig] = None
    freeze_vision: bool = True
    dtype: str = "bfloat16"
    # --- extensions (all optional; the reference's 7 fields above are unchanged) ---
    text_config: Optional[Qwen3Config] = None    # random-init text model of this shape (no files, no network)
    vision_config: Optional[dict] = None         # kwargs for vggt_qwen3_amd.vggt.VGGT (e.g. reduced depth in tests)
    vision_module: Optional[nn.Module] = None    # inject any module exposing .aggregator(images) and .embed_dim
    device: str = "cuda"
    seed: int = 0
    trim_padding: bool = False
    fp8_text_forward: bool = False               # BASELINE config C5: e4m3 forward GEMMs in the text model
    # False (default) = the reference as it runs: encode_images under no_grad, the Perceiver never receives a gradient although
    # train_sft.py:138-145 gives it a learning-rate group. True = the "corrected" mode of SURVEY.md 3.1: the projector's forward saves
    # for backward, d(loss)/d(visual tokens) flows into its hand-written backward, the trainer updates it with proj_lr.
    train_projector: bool = False


def build_srcmap(input_ids: torch.Tensor, image_id: int, S: int) -> torch.Tensor:
    """Integer image of `for b, pos in nonzero(ids == image_id): emb[b, pos:pos+S] = features[b]`
    (vggt_qwen3_vlm.py:191-195): srcmap[b,l] = feature row written at (b,l) or -1; the last writer wins for repeated
    <image> tokens; raises RuntimeError like the reference's index assignment when a span overruns the sequence."""
    B, L = input_ids.shape
    pos = (input_ids == image_id).nonzero(as_tuple=False).tolist()
    m = torch.full((B, L), -1, dtype=torch.int32)
    for b, p in pos:
        if p + S > L:
            raise RuntimeError(f"The expanded size of the tensor ({L - p}) must match the existing size ({S}) at "
                               f"non-singleton dimension 0 (visual span at position {p} overruns L={L})")
        m[b, p:p + S] = torch.arange(S, dtype=torch.int32)
    return m.to(input_ids.device)


class _StubTokenizer:
    """Used only when text_model_name has no tokenizer files (synthetic benchmarks): knows `<image>`."""

    def __init__(self, vocab_size: int):
        self.vocab = {"<pad>": 0}
        self._n = vocab_size
        self.pad_token_id = 0

    def get_vocab(self):
        return dict(self.vocab)

    def add_tokens(self, toks):
        for t in toks:
            self.vocab[t] = self._n
            self._n += 1
        return len(toks)

    def convert_tokens_to_ids(self, t):
        return self.vocab[t]

    def __len__(self):
        return self._n


class _GeomLinear(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        lin = nn.Linear(fin, fout)  # default torch init, like the reference's nn.Sequential(nn.Linear, ...)
        self.weight, self.bias = lin.weight, lin.bias


class _TextLossFn(torch.autograd.Function):
    """Connects the hand-written forward/backward to autograd: inputs are the trainable parameters (so their .grad
    gets populated) and the geom features; output is the scalar loss."""

    @staticmethod
    def forward(ctx, model: "VGGTQwen3VLM", state: dict, geom_feats, *params):
        ctx.model, ctx.state = model, state
        ctx.has_geom = geom_feats is not None
        return state["loss"].clone()

    @staticmethod
    def backward(ctx, grad_out):
        model, st = ctx.model, ctx.state
        tm = model.text_model
        gscale = float(grad_out.item())
        world, group = model._autograd_dp_world()
        if world > 1:
            return _TextLossFn._backward_multi_rank(ctx, model, st, gscale, world, group)
        # The parameters' .grad ARE views of the flat gradient buffer (no 8 GB copy per backward): autograd gets None for
        # them and the HIP backward accumulates in place, exactly when .grad is still our view from an earlier backward
        # (optimizer.zero_grad(set_to_none=True) drops the views -> the next backward overwrites; set_to_none=False zeroes
        # the buffer through the views -> accumulation into zeros).
        first = next(iter(tm.parameters()))
        live = first.grad is not None and first.grad.data_ptr() == tm.grad_views["model.embed_tokens.weight"].data_ptr()
        d_geom = model._backward_text(st, gscale, accumulate=live)
        tm.publish_grads()
        return (None, None, d_geom if ctx.has_geom else None, *([None] * len(model._text_param_names)))


    @staticmethod
    def _backward_multi_rank(ctx, model, st, gscale, world, group):
        """loss.backward() with torch.distributed initialised over more than one rank - the reference's own multi-GPU route
        (train_sft.py:119-133 Accelerator, :165-170 prepare -> DistributedDataParallel(find_unused_parameters=True), :217 backward).
        The text parameters never pass through autograd (their .grad are views of the flat buffer, filled by the HIP backward), so DDP's
        reducer cannot see them: it would mark all 4 B of them unused and every rank would train on its own gradient. This route therefore
        does what DDP does for them, itself: the gradient of THIS backward is all-reduced (SUM, / world: DDP's average) over the flat
        buffer before it joins what earlier backwards of the window left there. geom_head's fp32 parameters are ordinary autograd leaves:
        DDP (or the caller) reduces them as for any module. `model.autograd_dp`: "allreduce" (default), "raise", or "local" (the caller
        owns the exchange - Stage1Trainer never comes through here, it drives forward_state / _backward_text with its own bucketed,
        overlapped all-reduce and is the fast path)."""
        from . import dp
        tm = model.text_model
        mode = model.autograd_dp
        if mode == "raise" or (mode == "allreduce" and st.get("pctx") is not None):
            raise RuntimeError(
                "VGGTQwen3VLM: loss.backward() under torch.distributed with world_size %d: the text model's gradients live in a flat buffer "
                "that DistributedDataParallel's reducer never sees - replicas would diverge silently. Use vggt_qwen3_amd.trainer.Stage1Trainer "
                "(bucketed RCCL all-reduce inside the backward), or set model.autograd_dp = 'allreduce' (this route reduces the flat gradient "
                "itself; not available with train_projector=True) / 'local' (you reduce text_model.flat_g yourself)." % world)
        first = next(iter(tm.parameters()))
        live = first.grad is not None and first.grad.data_ptr() == tm.grad_views["model.embed_tokens.weight"].data_ptr()
        if mode == "local":
            d_geom = model._backward_text(st, gscale, accumulate=live)
        else:
            keep = tm.flat_g.clone() if live else None          # what earlier backwards of the window accumulated (already reduced)
            d_geom = model._backward_text(st, gscale, accumulate=False)
            torch.cuda.current_stream().synchronize()
            dp.allreduce_tensor(tm.flat_g, group=group)         # every rank comes here once per backward, labelled rows or not
            tm.flat_g.mul_(1.0 / world)
            if keep is not None:
                tm.flat_g.add_(keep)
        tm.publish_grads()
        return (None, None, d_geom if ctx.has_geom else None, *([None] * len(model._text_param_names)))


class VGGTQwen3VLM(nn.Module):
    def __init__(self, config: VisionLanguageConfig) -> None:
        super().__init__()
        dev = torch.device(config.device)
        self.device_ = dev
        # ---- text model + tokenizer (vggt_qwen3_vlm.py:31-42)
        if config.text_config is not None:
            self.text_model = Qwen3ForCausalLM(config.text_config, device=dev, seed=config.seed)
            self.tokenizer = self._load_tokenizer(config.text_model_name, self.text_model.vocab)
        else:
            path = Path(config.text_model_name)
            if not path.is_dir():
                raise FileNotFoundError(
                    f"text_model_name={config.text_model_name!r} is not a local directory; this build never fetches "
                    "from the network. Pass a directory with config.json + *.safetensors, or text_config=...")
            self.text_model = Qwen3ForCausalLM.from_pretrained_dir(path, device=dev)
            self.tokenizer = self._load_tokenizer(config.text_model_name, self.text_model.vocab)
        added = 0
        if "<image>" not in self.tokenizer.get_vocab():
            added = self.tokenizer.add_tokens(["<image>"])
        if added:
            self.text_model.resize_token_embeddings(len(self.tokenizer))
        self.image_id = self.tokenizer.convert_tokens_to_ids("<image>")
        # ---- vision tower (vggt_qwen3_vlm.py:43-45,60-111)
        self.vision_model = self._load_vggt(config)
        for p in self.vision_model.parameters():
            p.requires_grad_(not config.freeze_vision)
        # ---- projector + geom head (vggt_qwen3_vlm.py:46-56): fp32 parameters like the reference
        H = self.text_model.config.hidden_size
        g = torch.Generator().manual_seed(config.seed + 1)
        with torch.random.fork_rng():
            torch.manual_seed(config.seed + 1)
            self.projector = PerceiverProjector(config.projector_cfg or PerceiverConfig(),
                                                in_dim=self.vision_model.embed_dim, out_dim=H)
            self.geom_head = nn.ModuleList([_GeomLinear(GEOM_FEATURE_DIM, H), nn.SiLU(), _GeomLinear(H, H)])
        self.projector.to(dev)
        self.geom_head.to(dev)
        self.num_vis_tokens = config.num_vis_tokens
        self.geom_tokens = config.geom_tokens
        # False = compute all L positions like the reference; True = drop the all-padding tail of the batch (same
        # loss and gradients, fewer rows). Off by default so the dense figure stays comparable with the reference's.
        self.trim_padding = bool(config.trim_padding)
        self.train_projector = bool(config.train_projector)
        if config.fp8_text_forward:
            self.text_model.enable_fp8_forward(True)
        self._vis_stream = None
        self._vision_head = os.environ.get("VQ3_VISION_HEAD", "1") != "0"
        # OPT-IN (VQ3_EVAL_GRAPH=1 / model.eval_graphs = True; default off): eval-mode forwards (no_grad, .eval(): the reference's
        # inference / evaluation callers) replay HIP graphs of their two static parts - the tower + projector for an image shape, the 36
        # decoder layers for a (B, L) - captured at the second sight of a shape; the data-dependent glue between them (splice map,
        # labelled rows, loss) stays eager. Measured (round 5, bench.py forward_only): SLOWER than stream launches - 40.7 against 34.3 ms
        # per 6-sample forward, 195.3 against 190.5 ms at 48 samples: these ~750 kernels are 20-400 us each, the host is far ahead of the
        # device, and a hipGraph launch serialises the tower's two-stream row-tail chains; graphs pay where the host is the bottleneck
        # (the decode step, generate.py), not here. Kept because it is exact and tested (tests/test_eval_graph_gpu.py).
        self.eval_graphs = os.environ.get("VQ3_EVAL_GRAPH", "0") == "1"
        self._graphs = {}             # key -> [sightings, graph, static inputs, static output]
        self._graph_stream = None
        self._prefetched = None
        self._weights_gate = None        # event of an optimiser step still running on Stage1Trainer's side stream (trainer.py)
        self._vis_group = []          # [(images tensor, aggregator tokens)]: precompute_vision() results waiting for their micro-batch
        self._text_param_names = [n for n, _ in self.text_model.named_parameters()]
        # loss.backward() with torch.distributed initialised over > 1 rank (the reference's Accelerate / DDP route): see
        # _TextLossFn._backward_multi_rank. "allreduce" | "raise" | "local"; the process group is `autograd_dp_group` (None = WORLD)
        self.autograd_dp = os.environ.get("VQ3_AUTOGRAD_DP", "allreduce")
        self.autograd_dp_group = None

    def _autograd_dp_world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1, None
        return dist.get_world_size(self.autograd_dp_group), self.autograd_dp_group

    def _apply(self, fn, recurse=True):
        """`.to(...)`, `.cpu()`, `.cuda()`, `.float()`, `.bfloat16()` are no-ops: the weights live in HBM in the flat,
        GEMM-ready layout they were allocated in (config.device / config.dtype decide that at construction). Callers
        written for the reference - `VGGTQwen3VLM(cfg).to(device)` (qa_inference.py:44), the checkpoint loader's
        `model.to("cpu") ... model.to(orig_device)` (qa_inference.py:60,104) - therefore work unchanged;
        `load_state_dict` copies host tensors into the resident parameters."""
        return self

    # ------------------------------------------------------------------ loaders
    @staticmethod
    def _load_tokenizer(name: str, vocab: int):
        p = Path(name)
        if p.is_dir() and any((p / f).exists() for f in ("tokenizer.json", "tokenizer_config.json", "vocab.json")):
            from transformers import AutoTokenizer  # tokenizer only: host-side plumbing, local files only
            return AutoTokenizer.from_pretrained(str(p), local_files_only=True)
        return _StubTokenizer(vocab)

    def _load_vggt(self, config: VisionLanguageConfig) -> nn.Module:
        if config.vision_module is not None:
            return config.vision_module
        from .vggt import VGGT
        kw = dict(img_size=518, patch_size=14, embed_dim=1024)
        kw.update(config.vision_config or {})
        model = VGGT(device=self.device_, seed=config.seed + 2, **kw)
        ckpt = Path(config.vision_ckpt_dir) / "vggt_1B_commercial.pt"
        if ckpt.exists():
            sd = torch.load(ckpt, map_location="cpu")
            if isinstance(sd, dict):
                sd = sd.get("model", sd.get("state_dict", sd))
            model.load_reference_state_dict(sd)
        else:
            print(f"Warning: checkpoint file {ckpt} not found, using random initialization")
        model.eval()
        model.embed_dim = 2 * model.aggregator.embed_dim  # frame|global concat (vggt_qwen3_vlm.py:107-109)
        return model

    # ------------------------------------------------------------------ encoders
    def _aggregate(self, images: torch.Tensor) -> torch.Tensor:
        """aggregated_tokens_list[-1] of the tower (vggt_qwen3_vlm.py:144-147). Our aggregator can stop at the rows the slicing below
        keeps (Aggregator.forward_head: the last global block's proj / MLP for the first num_vis_tokens rows of a sample only - an exact
        shortcut, VQ3_VISION_HEAD=0 computes every row); any other tower is called the reference's way."""
        agg_mod = self.vision_model.aggregator
        if self._vision_head and hasattr(agg_mod, "forward_head"):
            return agg_mod.forward_head(images, self.num_vis_tokens)
        toks, _ = agg_mod(images)
        return toks[-1]

    @torch.no_grad()
    def precompute_vision(self, images_list) -> None:
        """The frozen aggregator (vggt_qwen3_vlm.py:44-45,128-144: no_grad, eval-mode arithmetic, no dropout) for SEVERAL upcoming
        micro-batches in one pass: their images are concatenated along the batch axis - every sample's frame / global attention and
        every token row of the GEMMs is computed exactly as in its own micro-batch - so that the tower's GEMMs run at 4 x 6174 rows
        instead of 6174 (at 8 x: fc1 713 vs 620 TF/s, q|k|v 711 vs 553, proj 634 vs 557; whole rounds of 256 x 256 tiles). Results wait in
        FIFO order; encode_images() of the SAME tensor object picks its slice up and runs only the (train-mode dropout) projector."""
        imgs = [im.to(self.device_) for im in images_list]
        if not imgs:
            return
        shapes = {tuple(im.shape[1:]) for im in imgs}
        if len(shapes) != 1:                               # different view counts / sizes cannot share a pass
            return
        agg = self._aggregate(torch.cat(imgs, dim=0) if len(imgs) > 1 else imgs[0])
        if agg.shape[0] != sum(im.shape[0] for im in imgs):    # (a tower that does not keep the batch axis first: no sharing)
            return
        b0 = 0
        for orig, im in zip(images_list, imgs):
            self._vis_group.append((orig, agg[b0:b0 + im.shape[0]]))
            b0 += im.shape[0]

    def _take_grouped(self, images: torch.Tensor):
        for k, (im, agg) in enumerate(self._vis_group):
            if im is images:
                del self._vis_group[k]
                # precompute_vision() allocated the slice's block on the stream it ran on; a consumer on another stream
                # (prefetch_images -> _vis_stream) must be known to the caching allocator before the last slice is dropped
                if agg.is_cuda:
                    agg.record_stream(torch.cuda.current_stream())
                return agg
        return None

    @torch.no_grad()
    def _vision_tokens(self, images: torch.Tensor, _orig: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The frozen tower's last output, first num_vis_tokens tokens of view 0 (vggt_qwen3_vlm.py:144-156): [B, Nv, 2048]."""
        B = images.shape[0]
        agg = self._take_grouped(images if _orig is None else _orig) if self._vis_group else None
        if agg is None:
            agg = self._aggregate(images)
        if agg.dim() == 3:
            agg = agg[:, : self.num_vis_tokens, :]
        elif agg.dim() == 4:
            agg = agg.reshape(B, -1, agg.shape[-1])[:, : self.num_vis_tokens, :]
        return agg.contiguous()

    @torch.no_grad()
    def encode_images(self, images: torch.Tensor, _orig: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images [B, V, C, H, W] -> [B, num_vis_tokens, hidden] (vggt_qwen3_vlm.py:128-162)."""
        return self.projector(self._vision_tokens(images, _orig))

    def prefetch_images(self, images: torch.Tensor) -> None:
        """Run encode_images for a FUTURE batch on a second HIP stream: the vision tower is frozen and under no_grad
        in the reference (vggt_qwen3_vlm.py:44-45,128), so its forward for micro-batch t+1 is independent of the text
        model's forward/backward of micro-batch t and can fill the CUs those GEMMs leave idle. The result is picked up
        by forward_state() when it is called with the same tensor object."""
        if self._vis_stream is None:
            self._vis_stream = torch.cuda.Stream(device=self.device_)
        images = images.to(self.device_)
        self._vis_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._vis_stream):
            # "corrected" mode (train_projector): only the FROZEN tower's tokens may be computed ahead - the projector's weights change
            # with every optimiser step and its train-mode dropout offsets belong to the pass that consumes them
            if self.train_projector:
                kind, vis = "tokens", self._vision_tokens(images)
            else:
                kind, vis = "encoded", self.encode_images(images)
        self._prefetched = (images, vis, kind)

    def _pass_weights_gate(self) -> None:
        ev = getattr(self, "_weights_gate", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._weights_gate = None
            self.text_model._weights_gate = None

    def _take_prefetched(self, images: torch.Tensor):
        """(kind, tensor) of a prefetch_images() result for this very tensor object - kind "encoded": projector output, "tokens": the
        frozen tower's tokens - or None."""
        pf = self._prefetched
        if pf is None or pf[0] is not images:
            return None
        self._prefetched = None
        torch.cuda.current_stream().wait_stream(self._vis_stream)
        pf[1].record_stream(torch.cuda.current_stream())
        return pf[2], pf[1]

    def _geom_inputs(self, geom_token) -> Optional[torch.Tensor]:
        if not geom_token or self.geom_tokens == 0:
            return None
        feats = torch.cat([geom_token["R"], geom_token["t"], geom_token["K"], geom_token["depth_hist"]], dim=-1)
        return feats.to(self.device_, F32).mean(dim=1)  # [B, 37]: input plumbing (mean over views)

    def _geom_fwd(self, feats: torch.Tensor, save: bool):
        """geom_head (vggt_qwen3_vlm.py:51-56,164-177): Linear(37->H) - SiLU - Linear(H->H), then 8 identical tokens."""
        B = feats.shape[0]
        H = self.text_model.config.hidden_size
        x = torch.zeros((B, 64), device=feats.device, dtype=BF16)
        x[:, :GEOM_FEATURE_DIM] = feats.to(BF16)
        w0 = torch.zeros((H, 64), device=feats.device, dtype=BF16)
        w0[:, :GEOM_FEATURE_DIM] = ops.cast(self.geom_head[0].weight.detach().contiguous(), BF16)
        w2 = ops.cast(self.geom_head[2].weight.detach().contiguous(), BF16)
        z = ops.linear(x, w0, bias=self.geom_head[0].bias.detach(), out_dtype=F32)            # pre-activation
        a = ops.linear(x, w0, bias=self.geom_head[0].bias.detach(), act=ops.ACT_SILU)         # bf16 SiLU(z)
        y = ops.linear(a, w2, bias=self.geom_head[2].bias.detach(), out_dtype=F32)            # [B, H] fp32
        ctx = dict(x=x, z=z, a=a, w2=w2) if save else None
        return y, ctx

    def encode_geom(self, geom_token: Optional[Dict[str, torch.Tensor]]) -> Optional[torch.Tensor]:
        feats = self._geom_inputs(geom_token)
        if feats is None:
            return None
        y, _ = self._geom_fwd(feats, False)
        return y.unsqueeze(1).expand(-1, self.geom_tokens, -1)

    # ------------------------------------------------------------------ splice (integer work, host side)
    def _srcmap(self, input_ids: torch.Tensor, S: int) -> torch.Tensor:
        return PLAN.get(("srcmap", self.image_id, S), (input_ids,), lambda: build_srcmap(input_ids, self.image_id, S))

    # ------------------------------------------------------------------ eval-mode HIP graphs
    GRAPH_SLOTS = 4

    def _graph_call(self, key, inputs, fn, guard=lambda: ()):
        """fn(*inputs) -> tensor, through a HIP graph once the key has been seen before: first sight runs eagerly (on the graph stream, so
        that the GEMM tuner's measurements, the split-K workspace of that stream and every host-side cache exist before the capture),
        second sight captures with static input copies, later sights copy the inputs in and replay. The returned tensor is the graph's
        static output: valid until the next call with the same key (forward_state consumes it at once). Launch gaps between the ~750
        kernels of a forward are what this removes (2-3 us each: 6 % of a 6-sample forward)."""
        dev = inputs[0].device
        if self._graph_stream is None:
            self._graph_stream = torch.cuda.Stream(device=dev)
        ent = self._graphs.get(key)
        cur = torch.cuda.current_stream()
        # guard(): the derived tensors the captured launches read by ADDRESS (bf16 compute copies of the projector / tower, e4m3 weight
        # copies): a graph whose guard objects were replaced (a weight load, an optimiser step on the projector, a re-quantisation) is dropped
        if ent is not None and ent[1] is not None:
            now = guard()
            if len(now) != len(ent[4]) or any(a is not b for a, b in zip(now, ent[4])):
                del self._graphs[key]
                ent = None
        if ent is None:
            if len(self._graphs) >= self.GRAPH_SLOTS:                      # oldest shape goes (its pool is released with it)
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = [1, None, None, None, ()]
            self._graph_stream.wait_stream(cur)
            with torch.cuda.stream(self._graph_stream):
                out = fn(*inputs)
            cur.wait_stream(self._graph_stream)
            for t in inputs:
                t.record_stream(self._graph_stream)
            out.record_stream(cur)
            return out
        if ent[1] is None:
            static_in = [torch.empty_like(t) for t in inputs]
            for d_, s_ in zip(static_in, inputs):
                d_.copy_(s_)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self._graph_stream):
                out = fn(*static_in)
            ent[1], ent[2], ent[3], ent[4] = graph, static_in, out, guard()
        else:
            for d_, s_ in zip(ent[2], inputs):
                d_.copy_(s_)
        ent[0] += 1
        ent[1].replay()
        return ent[3]

    # ------------------------------------------------------------------ forward / backward
    def forward_state(self, images, geom_token, input_ids, attention_mask, labels, need_grad: bool, loss_groups=None,
                      graphs: bool = False) -> dict:
        """Runs the whole path on the HIP kernels and returns a state dict with `loss` (+ what backward needs).
        loss_groups (sample counts): the batch is several micro-batches concatenated; `loss` is then the vector of their losses and
        the backward differentiates their sum (Qwen3ForCausalLM.loss_head)."""
        tm = self.text_model
        H = tm.config.hidden_size
        input_ids = input_ids.to(self.device_)
        attention_mask = attention_mask.to(self.device_)
        labels = labels.to(self.device_)
        attention_mask0, labels0 = attention_mask, labels       # the caller's tensors: keys of the remembered host-side facts
        B, L = input_ids.shape
        images0 = images
        images = images.to(self.device_)
        pctx = None
        pf = self._take_prefetched(images)
        if self.train_projector and need_grad:
            # "corrected" mode: the tower stays frozen (no_grad), the projector's forward keeps what its backward needs
            with torch.no_grad():
                tok = pf[1] if (pf is not None and pf[0] == "tokens") else self._vision_tokens(images, _orig=images0)
                self._pass_weights_gate()               # the projector trains: its weights may still be in the optimiser's hands
                vis, pctx = self.projector.forward_train(tok)
        elif pf is not None and pf[0] == "encoded":
            vis = pf[1]
        elif pf is not None:
            with torch.no_grad():
                if self.train_projector:
                    self._pass_weights_gate()
                vis = self.projector(pf[1])
        elif self.train_projector:
            with torch.no_grad():
                tok = self._vision_tokens(images, _orig=images0)
                self._pass_weights_gate()
                vis = self.projector(tok)
        elif graphs and not self._vis_group and not self.projector.training:
            agg_mod = getattr(self.vision_model, "aggregator", None)
            vis = self._graph_call(("vis", tuple(images.shape), images.dtype), [images], lambda im: self.encode_images(im),
                                   guard=lambda: (getattr(self.projector, "_cc", None), getattr(agg_mod, "_cc", None)))
        else:
            vis = self.encode_images(images, _orig=images0)                               # [B, Nv, H] fp32
        # everything below reads trainable tensors (geom_head, the embedding, the text model): an optimiser step that Stage1Trainer left
        # running beside this pass's (frozen) vision tower has to be complete from here on
        self._pass_weights_gate()
        gfeat = self._geom_inputs(geom_token)
        geom_ctx, gy = None, None
        if gfeat is not None:
            gy, geom_ctx = self._geom_fwd(gfeat, need_grad)
            feats = torch.cat([gy.unsqueeze(1).expand(-1, self.geom_tokens, -1), vis], dim=1)
        else:
            feats = vis
        S = feats.shape[1]
        feats16 = ops.cast(feats.contiguous(), BF16)
        srcmap = self._srcmap(input_ids, S)          # on the full L: overruns raise exactly like the reference
        if self.trim_padding:
            # Exact shortcut: columns after the last attended / labelled position of the whole batch are padding for
            # every row - masked as keys, never read as queries, no label - so they cannot change loss or gradients.
            def last_used():
                used = ((attention_mask != 0) | (labels != -100)).any(dim=0).nonzero()
                return int(used.max().item()) + 1 if used.numel() else 1
            L_eff = PLAN.get("last_used_column", (attention_mask0, labels0), last_used)
            L_eff = min(L, max(8, (L_eff + 7) // 8 * 8))
            if L_eff < L:
                input_ids = input_ids[:, :L_eff].contiguous()
                attention_mask = attention_mask[:, :L_eff].contiguous()
                labels = labels[:, :L_eff].contiguous()
                srcmap = srcmap[:, :L_eff].contiguous()
                L = L_eff
        self._last_L = L
        emb = ops.embed_splice_fwd(input_ids.contiguous(), tm._w["embed"], feats16, srcmap, B, L, H, S)
        if graphs and not need_grad:
            Lp_ = (L + 7) // 8 * 8
            h_last = self._graph_call(("text", B, L, bool(getattr(tm, "_fp8", None) is not None)), [emb.view(B, L, H), attention_mask.contiguous()],
                                      lambda e, m: tm.forward_hidden(e, m, save=False)[0],
                                      guard=lambda: ((tm._fp8.get("l0.qkv", (None,))[0],) if getattr(tm, "_fp8", None) else ()))
            saved = {"layers": [], "B": B, "L": Lp_, "L0": L}
        else:
            h_last, saved = tm.forward_hidden(emb, attention_mask, save=need_grad, plan_key=(attention_mask0,))
        loss, head_ctx = tm.loss_head(h_last, labels, save=need_grad, L=saved["L"], plan_key=(labels0,), groups=loss_groups)
        live = None
        if need_grad:
            # positions whose embedding row CAN receive a gradient: attended as a key, or carrying a loss term (position t predicts
            # labels[t + 1], loss_utils.py:49-71). Every other row of d(embeds) is exactly zero (masked keys get exact-zero dK / dV,
            # all other ops are row-wise), and with the reference's collator that is ~85 % of a batch: the padding id's run.
            nxt = torch.full_like(labels, -100)
            nxt[:, :-1] = labels[:, 1:]
            live = (attention_mask != 0) | (nxt != -100)
        return dict(loss=loss, saved=saved, head=head_ctx, srcmap=srcmap, input_ids=input_ids, B=B, L=L, S=S,
                    geom_ctx=geom_ctx, geom_y=gy, emb=emb, h_last=h_last, pctx=pctx, live=live)

    def _backward_text(self, st: dict, gscale: float, accumulate: bool, layer_done=None, flush: bool = True):
        """Backward of everything that has gradients in the reference: Qwen3 (all parameters, tied embedding) and,
        when geometry tokens are present, geom_head. Returns d(loss)/d(geom_head output) or None."""
        tm = self.text_model
        H = tm.config.hidden_size
        B, L, S = st["B"], st["L"], st["S"]
        if st["head"] is None:
            # No labelled token in this micro-batch (the collator truncated the answer away): the reference's mean over zero
            # targets is NaN and would poison every weight; here the micro-batch contributes a ZERO gradient. The window's
            # first micro-batch still has to overwrite flat_g, and the DP hooks still have to fire once per layer so that
            # every rank issues the same sequence of bucket all-reduces.
            if not accumulate:
                tm.zero_grad_flat()
            if flush and any(tm._wd_rows):
                tm.flush_deferred(layer_done)        # earlier micro-batches of the window still wait for their weight-gradient GEMMs
            elif layer_done is not None:
                for i in reversed(range(tm.config.num_hidden_layers)):
                    layer_done(i)
            return None
        Lp = st["saved"]["L"]                                   # forward_hidden pads L to a multiple of 8
        dh = tm.backward_loss_head(st["head"], B * Lp, gscale, accumulate)
        d_emb = tm.backward_hidden(st["saved"], dh, accumulate, layer_done=layer_done, flush=flush)
        if Lp != L:
            d_emb = d_emb.view(B, Lp, H)[:, :L].contiguous()
        ids = st["input_ids"].reshape(-1)
        dfeat = None
        if st["geom_ctx"] is not None or st.get("pctx") is not None:
            dfeat = torch.zeros((B, S, H), device=d_emb.device, dtype=F32)
        live = st.get("live")
        if live is None or os.environ.get("VQ3_EMBED_BWD_LIVE", "1") == "0":
            sorted_ids, order = torch.sort(ids, stable=True)
            ops.embed_splice_bwd(sorted_ids, order, st["srcmap"], d_emb, tm._g["embed"], dfeat, B, L, H, S)
        else:
            # Rows that cannot carry a gradient leave the id runs (index plumbing, no arithmetic): each gets a sort key of its own
            # behind every real id - a run of one, marked "skip" in the table pass's source map - so the run of the padding id, which
            # one workgroup per 512 columns used to sum serially over ~8 000 all-zero rows (1.2 ms per pass), is gone. Same sums, same
            # order inside every remaining run. The feature gradient (second call) still sees the real source map.
            live = live.reshape(-1)
            pos = torch.arange(ids.numel(), device=ids.device, dtype=ids.dtype)
            keys = torch.where(live, ids, pos + (1 << 40))
            sorted_keys, order = torch.sort(keys, stable=True)
            srcmap_tbl = torch.where(live, st["srcmap"].reshape(-1), torch.zeros_like(st["srcmap"].reshape(-1)))
            ops.embed_splice_bwd(sorted_keys, order, srcmap_tbl, d_emb, tm._g["embed"], None, B, L, H, S)
            if dfeat is not None:
                ops.embed_splice_bwd(sorted_keys, order, st["srcmap"], d_emb, None, dfeat, B, L, H, S)
        if dfeat is None:
            return None
        if st.get("pctx") is not None:
            # the visual rows' gradient goes on into the Perceiver (fp32 gradients accumulate in its parameters' .grad)
            ng = self.geom_tokens if st["geom_ctx"] is not None else 0
            with torch.no_grad():
                self.projector.backward(st["pctx"], dfeat[:, ng:].contiguous())
            st["pctx"] = None
        if st["geom_ctx"] is None:
            return None
        return dfeat[:, : self.geom_tokens].sum(dim=1)  # the 8 geom tokens are one expanded vector

    def geom_head_backward(self, st: dict, d_y: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Gradients of geom_head's four tensors given d(loss)/d(output) [B, H] fp32 (tiny: M = batch size)."""
        g = st["geom_ctx"]
        x, z, a, w2 = g["x"], g["z"], g["a"], g["w2"]
        B, H = d_y.shape
        dev = d_y.device
        dy16 = ops.cast(d_y.contiguous(), BF16)
        dyt, at_, xt = ops.transpose2d(dy16, 64), ops.transpose2d(a, 64), ops.transpose2d(x, 64)
        ones = torch.zeros((8, 64), device=dev, dtype=BF16)
        ones[0, :B] = 1.0                                     # row 0 sums over the batch inside the GEMM
        # layer 2: dW2 = dy^T a ; db2 = sum_b dy ; da = dy W2
        dW2 = torch.empty((H, H), device=dev, dtype=F32)
        ops.gemm_raw(dyt, at_, dW2, H, H, 64, 64, 64, H)
        db2 = torch.empty((8, H), device=dev, dtype=F32)
        ops.gemm_raw(ones, dyt, db2, 8, H, 64, 64, 64, H)
        da = ops.linear(dy16, ops.transpose2d(w2, 64))         # [B, H] bf16
        # SiLU backward through the SwiGLU kernel with up == 1: dgate = da * silu'(z)
        gu = torch.ones((B, 2 * H), device=dev, dtype=BF16)
        gu[:, :H] = ops.cast(z, BF16)
        dz16 = ops.silu_mul_bwd(da, gu)[:, :H].contiguous()
        dzt = ops.transpose2d(dz16, 64)
        dW0 = torch.empty((H, 64), device=dev, dtype=F32)
        ops.gemm_raw(dzt, xt, dW0, H, 64, 64, 64, 64, 64)
        db0 = torch.empty((8, H), device=dev, dtype=F32)
        ops.gemm_raw(ones, dzt, db0, 8, H, 64, 64, 64, H)
        return {"0.weight": dW0[:, :GEOM_FEATURE_DIM].contiguous(), "0.bias": db0[0].contiguous(), "2.weight": dW2,
                "2.bias": db2[0].contiguous()}

    def forward(self, images, geom_token, input_ids, attention_mask, labels) -> torch.Tensor:
        need_grad = torch.is_grad_enabled() and self.training
        # (eval mode without gradients - qa_inference / eval_3dqa style callers, bench.py's forward_only: the static parts replay HIP graphs)
        graphs = self.eval_graphs and not self.training and not torch.is_grad_enabled() and self.device_.type == "cuda"
        st = self.forward_state(images, geom_token, input_ids, attention_mask, labels, need_grad, graphs=graphs)
        if not need_grad:
            return st["loss"]
        self._last_state = st
        params = [p for _, p in self.text_model.named_parameters()]
        geom_in = None
        if st["geom_ctx"] is not None:
            geom_in = _GeomBridge.apply(self, st, *[self.geom_head[0].weight, self.geom_head[0].bias,
                                                    self.geom_head[2].weight, self.geom_head[2].bias])
        return _TextLossFn.apply(self, st, geom_in, *params)


class _GeomBridge(torch.autograd.Function):
    """Carries d(loss)/d(geom features) back into geom_head's fp32 parameters."""

    @staticmethod
    def forward(ctx, model, st, *params):
        ctx.model, ctx.st = model, st
        return st["geom_y"].detach().clone()

    @staticmethod
    def backward(ctx, d_y):
        g = ctx.model.geom_head_backward(ctx.st, d_y)
        return (None, None, g["0.weight"], g["0.bias"], g["2.weight"], g["2.bias"])
