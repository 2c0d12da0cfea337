"""VGGT aggregator (multi-view ViT) on the HIP kernels.

The reference does not vendor this model: it imports `vggt.models.vggt.VGGT` from an external package
(third_party/README.md; src/models/vggt_qwen3_vlm.py:72-83) and only ever calls `.aggregator(images)`
(:144) -> `(list_of_tokens, patch_start_idx)`, reading `list[-1]` of shape [B, S, P, 2*embed_dim] (:148).
This module rebuilds that call surface from the published architecture (facebookresearch/vggt:
models/aggregator.py, layers/{block,attention,rope,vision_transformer}.py):

  ImageNet-normalise -> DINOv2 ViT-L/14 (+4 register tokens, 24 blocks, final LayerNorm) patch tokens ->
  [camera | 4 register | patches] per frame -> 24 x { frame-attention block over P tokens ;
  global-attention block over S*P tokens }, each block = pre-LN, qkv, per-head LayerNorm on q/k, 2-D RoPE
  (frequency 100), SDPA, proj, LayerScale, GELU MLP, LayerScale -> concat(frame_i, global_i) on the feature dim.

Parameter names follow the upstream state dict (`aggregator.patch_embed.blocks.N.attn.qkv.weight`,
`aggregator.frame_blocks.N.attn.q_norm.weight`, ...) so `vggt_1B_commercial.pt` can be loaded by name when it is
available. PARITY UNPINNED: no source/weights/test vector of the package exists in the reference repository; the HIP
path is checked against this repo's own CPU restatement (oracle/vggt.py) only. Forward only (frozen, no_grad).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ops import BF16, F32, round_up


def _block_param_shapes(C: int, mlp: int, qk_norm: bool) -> Dict[str, tuple]:
    hd = 64
    s = {"norm1.weight": (C,), "norm1.bias": (C,), "attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,),
         "attn.proj.weight": (C, C), "attn.proj.bias": (C,), "ls1.gamma": (C,), "norm2.weight": (C,),
         "norm2.bias": (C,), "mlp.fc1.weight": (mlp, C), "mlp.fc1.bias": (mlp,), "mlp.fc2.weight": (C, mlp),
         "mlp.fc2.bias": (C,), "ls2.gamma": (C,)}
    if qk_norm:
        s.update({"attn.q_norm.weight": (hd,), "attn.q_norm.bias": (hd,), "attn.k_norm.weight": (hd,),
                  "attn.k_norm.bias": (hd,)})
    return s


class Aggregator(nn.Module):
    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 num_register_tokens=4, rope_freq=100.0, init_values=0.01, dino_depth: Optional[int] = None,
                 device="cuda", seed: int = 0):
        super().__init__()
        if embed_dim % 64 or embed_dim // num_heads != 64:
            raise ops._lib.Vq3Error("VGGT HIP path requires head_dim == 64")
        self.img_size, self.patch_size, self.embed_dim = img_size, patch_size, embed_dim
        # the blocks' two pre-LayerNorms folded into the q|k|v and fc1 GEMMs (needs 128-column statistic groups; VQ3_VGGT_LN_FOLD=0: off)
        self._ln_fold = embed_dim % 128 == 0 and os.environ.get("VQ3_VGGT_LN_FOLD", "1") != "0"
        self.depth, self.num_heads = depth, num_heads
        self.dino_depth = depth if dino_depth is None else dino_depth
        self.mlp_dim = int(embed_dim * mlp_ratio)
        self.num_register_tokens = num_register_tokens
        self.patch_start_idx = 1 + num_register_tokens
        self.rope_freq = rope_freq
        self.kp = round_up(3 * patch_size * patch_size, 64)
        C = embed_dim
        g = torch.Generator(device="cpu").manual_seed(seed)
        shapes: Dict[str, tuple] = {}
        M = img_size // patch_size
        shapes["patch_embed.cls_token"] = (1, 1, C)
        shapes["patch_embed.pos_embed"] = (1, 1 + M * M, C)
        shapes["patch_embed.register_tokens"] = (1, num_register_tokens, C)
        shapes["patch_embed.patch_embed.proj.weight"] = (C, 3, patch_size, patch_size)
        shapes["patch_embed.patch_embed.proj.bias"] = (C,)
        for i in range(self.dino_depth):
            for k, s in _block_param_shapes(C, self.mlp_dim, False).items():
                shapes[f"patch_embed.blocks.{i}.{k}"] = s
        shapes["patch_embed.norm.weight"] = (C,)
        shapes["patch_embed.norm.bias"] = (C,)
        for kind in ("frame_blocks", "global_blocks"):
            for i in range(depth):
                for k, s in _block_param_shapes(C, self.mlp_dim, True).items():
                    shapes[f"{kind}.{i}.{k}"] = s
        shapes["camera_token"] = (1, 2, 1, C)
        shapes["register_token"] = (1, 2, num_register_tokens, C)
        self._names: List[str] = []
        for name, shp in shapes.items():
            t = torch.empty(shp, dtype=F32)
            leaf = name.split(".")[-1]
            if name in ("camera_token", "register_token", "patch_embed.cls_token", "patch_embed.register_tokens"):
                t.normal_(0, 1e-6, generator=g)
            elif name.endswith("gamma"):
                t.fill_(1.0 if name.startswith("patch_embed") else init_values)
            elif leaf == "bias":
                t.zero_()
            elif len(shp) == 1:
                t.fill_(1.0)
            else:
                t.normal_(0, 0.02, generator=g)
            pname = name.replace(".", "__")
            self.register_parameter(pname, nn.Parameter(t.to(BF16).to(device), requires_grad=False))
            self._names.append(name)
        self._cc = None
        self._pos_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        self._rope_cache: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        # Checkpoint key space: a parameter is REGISTERED as `patch_embed__blocks__0__attn__qkv__weight`
        # (nn.Module forbids dots in names) but state_dict() / load_state_dict() speak the upstream dotted names, so a
        # reference-written `vision_model.aggregator.patch_embed.blocks.0.attn.qkv.weight` round-trips by name.
        self._register_state_dict_hook(Aggregator._dotted_keys_hook)
        self._register_load_state_dict_pre_hook(self._internal_keys_pre_hook)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_compute_copies())

    @staticmethod
    def _dotted_keys_hook(module, state_dict, prefix, local_metadata):
        for name in module._names:
            k = prefix + name.replace(".", "__")
            if k in state_dict:
                state_dict[prefix + name] = state_dict.pop(k)
        return state_dict

    def _internal_keys_pre_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in self._names:
            k = prefix + name
            if k in state_dict and "." in name:
                state_dict[prefix + name.replace(".", "__")] = state_dict.pop(k)

    # state-dict keys use the upstream dotted names
    def p(self, name: str) -> torch.Tensor:
        return getattr(self, name.replace(".", "__"))

    def named_tensors(self) -> Dict[str, torch.Tensor]:
        return {n: self.p(n) for n in self._names}

    def load_named(self, sd: Dict[str, torch.Tensor]) -> List[str]:
        missing = []
        with torch.no_grad():
            for n in self._names:
                if n in sd:
                    self.p(n).copy_(sd[n].to(self.p(n).dtype))
                else:
                    missing.append(n)
        self._cc = None
        self._pos_cache.clear()
        return missing

    def invalidate_compute_copies(self) -> None:
        self._cc = None
        self._pos_cache.clear()

    # ------------------------------------------------------------------ compute copies (fp32 vectors)
    def _prepare(self):
        if self._cc is not None:
            return self._cc
        f = lambda n: self.p(n).detach().float().contiguous()
        cc = {"blocks": {}}
        def blk(prefix, qk):
            d = {"n1": (f(prefix + "norm1.weight"), f(prefix + "norm1.bias")),
                 "n2": (f(prefix + "norm2.weight"), f(prefix + "norm2.bias")),
                 "qkv_w": self.p(prefix + "attn.qkv.weight"), "qkv_b": f(prefix + "attn.qkv.bias"),
                 "proj_w": self.p(prefix + "attn.proj.weight"), "proj_b": f(prefix + "attn.proj.bias"),
                 "fc1_w": self.p(prefix + "mlp.fc1.weight"), "fc1_b": f(prefix + "mlp.fc1.bias"),
                 "fc2_w": self.p(prefix + "mlp.fc2.weight"), "fc2_b": f(prefix + "mlp.fc2.bias"),
                 "ls1": f(prefix + "ls1.gamma"), "ls2": f(prefix + "ls2.gamma")}
            if qk:
                d["qn"] = (f(prefix + "attn.q_norm.weight"), f(prefix + "attn.q_norm.bias"))
                d["kn"] = (f(prefix + "attn.k_norm.weight"), f(prefix + "attn.k_norm.bias"))
                # A bound on |q . k| / sqrt(hd) from the WEIGHTS alone: LayerNorm over the hd head channels leaves sum(xhat^2) <= hd, so
                # |gamma o xhat + beta| <= sqrt(hd) max|gamma| + |beta|, RoPE rotates pairs (norm-preserving); 5 % for the bf16 roundings
                # of q, k and the scaled q. A small bound lets the attention run without a running maximum (vq3_flash_attn_fwd_bounded).
                hd = d["qn"][0].numel()
                lim = lambda wb: float(hd ** 0.5 * wb[0].abs().max() + wb[1].norm())
                d["score_bound"] = 1.05 * lim(d["qn"]) * lim(d["kn"]) * hd ** -0.5
            if self._ln_fold:
                # Linear(LayerNorm(x)) = rstd (x . (gamma o W)^T - mu colsum) + (b + W . beta): the frozen tower's two pre-norms are
                # folded into the weights once (ops.ln_fold); colsum is taken from the bf16-rounded product the GEMM will read
                for lin, nrm in (("qkv", "n1"), ("fc1", "n2")):
                    name = prefix + ("attn.qkv." if lin == "qkv" else "mlp.fc1.")
                    W = self.p(name + "weight").detach().float()
                    wf = (W * d[nrm][0][None, :]).to(BF16).contiguous()
                    d[lin + "_wf"] = wf
                    d[lin + "_c"] = wf.float().sum(1).contiguous()
                    d[lin + "_d"] = (f(name + "bias") + W @ d[nrm][1]).contiguous()
            return d
        cc["dino"] = [blk(f"patch_embed.blocks.{i}.", False) for i in range(self.dino_depth)]
        cc["frame"] = [blk(f"frame_blocks.{i}.", True) for i in range(self.depth)]
        cc["global"] = [blk(f"global_blocks.{i}.", True) for i in range(self.depth)]
        C = self.embed_dim
        w = self.p("patch_embed.patch_embed.proj.weight").detach().reshape(C, -1)
        wpe = torch.zeros((C, self.kp), device=w.device, dtype=BF16)
        wpe[:, : w.shape[1]] = w
        cc["pe_w"], cc["pe_b"] = wpe, f("patch_embed.patch_embed.proj.bias")
        cc["dino_norm"] = (f("patch_embed.norm.weight"), f("patch_embed.norm.bias"))
        self._cc = cc
        return cc

    def _pos_embed(self, Hp: int, Wp: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """DINOv2 interpolate_pos_encoding (bicubic, antialias, fp32 then cast): a parameter-only precompute,
        cached per grid size. Returns (special rows [5, C] = [cls+pos0 | registers], patch pos [Hp*Wp, C]) bf16."""
        key = (Hp, Wp)
        if key not in self._pos_cache:
            pe = self.p("patch_embed.pos_embed").detach().float().cpu()
            C = pe.shape[-1]
            Mg = int(math.sqrt(pe.shape[1] - 1))
            cls_pos, patch_pos = pe[:, 0], pe[:, 1:]
            if Mg * Mg == Hp * Wp and Hp == Wp:
                pp = patch_pos
            else:
                pp = F.interpolate(patch_pos.reshape(1, Mg, Mg, C).permute(0, 3, 1, 2), size=(Hp, Wp), mode="bicubic",
                                   antialias=True).permute(0, 2, 3, 1).reshape(1, -1, C)
            dev = self.p("camera_token").device
            cls = self.p("patch_embed.cls_token").detach()[0, 0]
            cls_row = (cls + cls_pos[0].to(BF16).to(dev)).to(BF16)  # bf16 + bf16 -> bf16
            special = torch.cat([cls_row[None], self.p("patch_embed.register_tokens").detach()[0]], dim=0)
            self._pos_cache[key] = (special.contiguous(), pp[0].to(BF16).to(dev).contiguous())
        return self._pos_cache[key]

    def _rope_tables(self, maxpos: int):
        if maxpos not in self._rope_cache:
            fd = 32
            inv = 1.0 / (self.rope_freq ** (torch.arange(0, fd, 2).float() / fd))
            ang = torch.einsum("i,j->ij", torch.arange(maxpos + 1).float(), inv).to(BF16)
            ang = torch.cat((ang, ang), dim=-1)
            dev = self.p("camera_token").device
            self._rope_cache[maxpos] = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
        return self._rope_cache[maxpos]

    # ------------------------------------------------------------------ row split of the row-wise chain of a block
    ROW_UNIT = 256 * 64          # rows that make whole rounds of 256 CUs for 4, 12 and 16 column tiles of 256 (N = C, 3 C, 4 C at C = 1024)

    def _split_rows(self, T: int) -> int:
        """Rows of the main chain (0: no split): whole multiples of ROW_UNIT, when 1 .. 2048 rows are left behind them."""
        mode = os.environ.get("VQ3_VGGT_ROW_SPLIT", "1")        # 0: never; 2: at any width (tests); default: at the width ROW_UNIT is for
        if mode == "0" or (self.embed_dim != 1024 and mode != "2"):
            return 0
        main = T // self.ROW_UNIT * self.ROW_UNIT
        return main if (main > 0 and 0 < T - main <= 2048) else 0

    def _tail_stream_for(self, dev) -> "torch.cuda.Stream":
        ts = getattr(self, "_tail_streams", None)
        if ts is None:
            ts = self._tail_streams = {}
        key = torch.device(dev).index
        if key not in ts:
            ts[key] = torch.cuda.Stream(device=dev)
        return ts[key]

    # ------------------------------------------------------------------ one transformer block
    def _block(self, x, w, N, *, rope, eps, P, Wp, st=None, keep=None):
        """One pre-norm block. Returns (x_out, statistics of x_out's rows for the next block's first LayerNorm, or None).
        keep = n: only the first n rows of every N-token group leave the block ([G*n, C]) - q|k|v still covers all rows (every key and
        value is attended), attention / proj / MLP run on the kept queries alone. Row-wise ops: the kept rows equal the full block's."""
        NH = self.num_heads
        if keep is not None and keep < N:
            G = x.shape[0] // N
            head = lambda t: t.view(G, N, *t.shape[1:])[:, :keep].reshape(G * keep, *t.shape[1:]).contiguous()
        else:
            keep, head = None, (lambda t: t)
        if self._ln_fold:
            # no LayerNorm launch, no normalised copy of x: the q|k|v and fc1 GEMMs read the raw rows and apply (mu, rstd) in their
            # epilogues; the residual GEMMs that form x leave the (sum, sum of squares) pairs the next fold needs
            T, C = x.shape
            if st is None:
                st = ops.rowstats128(x)
            kw = dict(ln_fold=ops.ln_fold(stats_in=st, eps=eps, colsum=w["qkv_c"]))
            if rope is not None:
                Q, K, V = ops.linear_vit_qkv(x, w["qkv_wf"], w["qkv_d"], N, NH, qn=w["qn"], kn=w["kn"], cos=rope[0], sin=rope[1],
                                             tokens_per_frame=P, patch_start=self.patch_start_idx, Wp=Wp, eps=1e-5, **kw)
            else:
                Q, K, V = ops.linear_vit_qkv(x, w["qkv_wf"], w["qkv_d"], N, NH, **kw)
            o = ops.flash_attn(Q, K, V, q_rows=keep, score_bound=w.get("score_bound") if rope is not None else None)
            x = head(x)
            st2 = torch.empty((x.shape[0], C // 128, 2), device=x.device, dtype=torch.float32)
            st3 = torch.empty_like(st2)
            Mm = self._split_rows(x.shape[0]) if keep is None else 0
            if Mm:
                # proj -> fc1 -> fc2 are row-wise: the rows that fill whole rounds of the 256 x 256 kernel go down the caller's stream,
                # the <= 2048 rows behind them as a chain of their own on a second stream - forked here (the attention output is
                # complete), joined before the next block's q|k|v, which reads every row. Inside one GEMM call the row tail is a
                # 20-40 us launch that 64 CUs run ALONE behind the main launch (cfg 30: 36 of them per micro-batch, 0.95 ms); as a chain
                # beside the main one its workgroups are dispatched whenever a CU has nothing else.
                T = x.shape[0]
                x1, x2 = torch.empty_like(x), torch.empty_like(x)
                h = torch.empty((T, w["fc1_wf"].shape[0]), device=x.device, dtype=BF16)
                cur = torch.cuda.current_stream()
                side = self._tail_stream_for(x.device)

                def chain(sl):
                    ops.linear(o[sl], w["proj_w"], bias=w["proj_b"], colscale=w["ls1"], residual=x[sl], out=x1[sl],
                               ln_fold=ops.ln_fold(stats_out=st2[sl]))
                    ops.linear(x1[sl], w["fc1_wf"], bias=w["fc1_d"], act=ops.ACT_GELU, out=h[sl],
                               ln_fold=ops.ln_fold(stats_in=st2[sl], eps=eps, colsum=w["fc1_c"]))
                    ops.linear(h[sl], w["fc2_w"], bias=w["fc2_b"], colscale=w["ls2"], residual=x1[sl], out=x2[sl],
                               ln_fold=ops.ln_fold(stats_out=st3[sl]))
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    chain(slice(Mm, T))
                chain(slice(0, Mm))
                cur.wait_stream(side)
                return x2, st3
            x = ops.linear(o, w["proj_w"], bias=w["proj_b"], colscale=w["ls1"], residual=x, ln_fold=ops.ln_fold(stats_out=st2))
            h = ops.linear(x, w["fc1_wf"], bias=w["fc1_d"], act=ops.ACT_GELU,
                           ln_fold=ops.ln_fold(stats_in=st2, eps=eps, colsum=w["fc1_c"]))
            x = ops.linear(h, w["fc2_w"], bias=w["fc2_b"], colscale=w["ls2"], residual=x, ln_fold=ops.ln_fold(stats_out=st3))
            return x, st3
        xn, _ = ops.layernorm_fwd(x, w["n1"][0], w["n1"][1], eps)
        # qkv projection with the head split, q/k LayerNorm and 2-D RoPE fused into the GEMM epilogue (one launch, no [T, 3C] tensor)
        if rope is not None:
            Q, K, V = ops.linear_vit_qkv(xn, w["qkv_w"], w["qkv_b"], N, NH, qn=w["qn"], kn=w["kn"], cos=rope[0], sin=rope[1],
                                         tokens_per_frame=P, patch_start=self.patch_start_idx, Wp=Wp, eps=1e-5)
        else:
            Q, K, V = ops.linear_vit_qkv(xn, w["qkv_w"], w["qkv_b"], N, NH)
        o = ops.flash_attn(Q, K, V, q_rows=keep, score_bound=w.get("score_bound") if rope is not None else None)
        x = ops.linear(o, w["proj_w"], bias=w["proj_b"], colscale=w["ls1"], residual=head(x))
        xn2, _ = ops.layernorm_fwd(x, w["n2"][0], w["n2"][1], eps)
        h = ops.linear(xn2, w["fc1_w"], bias=w["fc1_b"], act=ops.ACT_GELU)
        return ops.linear(h, w["fc2_w"], bias=w["fc2_b"], colscale=w["ls2"], residual=x), None

    def _dino(self, cc, images: torch.Tensor):
        """ImageNet normalisation + DINOv2-with-registers backbone (`patch_embed`): images [B, S, 3, H, W] in [0,1] ->
        tokens after the final LayerNorm [B*S*P, C], rows ordered [cls | 4 registers | patches] per frame."""
        B, S, Cin, H, W = images.shape
        p, C = self.patch_size, self.embed_dim
        if Cin != 3 or H % p or W % p:
            raise ValueError(f"Expected 3 input channels and sizes divisible by {p}, got {tuple(images.shape)}")
        Hp, Wp = H // p, W // p
        Np = Hp * Wp
        P = self.patch_start_idx + Np
        BS = B * S
        dev = images.device
        img = images.reshape(BS, 3, H, W).to(F32).contiguous()
        patches = ops.im2col_norm(img, p, self.kp)
        special, pos_patch = self._pos_embed(Hp, Wp)
        tok = torch.empty((BS, P, C), device=dev, dtype=BF16)
        ops.gemm_raw(patches, cc["pe_w"], tok, Np, C, self.kp, self.kp, self.kp, C, bias=cc["pe_b"], R=pos_patch,
                     ldr=C, nb1=BS, sA=(Np * self.kp, 0), sC=(P * C, 0), c_off=self.patch_start_idx * C)
        tok[:, : self.patch_start_idx] = special  # row copy (data movement)
        x = tok.view(BS * P, C)
        st = None
        for w in cc["dino"]:
            x, st = self._block(x, w, P, rope=None, eps=1e-6, P=P, Wp=Wp, st=st)
        x, _ = ops.layernorm_fwd(x, cc["dino_norm"][0], cc["dino_norm"][1], 1e-6)
        return x

    @torch.no_grad()
    def dino_tokens(self, images: torch.Tensor) -> torch.Tensor:
        """The backbone stage alone: [B, S, 3, H, W] -> [B*S, 1 + 4 + Hp*Wp, C] (what transformers'
        Dinov2WithRegistersModel calls last_hidden_state; pinned by tests/golden/dinov2_tiny.npz)."""
        x = self._dino(self._prepare(), images)
        B, S = images.shape[:2]
        return x.view(B * S, -1, self.embed_dim)

    @torch.no_grad()
    def forward(self, images: torch.Tensor, return_all: bool = False, _head_rows: Optional[int] = None):
        """images [B, S, 3, H, W] in [0,1] -> ([... , tokens [B, S, P, 2C]], patch_start_idx).
        Only the last iterate is materialised unless return_all (the reference reads list[-1] only)."""
        cc = self._prepare()
        B, S, Cin, H, W = images.shape
        p, C = self.patch_size, self.embed_dim
        x = self._dino(cc, images)
        Hp, Wp = H // p, W // p
        P = self.patch_start_idx + Hp * Wp
        dev = images.device
        # [camera | register | patch tokens]: frame 0 takes slot 0 of the learned tokens, the other frames slot 1
        cam, reg = self.p("camera_token").detach()[0], self.p("register_token").detach()[0]   # [2,1,C], [2,4,C]
        sp = torch.cat([cam, reg], dim=1)                                                     # [2, 5, C]
        sel = (torch.arange(S, device=dev) != 0).long()         # frame 0 -> slot 0, the others -> slot 1 (no host scalar: capturable)
        xv = x.view(B, S, P, C)
        xv[:, :, : self.patch_start_idx] = sp[sel][None]
        rope = self._rope_tables(max(Hp, Wp))
        outs = []
        st = None                                          # (the special-token rows were just rewritten: fresh statistics)
        for i in range(self.depth):
            x, st = self._block(x, cc["frame"][i], P, rope=rope, eps=1e-5, P=P, Wp=Wp, st=st)
            fr = x
            if _head_rows is not None and not return_all and i == self.depth - 1:
                n = min(_head_rows, S * P)
                x, _ = self._block(x, cc["global"][i], S * P, rope=rope, eps=1e-5, P=P, Wp=Wp, st=st, keep=n)
                outs.append(torch.cat([fr.view(B, S * P, C)[:, :n], x.view(B, n, C)], dim=-1))          # [B, n, 2C]
                break
            x, st = self._block(x, cc["global"][i], S * P, rope=rope, eps=1e-5, P=P, Wp=Wp, st=st)
            if return_all or i == self.depth - 1:
                outs.append(torch.cat([fr.view(B, S, P, C), x.view(B, S, P, C)], dim=-1))
        return outs, self.patch_start_idx

    @torch.no_grad()
    def forward_head(self, images: torch.Tensor, n: int) -> torch.Tensor:
        """forward(images)[0][-1].reshape(B, S*P, 2C)[:, :n] - all the reference reads of the tower (vggt_qwen3_vlm.py:144-156: the last
        iterate, flattened over views, first num_vis_tokens rows) - without the rows nobody reads: the LAST global block computes its
        attention output, proj and MLP for the first n tokens of every sample only (its q|k|v, and every earlier block, are needed in
        full: all tokens are keys / values of the kept queries)."""
        toks, _ = self.forward(images, _head_rows=int(n))
        return toks[-1]


class VGGT(nn.Module):
    """Call surface the reference uses: VGGT(img_size=518, patch_size=14, embed_dim=1024, enable_*=...) with an
    `.aggregator` sub-module. The camera/point/depth/track heads are never run by the reference
    (vggt_qwen3_vlm.py:142-144) and are not built."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, enable_camera=True, enable_point=True,
                 enable_depth=True, enable_track=True, depth=24, num_heads=None, dino_depth=None, device="cuda",
                 seed: int = 0, **_unused):
        super().__init__()
        self.aggregator = Aggregator(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                                     num_heads=num_heads or embed_dim // 64, dino_depth=dino_depth, device=device,
                                     seed=seed)
        self.embed_dim = 2 * embed_dim

    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor]) -> List[str]:
        """Loads `aggregator.*` entries of an upstream VGGT checkpoint by name (heads are ignored)."""
        agg = {k[len("aggregator."):]: v for k, v in sd.items() if k.startswith("aggregator.")}
        return self.aggregator.load_named(agg)
