"""MultiViewCollator on the device-side batch builder - drop-in for src/dataio/collate_multiview.py:12-102.

Same constructor, same `__call__(batch) -> {"pixel_values", "geom_token", "input_ids", "attention_mask", "labels"}`,
same layout rules: prompt "{question}\\n<image>\\n", non-string answers JSON-serialised, labels -100 on prompt and
padding, truncation to max_length, rows padded to max(longest row, num_vis_tokens + geom_tokens + 64), geometry
dicts stacked key by key with zeros for samples that have none plus a boolean "mask".

What moved to the GPU: the image transform Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor() for all B*V views in one
launch (vq3_resize_crop_u8, bit-identical to PIL/torchvision) and the id/label/mask packing (vq3_pack_tokens).
JPEG decoding and tokenisation stay on the host, as in the reference; decoded uint8 pixels go up in ONE pinned
copy and come back as the fp32 [B, V, 3, S, S] tensor already resident in HBM, where the model wants it.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib

_PLAN_CACHE: Dict[Tuple[int, int], Tuple[int, np.ndarray, np.ndarray]] = {}


def resized_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): shorter side -> size, longer = int(size * long / short); (h, w) order."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def crop_offset(full: int, size: int) -> int:
    """torchvision center_crop: int(round((full - size) / 2.0)) - Python round, i.e. half to even."""
    return int(round((full - size) / 2.0))


def axis_plan(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(ksize, bounds[out,2] int32, coefs[out,ksize] int32) for one axis, from the library's host routine."""
    key = (in_size, out_size)
    if key not in _PLAN_CACHE:
        lib = _lib.load()
        ks = lib.vq3_resample_ksize(in_size, out_size)
        if ks <= 0:
            raise _lib.Vq3Error(f"resample plan: bad sizes {in_size} -> {out_size}")
        bounds = np.empty((out_size, 2), dtype=np.int32)
        coefs = np.empty((out_size, ks), dtype=np.int32)
        _lib.check(lib.vq3_resample_plan(in_size, out_size, bounds.ctypes.data, coefs.ctypes.data), "vq3_resample_plan")
        _PLAN_CACHE[key] = (ks, bounds, coefs)
    return _PLAN_CACHE[key]


def _max_src_rows(bv: np.ndarray, crop_y: int, S: int, ty: int) -> int:
    first = bv[crop_y:crop_y + S:ty, 0]
    last_idx = np.minimum(np.arange(crop_y, crop_y + S, ty) + ty - 1, crop_y + S - 1)
    return int((bv[last_idx, 0] + bv[last_idx, 1] - first).max())


def preprocess_images(images: List[np.ndarray], size: int, device="cuda", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 RGB arrays [h, w, 3] (any sizes) -> f32 [n, 3, size, size] on `device`; one upload, one launch."""
    lib = _lib.load()
    n = len(images)
    if n == 0:
        raise ValueError("preprocess_images: empty batch")
    arrs = []
    for a in images:
        a = np.asarray(a)
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError(f"expected uint8 RGB [h, w, 3], got {a.dtype} {a.shape}")
        if min(a.shape[0], a.shape[1]) < 1:
            raise ValueError("empty image")
        arrs.append(np.ascontiguousarray(a))
    # plans (deduplicated per (in, out) pair) -> one int32 blob for coefs, one for bounds
    coef_parts, bound_parts, where = [], [], {}
    ncoef = nbound = 0

    def place(in_size, out_size):
        nonlocal ncoef, nbound
        key = (in_size, out_size)
        if key not in where:
            ks, b, c = axis_plan(in_size, out_size)
            where[key] = (ks, ncoef, nbound, b)
            coef_parts.append(c.reshape(-1))
            bound_parts.append(b.reshape(-1))
            ncoef += c.size
            nbound += b.size
        return where[key]

    descs = (_lib.ImageDesc * n)()
    offs, total = [], 0
    geo = []
    for i, a in enumerate(arrs):
        h, w = a.shape[:2]
        nh, nw = resized_size(h, w, size)
        ksh, kh_off, bh_off, _ = place(w, nw)
        ksv, kv_off, bv_off, bv = place(h, nh)
        cx, cy = crop_offset(nw, size), crop_offset(nh, size)
        geo.append((bv, cy))
        offs.append(total)
        total += (a.size + 255) // 256 * 256
        d = descs[i]
        d.h, d.w, d.pitch = h, w, 3 * w
        d.ksize_h, d.ksize_v, d.kh_off, d.kv_off, d.bh_off, d.bv_off = ksh, ksv, kh_off, kv_off, bh_off, bv_off
        d.crop_x, d.crop_y = cx, cy
    # largest tile height whose source-row span fits the 64 KiB LDS budget (256 B per source row)
    ty, rows = 16, 0
    while True:
        rows = max(_max_src_rows(bv, cy, size, ty) for bv, cy in geo)
        if rows * 256 <= 64 * 1024 or ty == 1:
            break
        ty //= 2
    if rows * 256 > 64 * 1024:
        raise _lib.Vq3Error(f"image shrink factor too large for the resize kernel ({rows} source rows per output row)")
    stage = torch.empty(total, dtype=torch.uint8, pin_memory=torch.device(device).type == "cuda")
    sn = stage.numpy()
    for a, o in zip(arrs, offs):
        sn[o:o + a.size] = a.reshape(-1)
    pix = stage.to(device, non_blocking=True)
    for i, o in enumerate(offs):
        descs[i].src = pix.data_ptr() + o
    dev_desc = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
    dev_coef = torch.from_numpy(np.concatenate(coef_parts)).to(device)
    dev_bound = torch.from_numpy(np.concatenate(bound_parts)).to(device)
    if out is None:
        out = torch.empty((n, 3, size, size), device=device, dtype=torch.float32)
    elif out.numel() != n * 3 * size * size or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous f32 tensor of n*3*size*size elements")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vq3_resize_crop_u8(dev_desc.data_ptr(), n, dev_coef.data_ptr(), dev_bound.data_ptr(), out.data_ptr(),
                                      size, ty, rows, C.c_void_p(stream)), "vq3_resize_crop_u8")
    # the staging buffers must outlive the launch on this stream
    for t in (pix, dev_desc, dev_coef, dev_bound):
        t.record_stream(torch.cuda.current_stream())
    return out


def pack_tokens(prompt_ids: List[List[int]], answer_ids: List[List[int]], max_length: int, min_length: int,
                pad_id: int, device="cuda") -> Dict[str, torch.Tensor]:
    """collate_multiview.py:56-79 on the device: returns int64 input_ids / attention_mask / labels [B, L]."""
    lib = _lib.load()
    B = len(prompt_ids)
    if B == 0 or len(answer_ids) != B:
        raise ValueError("pack_tokens: need one prompt and one answer list per row")
    L = max(max(min(len(p) + len(a), max_length) for p, a in zip(prompt_ids, answer_ids)), min_length)
    poff = np.zeros(B + 1, dtype=np.int32)
    aoff = np.zeros(B + 1, dtype=np.int32)
    poff[1:] = np.cumsum([len(p) for p in prompt_ids])
    aoff[1:] = np.cumsum([len(a) for a in answer_ids])
    host = np.concatenate([poff, aoff, np.fromiter((t for p in prompt_ids for t in p), dtype=np.int32, count=int(poff[-1])),
                           np.fromiter((t for a in answer_ids for t in a), dtype=np.int32, count=int(aoff[-1])),
                           np.zeros(1, dtype=np.int32)])
    dev = torch.from_numpy(host).to(device)
    base = dev.data_ptr()
    p_poff, p_aoff = base, base + 4 * (B + 1)
    p_prompt = base + 8 * (B + 1)
    p_answer = p_prompt + 4 * int(poff[-1])
    out = torch.empty((3, B, L), device=device, dtype=torch.int64)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vq3_pack_tokens(p_prompt, p_poff, p_answer, p_aoff, B, L, max_length, pad_id, out[0].data_ptr(),
                                   out[1].data_ptr(), out[2].data_ptr(), C.c_void_p(stream)), "vq3_pack_tokens")
    dev.record_stream(torch.cuda.current_stream())
    return {"input_ids": out[0], "labels": out[1], "attention_mask": out[2]}


class _Transform:
    """Callable with the reference's `transform(img) -> [3, S, S] float tensor` contract (build_default_transform)."""

    def __init__(self, image_size: int, device="cuda"):
        self.image_size, self.device = image_size, device

    def __call__(self, img) -> torch.Tensor:
        return preprocess_images([np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)], self.image_size,
                                 self.device)[0]


def build_default_transform(image_size: int, device="cuda") -> _Transform:
    return _Transform(image_size, device)


class MultiViewCollator:
    def __init__(self, image_size: int, tokenizer, max_length: int, num_vis_tokens: int = 128, geom_tokens: int = 8,
                 device="cuda") -> None:
        self.image_size = image_size
        self.transform = build_default_transform(image_size, device)
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.num_vis_tokens = num_vis_tokens
        self.geom_tokens = geom_tokens
        self.min_text_length = num_vis_tokens + geom_tokens + 64
        self.device = device

    def __call__(self, batch: List[Dict]) -> Dict:
        views = [len(s["images"]) for s in batch]
        if len(set(views)) != 1:
            raise RuntimeError(f"stack expects each tensor to be equal size, but samples have {views} views")
        flat = [np.asarray(img.convert("RGB") if hasattr(img, "convert") else img) for s in batch for img in s["images"]]
        S = self.image_size
        pixel = preprocess_images(flat, S, self.device).view(len(batch), views[0], 3, S, S)
        prompts, answers, geom = [], [], []
        for s in batch:
            a = s["answer"]
            if not isinstance(a, str):
                a = json.dumps(a, ensure_ascii=False)
            prompts.append(self.tokenizer(f"{s['question']}\n<image>\n", add_special_tokens=False)["input_ids"])
            answers.append(self.tokenizer(a, add_special_tokens=False)["input_ids"])
            geom.append(s.get("geom_token"))
        toks = pack_tokens(prompts, answers, self.max_length, self.min_text_length, self.tokenizer.pad_token_id, self.device)
        geom_batch = None
        if any(g is not None for g in geom):
            template = next(g for g in geom if g is not None)
            geom_batch = {}
            for key, tv in template.items():
                t0 = np.asarray(tv, dtype=np.float32)
                rows = [np.zeros_like(t0) if g is None else np.asarray(g[key], dtype=np.float32) for g in geom]
                geom_batch[key] = torch.from_numpy(np.stack(rows, axis=0)).to(self.device)
            geom_batch["mask"] = torch.tensor([g is not None for g in geom], dtype=torch.bool, device=self.device)
        return {"pixel_values": pixel, "geom_token": geom_batch, "input_ids": toks["input_ids"],
                "attention_mask": toks["attention_mask"], "labels": toks["labels"]}
