"""TEST INFRASTRUCTURE - CPU restatement of the reference's image transform (src/dataio/collate_multiview.py:12-19):
torchvision Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor() on a PIL RGB image.

The arithmetic lives in third-party code that is not under /root/reference: Pillow (`Image.resize` ->
ImagingResample, two passes with uint8 intermediate and 22-bit fixed-point taps; the reference pins no version,
env/environment.yml lists `pillow`; this container has Pillow 12.2.0) and torchvision (size rule and crop offsets;
absent here, rule restated from torchvision.transforms.functional.resize/center_crop). Pinned by running Pillow
itself on the same inputs (tests/test_oracle_golden.py) and by tests/golden/preprocess_tiny.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)          # C (int) cast truncates toward zero
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """One Pillow pass along axis 0 of a uint8 array [n, ...]."""
    _, bounds, kk = precompute_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for i in range(out_size):
        lo, n = bounds[i]
        k = kk[i, :n].reshape((n,) + (1,) * (img.ndim - 1))
        out[i] = _clip8((1 << (PRECISION_BITS - 1)) + (src[lo:lo + n] * k).sum(axis=0))
    return out


def pil_resize_bicubic(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Image.resize((new_w, new_h), BICUBIC) on uint8 [h, w, 3]: horizontal pass, then vertical (each skipped when that
    size is unchanged; Pillow additionally restricts the horizontal pass to the rows the vertical one reads, which
    does not change values)."""
    h, w = img.shape[:2]
    if w != new_w:
        img = np.swapaxes(resample_axis0(np.swapaxes(img, 0, 1), new_w), 0, 1)
    if h != new_h:
        img = resample_axis0(img, new_h)
    return np.ascontiguousarray(img)


def resized_size(h: int, w: int, size: int):
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def transform(img: np.ndarray, size: int) -> np.ndarray:
    """uint8 RGB [h, w, 3] -> float32 [3, size, size] in [0, 1]."""
    nh, nw = resized_size(img.shape[0], img.shape[1], size)
    r = pil_resize_bicubic(img, nh, nw)
    top = int(round((nh - size) / 2.0))
    left = int(round((nw - size) / 2.0))
    c = r[top:top + size, left:left + size]
    return (np.transpose(c, (2, 0, 1)).astype(np.float32) / np.float32(255.0)).astype(np.float32)
