"""Helpers to read tests/golden/*.npz (bf16 tensors are stored as uint16 bit patterns)."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def load(name: str):
    z = np.load(GOLDEN / name)
    return {k: z[k] for k in z.files}


def meta(z, key="meta"):
    return json.loads(bytes(z[key]).decode())


def weights(z, prefix="w:"):
    """state dict of bf16 tensors"""
    return {k[len(prefix):]: bf16(v) for k, v in z.items() if k.startswith(prefix)}
