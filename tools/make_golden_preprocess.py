"""Writes tests/golden/preprocess_tiny.npz: seeded uint8 images and what Pillow (the reference's image backend,
src/dataio/collate_multiview.py:12-19 via torchvision -> PIL.Image.resize) produces for them. torchvision is not
installed here, so its two integer rules (resize target, centre-crop offsets) are applied by hand around the real
PIL.Image.resize call; ToTensor() is `uint8 / 255` in float32. Run in the build container only."""
import json
from pathlib import Path

import numpy as np
from PIL import Image
import PIL

OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "preprocess_tiny.npz"
CASES = [(37, 53, 16), (100, 80, 32), (64, 64, 64), (30, 200, 24), (97, 33, 40), (20, 30, 28), (241, 163, 56), (57, 56, 56)]


def main():
    rng = np.random.default_rng(20251205)
    arrays = {}
    for i, (h, w, S) in enumerate(CASES):
        # smooth field + noise + hard edges, so both ringing (clipping) and fine detail are exercised
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 120 * np.sin(xx / 7.0 + i)[..., None] * np.cos(yy / 5.0)[..., None] * np.array([1, -1, 0.5])
        img = np.clip(base + rng.normal(0, 30, (h, w, 3)), 0, 255)
        img[h // 3: h // 2, w // 4: w // 2] = rng.choice([0, 255])
        img = img.astype(np.uint8)
        short, long = (w, h) if w <= h else (h, w)
        if short == S:
            nh, nw = h, w
        else:
            ns, nl = S, int(S * long / short)
            nh, nw = (nl, ns) if w <= h else (ns, nl)
        resized = np.asarray(Image.fromarray(img, "RGB").resize((nw, nh), Image.BICUBIC))
        top, left = int(round((nh - S) / 2.0)), int(round((nw - S) / 2.0))
        crop = resized[top:top + S, left:left + S]
        out = np.transpose(crop, (2, 0, 1)).astype(np.float32) / np.float32(255)
        arrays[f"in{i}"] = img
        arrays[f"resized{i}"] = resized
        arrays[f"out{i}"] = out.astype(np.float32)
    arrays["meta"] = np.frombuffer(json.dumps({"cases": CASES, "pillow": PIL.__version__}).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
