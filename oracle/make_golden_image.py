#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the image half of SURVEY.md 8(f) f2 (pretrain_datasets.py:47-52): crop boxes and flips
drawn by oracle/tv_transforms.py (torchvision 0.14.1's get_params restated; torchvision itself is not installed anywhere this repo
runs) under fixed torch seeds for image shapes that reach every branch (a hit on the first try, several misses, the three
fallbacks), and a checksum + sample of the final uint8 item on a synthetic radiograph.  Writes tests/golden/image_transform.npz.

    python oracle/make_golden_image.py
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tv_transforms as tv  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "image_transform.npz")
SHAPES = [(2544, 3056), (3056, 2544), (1024, 1024), (500, 448), (448, 448), (300, 200), (100, 2000), (2000, 100), (37, 41)]   # (W, H)


def synthetic_radiograph(W, H, seed):
    """Smooth structure + noise, uint8 [H, W] (a stand-in: MIMIC-CXR is licensed)."""
    g = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    v = 120 + 80 * np.sin(x / (W / 7.0)) * np.cos(y / (H / 5.0)) + 30 * np.sin((x + y) / 23.0) + g.normal(0, 12, (H, W))
    return np.clip(v, 0, 255).astype(np.uint8)


def main():
    rec = {"shapes": np.asarray(SHAPES, dtype=np.int64)}
    params = []
    for n, (W, H) in enumerate(SHAPES):
        for seed in range(6):
            torch.manual_seed(1000 * n + seed)
            i, j, h, w = tv.tv_get_params(W, H)
            flip = tv.tv_flip()
            params.append((n, 1000 * n + seed, i, j, h, w, int(flip)))
    rec["params"] = np.asarray(params, dtype=np.int64)          # rows: shape index, torch seed, i, j, h, w, flip
    items = []
    for n, (W, H) in enumerate([(900, 1100), (300, 200), (100, 2000)]):
        img = synthetic_radiograph(W, H, seed=n)
        torch.manual_seed(77 + n)
        out, p = tv.reference_item_u8(img, 448)
        items.append((W, H, n, 77 + n) + p + (zlib.crc32(out.tobytes()),))
        rec["item%d_sample" % n] = out[::37, ::41].copy()
        t = tv.to_tensor_normalize(out)
        rec["item%d_f32_sample" % n] = t[:, ::37, ::41].numpy().copy()
    rec["items"] = np.asarray(items, dtype=np.int64)           # W, H, image seed, torch seed, i, j, h, w, flip, crc32 of the uint8 item
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(params), "draws,", len(items), "items")


if __name__ == "__main__":
    main()
