#!/usr/bin/env python3
"""Random radiograph shapes (16 ... 3200 px a side, any aspect ratio) through the device image pipeline against the host transform on the same
stored pixels and the same torch seed: byte for byte (module/pretrain_datasets.py: device_crop_item -> pack_crops -> DeviceAugmenter versus
default_image_transform = the reference's RandomResizedCrop(448, bicubic) / flip / Grayscale).   python tools/fuzz_augment.py [--cases 96]"""
import argparse, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from ecamp_amd.module import pretrain_datasets as pd
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=96); ap.add_argument("--seed", type=int, default=0); args = ap.parse_args()
dev = torch.device("cuda:0")
rng = random.Random(args.seed)
tf = pd.default_image_transform(448, image_u8=True)
aug = pd.DeviceAugmenter(dev)
bad = 0
for c0 in range(0, args.cases, 16):
    imgs, want, items = [], [], []
    for c in range(c0, min(args.cases, c0 + 16)):
        W = rng.choice([rng.randint(16, 200), rng.randint(200, 1200), rng.randint(1200, 3200)]); H = rng.choice([rng.randint(16, 200), rng.randint(200, 1200), rng.randint(1200, 3200)])
        g = np.random.default_rng(1000 + c)
        im = (g.random((H, W)) * 255).astype(np.uint8) if c % 3 else np.add.outer(np.arange(H) * 7 % 256, np.arange(W) * 3 % 256).astype(np.uint8)
        torch.manual_seed(5000 + c); want.append(tf(Image.fromarray(im, "L").convert("RGB")))
        torch.manual_seed(5000 + c); items.append(pd.device_crop_item(im))
        imgs.append((W, H))
    flat, table = pd.pack_crops(items, pin=True)
    got = aug(flat, table, check=True).cpu()
    for n, (W, H) in enumerate(imgs):
        if not torch.equal(got[n], want[n]):
            bad += 1; print("FAIL image %d x %d: %d bytes differ (max |d| %d)" % (W, H, int((got[n] != want[n]).sum()), int((got[n].int() - want[n].int()).abs().max())), flush=True)
print("fuzz_augment: %d images, %d differ from the host item" % (args.cases, bad), flush=True)
sys.exit(1 if bad else 0)
