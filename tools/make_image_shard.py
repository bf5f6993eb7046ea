#!/usr/bin/env python3
"""Decode the radiographs of <data_path>/mimic-cxr-2.0.0-entity-llm.csv ONCE into a uint8 shard for `main_pretrain.py --image_shard`
(module/pretrain_datasets.py: U8ShardWriter).

    python tools/make_image_shard.py --data_path <dir> --out <dir>/images.u8 [--max_side 1024] [--workers 16]

Without --max_side the shard holds the JPEGs' own pixels (7.8 MB per 2544 x 3056 radiograph: the crops are then exactly the reference's);
--max_side shrinks the longer side first (Pillow bicubic) -- smaller files and host -> HBM copies, crops from the shrunk image."""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def decode(args):
    path, max_side = args
    from PIL import Image
    img = Image.open(path).convert("L")
    if max_side and max(img.size) > max_side:
        r = max_side / float(max(img.size))
        img = img.resize((max(1, int(round(img.size[0] * r))), max(1, int(round(img.size[1] * r)))), Image.BICUBIC)
    return np.asarray(img, dtype=np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_path", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--max_side", type=int, default=0)
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    import pandas as pd
    from ecamp_amd.module.pretrain_datasets import U8ShardWriter
    paths = list(pd.read_csv(os.path.join(a.data_path, "mimic-cxr-2.0.0-entity-llm.csv"))["img_path"])
    with U8ShardWriter(a.out) as w, ProcessPoolExecutor(a.workers) as ex:
        for n, arr in enumerate(ex.map(decode, [(p, a.max_side) for p in paths], chunksize=8)):
            w.add(arr)
            if n % 1000 == 0:
                print("%d / %d" % (n, len(paths)), flush=True)
    print("wrote %s (%.1f GB) + .idx.npy" % (a.out, os.path.getsize(a.out) / 1e9))


if __name__ == "__main__":
    main()
