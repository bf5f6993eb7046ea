#!/usr/bin/env python3
"""What ONE loader worker delivers on MIMIC-CXR-sized inputs, host path against device path.

    python tools/loader_rate.py [--n 24] [--workers 16] [--step-rate 7100]

Synthetic stand-ins (MIMIC-CXR-JPG is licensed): grayscale JPEGs of 2544 x 3056 pixels (the dataset's usual size), quality 95, with
smooth structure + noise so that the entropy coder has real work.  Timed per item, one process, one thread:
  host  = the reference's __getitem__ image half (pretrain_datasets.py:28-31,47-52,113-115): open + JPEG decode + convert('RGB'),
          RandomResizedCrop(448, bicubic) + flip + Grayscale + ToTensor + Normalize        (ecamp_amd default_image_transform)
  u8    = the same up to the uint8 crop (image_u8 schema)
  dev   = the device path's worker share: memory-mapped uint8 shard -> draw box / flip -> copy the box's bytes (device_crop_item);
          the resample runs on the GPU (csrc/augment.hip; timed by tests/test_augment.py::test_device_augmenter_at_batch_size)
          for shards stored at full size and with max_side = 1024
and turned into the pairs/s a loader of `--workers` processes can feed one GPU, next to the GPU's step rate.
"""
import argparse
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def radiograph(W, H, seed):
    g = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    v = 120 + 80 * np.sin(x / (W / 7.0)) * np.cos(y / (H / 5.0)) + 30 * np.sin((x + y) / 23.0) + g.normal(0, 6, (H, W)).astype(np.float32)
    return np.clip(v, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--workers", type=int, default=16, help="DataLoader workers per GPU (main_pretrain.py:222 uses 16)")
    ap.add_argument("--step-rate", type=float, default=7100.0, help="pairs/s one MI355X trains at (bench.py)")
    ap.add_argument("--width", type=int, default=2544)
    ap.add_argument("--height", type=int, default=3056)
    args = ap.parse_args()
    from PIL import Image
    from ecamp_amd.module import pretrain_datasets as pd
    torch.set_num_threads(1)
    tmp = tempfile.mkdtemp(prefix="ecamp_loader_")
    paths, raws = [], []
    for k in range(4):
        a = radiograph(args.width, args.height, k)
        raws.append(a)
        p = os.path.join(tmp, "img%d.jpg" % k)
        Image.fromarray(a, "L").save(p, quality=95)
        paths.append(p)
    jpg_mb = np.mean([os.path.getsize(p) for p in paths]) / 1e6
    full, small = os.path.join(tmp, "full.u8"), os.path.join(tmp, "s1024.u8")
    with pd.U8ShardWriter(full) as wf, pd.U8ShardWriter(small, max_side=1024) as ws:
        for a in raws:
            wf.add(a)
            ws.add(a)
    rf, rs = pd.U8ShardReader(full), pd.U8ShardReader(small)
    tf32, tf8 = pd.default_image_transform(448), pd.default_image_transform(448, image_u8=True)

    def timed(fn):
        torch.manual_seed(0)
        fn(0)
        t0 = time.perf_counter()
        for i in range(args.n):
            fn(i)
        return (time.perf_counter() - t0) / args.n

    t_dec = timed(lambda i: pd.pil_loader(paths[i % 4]))
    t_host = timed(lambda i: tf32(pd.pil_loader(paths[i % 4])))
    t_u8 = timed(lambda i: tf8(pd.pil_loader(paths[i % 4])))
    bytes_f, bytes_s = [], []

    def crop_full(i):
        c, _ = pd.device_crop_item(rf[i % 4])
        bytes_f.append(c.size)

    def crop_small(i):
        c, _ = pd.device_crop_item(rs[i % 4])
        bytes_s.append(c.size)

    t_cf, t_cs = timed(crop_full), timed(crop_small)
    W = args.workers
    print("# loader rate per worker process, %d x %d grayscale JPEGs (%.2f MB each, quality 95), Pillow %s, one thread; %d items per row"
          % (args.width, args.height, jpg_mb, __import__("PIL").__version__, args.n))
    print("# CPU: %s" % next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"))
    rows = [("open + JPEG decode + convert('RGB') alone", t_dec, None),
            ("host item, f32 [3,448,448] (the reference's transform)", t_host, 3 * 448 * 448 * 4),
            ("host item, uint8 [448,448] (--image_u8)", t_u8, 448 * 448),
            ("device path, full-size shard: box + flip + copy of the box", t_cf, float(np.mean(bytes_f))),
            ("device path, max_side 1024 shard: box + flip + copy of the box", t_cs, float(np.mean(bytes_s)))]
    print("%-66s %10s %12s %16s %14s" % ("path", "ms/item", "items/s", "pairs/s @%dw" % W, "H2D MB/pair"))
    for name, t, nbytes in rows:
        print("%-66s %10.2f %12.1f %16.0f %14s" % (name, 1e3 * t, 1.0 / t, W / t, "-" if nbytes is None else "%.2f" % (nbytes / 1e6)))
    need = args.step_rate
    print("# one MI355X trains at ~%.0f pairs/s (bench.py): the reference's host transform needs %.0f worker processes per GPU to keep up, "
          "the device path %.1f (full-size shards; %.1f GB/s of host -> HBM copies) or %.1f (max_side 1024; %.1f GB/s)"
          % (need, need * t_host, need * t_cf, need * np.mean(bytes_f) / 1e9, need * t_cs, need * np.mean(bytes_s) / 1e9))
    for p in paths + [full, full + ".idx.npy", small, small + ".idx.npy"]:
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
