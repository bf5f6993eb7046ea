#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Golden vectors for SURVEY.md 8(f) f3 (logging): the reference's own `SmoothedValue` / `MetricLogger`
(ECAMP/Pre-training/util/misc.py:24-167) are imported through oracle/ref_shim.py and fed a fixed series of losses and learning rates
the way `train_one_epoch` feeds them (main_pretrain.py:118-120,157-162: meters mim_loss / res_loss / mlm_loss with the default
window 20 and format, `lr` with window 1 and "{value:.6f}").  Written to tests/golden/meters.npz: the inputs, and after every update
the reference's median / avg / global_avg / max / value of each meter and the formatted `str(metric_logger)` line (utf-8 bytes).

    python oracle/make_golden_meters.py        # needs /root/reference (authoring container only)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "meters.npz")
NAMES = ("mim_loss", "res_loss", "mlm_loss")


def series(n=57, seed=7):
    """Loss-like values (decaying, noisy, with ties and a spike) as float32 -- what `.item()` of an f32 loss returns -- and an lr ramp."""
    g = np.random.default_rng(seed)
    t = np.arange(n)
    vals = np.stack([1.2 * np.exp(-t / 40.0) + 0.05 * g.standard_normal(n), 0.3 + 0.02 * g.standard_normal(n),
                     9.5 * np.exp(-t / 25.0) + 0.4 * g.standard_normal(n)], 1).astype(np.float32)
    vals[10] = vals[9]          # a tie inside the window (torch.median returns the lower middle value)
    vals[30, 2] = 40.0          # a spike: median and average part ways
    lr = (1.5e-4 * np.minimum(1.0, (t + 1) / 40.0)).astype(np.float64)
    return vals, lr


def main():
    ref_shim.install()
    sys.path.insert(0, ref_shim.REF_ROOT)
    import util.misc as ref_misc
    vals, lr = series()
    ml = ref_misc.MetricLogger(delimiter="  ")
    ml.add_meter("lr", ref_misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    stats, lines = [], []
    for i in range(len(vals)):
        ml.update(mim_loss=torch.tensor(vals[i, 0]), res_loss=float(vals[i, 1]), mlm_loss=torch.tensor(vals[i, 2]))
        ml.update(lr=float(lr[i]))
        row = []
        for k in NAMES + ("lr",):
            m = ml.meters[k]
            row.append([m.median, m.avg, m.global_avg, m.max, m.value])
        stats.append(row)
        lines.append(str(ml))
    width = max(len(s.encode()) for s in lines)
    arr = np.zeros((len(lines), width), dtype=np.uint8)
    lens = np.zeros(len(lines), dtype=np.int32)
    for i, s in enumerate(lines):
        b = s.encode()
        arr[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        lens[i] = len(b)
    np.savez_compressed(OUT, values=vals, lr=lr, stats=np.asarray(stats, dtype=np.float64), lines=arr, line_lens=lens,
                        names=np.array(NAMES + ("lr",)))
    print("wrote", OUT, "|", lines[-1])


if __name__ == "__main__":
    main()
