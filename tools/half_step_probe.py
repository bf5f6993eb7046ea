#!/usr/bin/env python3
"""How much of the forward pass's non-GEMM time could hide under GEMMs if TWO independent half-batches ran side by side?

    python tools/half_step_probe.py            (ECAMP_GEMM_GRID_CAP=192 python tools/half_step_probe.py: the GEMMs leave 64 CUs free)

Forward pass only, train mode, no autograd, configs[1] (B = 256, S = 128, bf16):
  A  one batch of 256 on one stream (every kernel behind the one before)
  B  the product: one batch of 256, image decoder and report side on two streams
  C  two batches of 128 on two streams, each a whole forward pass (its branches on its own stream)
Inside one batch every kernel depends on the one before, so only C lets a LayerNorm / attention / cross-entropy kernel of one half run beside a
GEMM of the other.  The GEMMs of a half batch are half as tall (the encoder's 768-wide outputs: 75 tiles of 256 x 256 / 100 of 256 x 192).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp

dev = torch.device("cuda:0")


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    torch.manual_seed(0)
    model = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev)
    model.prepare()
    model.train()
    full = synthetic_batch(256, 128, 448, seed=0, device=dev)
    h1 = synthetic_batch(128, 128, 448, seed=1, device=dev)
    h2 = synthetic_batch(128, 128, 448, seed=2, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def one():
        with torch.no_grad():
            model(full)

    def halves():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.no_grad():
            with torch.cuda.stream(s1):
                model(h1)
            with torch.cuda.stream(s2):
                model(h2)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def halves_serial():
        with torch.no_grad():
            model(h1)
            model(h2)

    hip_ops.OVERLAP_BRANCHES = False
    ta = timeit(one)
    tc = timeit(halves)
    td = timeit(halves_serial)
    hip_ops.OVERLAP_BRANCHES = True
    tb = timeit(one)
    print("# forward pass, B = 256 in all (ms): cap on persistent GEMM grids = %s" % os.environ.get("ECAMP_GEMM_GRID_CAP", "none"))
    print("A  one batch, one stream                         %.2f" % ta)
    print("B  one batch, decoder and report side overlapped  %.2f   (the product)" % tb)
    print("C  two half batches on two streams                %.2f" % tc)
    print("D  two half batches one after the other           %.2f   (what half-size kernels cost by themselves)" % td)


if __name__ == "__main__":
    main()
