#!/usr/bin/env python3
"""The four weight gradients of one transformer block: per-layer split-K launches + reduces against the grouped launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
NC = 4

def timeit(fn, rounds=4):
    for i in range(NC): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rounds):
        for i in range(NC): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (NC * rounds) * 1e3

for name, rows, shapes in [("encoder block", 12800, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                           ("decoder block", 50432, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]),
                           ("report layer ", 32768, [(768, 1536), (1536, 768), (768, 768), (2304, 768)])]:
    sets = []
    for c in range(NC):   # operands rotated so that they come from HBM, as in a step
        sets.append([(torch.randn(rows, n, device=dev).bfloat16(), torch.randn(rows, k, device=dev).bfloat16(), torch.zeros(n, k, device=dev),
                      torch.zeros(n, device=dev), False) for n, k in shapes])
    def per_layer(i):
        for dy, x, gw, gb, acc in sets[i]: o.linear_wgrad(dy, x, gw, gb=gb, accumulate=acc)
    def grouped(i):
        o.wgrad_group(sets[i])
    t0, t1 = timeit(per_layer), timeit(grouped)
    fl = sum(2.0 * rows * n * k for n, k in shapes)
    print("%s rows %5d: per layer %6.1f us (%4.0f TF)   grouped %6.1f us (%4.0f TF)" % (name, rows, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
    del sets
