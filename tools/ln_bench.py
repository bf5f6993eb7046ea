#!/usr/bin/env python3
"""LayerNorm fwd/bwd micro-benchmark: achieved HBM GB/s vs algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for rows, cols in [(12800, 768), (32768, 768), (50432, 512)]:
    x = torch.randn(rows, cols, device=dev).bfloat16(); r = torch.randn_like(x); dy = torch.randn_like(x)
    g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
    gg, gb = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    y, z, m, s = o.layernorm_fwd(x, g, b, 1e-6)
    t1 = timeit(lambda: o.layernorm_fwd(x, g, b, 1e-6))
    t2 = timeit(lambda: o.layernorm_fwd(x, g, b, 1e-12, residual=r, drop_p=0.1, seed=1, offset=2))
    t3 = timeit(lambda: o.layernorm_bwd(dy, z, m, s, g, gg, gb, dres=r))
    t4 = timeit(lambda: o.layernorm_bwd(dy, z, m, s, g, gg, gb, drop_p=0.1, seed=1, offset=2, want_drop=True))
    e = rows * cols * 2
    print("rows %6d cols %4d | fwd %5.1fus %4.0f GB/s | fwd+res+drop %5.1fus %4.0f GB/s | bwd+dres %5.1fus %4.0f GB/s | bwd+drop %5.1fus %4.0f GB/s" %
          (rows, cols, t1, 2 * e / t1 / 1e3, t2, 4 * e / t2 / 1e3, t3, 4 * e / t3 / 1e3, t4, 4 * e / t4 / 1e3))
