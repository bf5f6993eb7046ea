#!/usr/bin/env python3
"""LayerNorm fwd/bwd micro-benchmark: achieved HBM GB/s vs algorithmic bytes.  Every call works on a different copy of its tensors,
rotating through enough copies (> 600 MB per call site) that nothing is served by the 256 MB Infinity Cache -- the state the kernels find
between two GEMMs of a training step (a benchmark that re-runs one 20-50 MB tensor reads 5.4 TB/s the step never sees)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")


def timeit(fn, ncopy, n=40):
    for i in range(ncopy):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i % ncopy)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for rows, cols in [(12800, 768), (32768, 768), (50432, 512), (12608, 1024), (50240, 512)]:
    e = rows * cols * 2
    nc = max(2, int(700e6 // (3 * e)) + 1)
    xs = [torch.randn(rows, cols, device=dev).bfloat16() for _ in range(nc)]
    rs = [torch.randn(rows, cols, device=dev).bfloat16() for _ in range(nc)]
    dys = [torch.randn(rows, cols, device=dev).bfloat16() for _ in range(nc)]
    g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
    gg, gb = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    st = [o.layernorm_fwd(xs[i], g, b, 1e-6) for i in range(nc)]
    t1 = timeit(lambda i: o.layernorm_fwd(xs[i], g, b, 1e-6), nc)
    t2 = timeit(lambda i: o.layernorm_fwd(xs[i], g, b, 1e-12, residual=rs[i], drop_p=0.1, seed=1, offset=2), nc)
    t3 = timeit(lambda i: o.layernorm_bwd(dys[i], st[i][1], st[i][2], st[i][3], g, gg, gb, dres=rs[i]), nc)
    t4 = timeit(lambda i: o.layernorm_bwd(dys[i], st[i][1], st[i][2], st[i][3], g, gg, gb, drop_p=0.1, seed=1, offset=2, want_drop=True), nc)
    print("rows %6d cols %4d (%2d copies) | fwd %5.1fus %4.0f GB/s | fwd+res+drop %5.1fus %4.0f GB/s | bwd+dres %5.1fus %4.0f GB/s | bwd+drop %5.1fus %4.0f GB/s" %
          (rows, cols, nc, t1, 2 * e / t1 / 1e3, t2, 4 * e / t2 / 1e3, t3, 4 * e / t3 / 1e3, t4, 4 * e / t4 / 1e3))
