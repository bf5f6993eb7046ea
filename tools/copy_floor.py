#!/usr/bin/env python3
"""Development yardstick: what does the plainest memory-bound kernel cost at LayerNorm's sizes, back to back over rotating buffers
(> 600 MB, nothing cache-resident)?  torch's copy / add kernels against ecamp layernorm_fwd on the same shapes.
    python3 tools/copy_floor.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o

dev = torch.device("cuda:0")
for rows, cols in ((12800, 768), (32768, 768), (50432, 512)):
    n = max(4, int(700e6 / (rows * cols * 2)) + 1)
    xs = [torch.randn(rows, cols, device=dev).bfloat16() for _ in range(n)]
    ys = [torch.empty_like(x) for x in xs]
    g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)

    def timeit(fn, reps=3):
        fn(0); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(reps):
            for i in range(n):
                fn(i)
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / (reps * n) * 1e3
    t_copy = timeit(lambda i: ys[i].copy_(xs[i]))
    t_add = timeit(lambda i: torch.add(xs[i], xs[(i + 1) % n], out=ys[i]))
    t_ln = timeit(lambda i: o.layernorm_fwd(xs[i], g, b, 1e-6))
    mb = rows * cols * 2 / 1e6
    print("%6d x %4d bf16 (%5.1f MB): copy %5.1f us (%.2f TB/s)   add (2 reads) %5.1f us (%.2f TB/s)   layernorm_fwd %5.1f us (%.2f TB/s)" % (
        rows, cols, mb, t_copy, 2 * mb / t_copy, t_add, 3 * mb / t_add, t_ln, 2 * mb / t_ln), flush=True)
