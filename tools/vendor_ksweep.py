#!/usr/bin/env python3
"""Yardstick: the per-K-tile slope and the per-launch/per-tile intercept of the vendor library's forward GEMM (torch -> hipBLASLt; never
linked or called by the product) against the persistent Q8 kernel, on one exact round of 256 tiles (4096 x 4096, K swept) and on three
exact rounds at the model's K.  T(K) = a + b * K/64: b is the loop, a is launch + ramp + epilogue.

    python3 tools/vendor_ksweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


ZERO = "--zero" in sys.argv   # all-zero operands: the matrix pipes draw far less power (is the gap power or structure?)
for rep in range(2):
    for (M, N, K) in [(4096, 4096, 256), (4096, 4096, 512), (4096, 4096, 1024), (4096, 4096, 2048), (4096, 4096, 4096), (12288, 4096, 768),
                      (32768, 1536, 768), (12800, 3072, 768)]:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        if ZERO:
            x.zero_(); w.zero_()
        b = torch.randn(N, device=dev)
        bb = b.bfloat16()
        tq = timeit(lambda: o.linear_fwd(x, w, b))
        tl = timeit(lambda: torch.nn.functional.linear(x, w, bb))
        tn = timeit(lambda: torch.nn.functional.linear(x, w))
        print(("ZERO " if ZERO else "") + "M=%-6d N=%-5d K=%-5d  q8 %7.1f us   lib(bias) %7.1f us   lib(no bias) %7.1f us   tiles %d" % (M, N, K, tq, tl, tn, ((M + 255) // 256) * ((N + 255) // 256)), flush=True)
