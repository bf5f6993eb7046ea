#!/usr/bin/env python3
"""Slack stagger of the persistent GEMMs (gemm_q8.h / gemm_q16.h, option "stagger" = percent of a tile time): the launches of the step with a
partial last round, alone, back to back on rotating buffers (cold operands), stagger 0 / 25 / 50 / 75 -- microseconds and bit-identity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
bf = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(3): fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cases = [("f:e1 enc fc1   12800x3072x768", 12800, 3072, 768, "fwd_gelu"), ("d:e3 enc fc2'  12800x3072x768", 12800, 3072, 768, "dgrad_gelu"),
         ("f:e0 enc qkv   12800x2304x768", 12800, 2304, 768, "fwd"), ("f:e1 dec fc1   50432x2048x512", 50432, 2048, 512, "fwd_gelu"),
         ("d:e3 dec fc2'  50432x2048x512", 50432, 2048, 512, "dgrad_gelu"), ("f:e0 bert qkv  32768x2304x768", 32768, 2304, 768, "fwd"),
         ("f:e2 enc fc2   12800x768x3072", 12800, 768, 3072, "fwd_res"), ("f:e1 bert inter 32768x1536x768 (exact rounds)", 32768, 1536, 768, "fwd_gelu")]
R = 6   # rotating copies so that operands come from HBM
for name, M, N, K, kind in cases:
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(M, K if kind != "dgrad_gelu" else K, generator=g).to(dev, bf) for _ in range(R)]
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev, bf)
    b = torch.randn(N, generator=g).to(dev)
    res = [torch.randn(M, N, generator=g).to(dev, bf) for _ in range(R)]
    row, ref = [], None
    for pct in (0, 25, 50, 75):
        o.set_option("stagger", pct)
        if kind == "fwd": fn = lambda i: o.linear_fwd(xs[i % R], w, b)
        elif kind == "fwd_res": fn = lambda i: o.linear_fwd(xs[i % R], w, b, residual=res[i % R])
        elif kind == "fwd_gelu": fn = lambda i: o.linear_fwd(xs[i % R], w, b, act=2, save_pre=True)
        else:   # dx[M, N] = (dy[M, K] w2[K, N]) * gelu'[M, N]
            w2 = w.t().contiguous()
            fn = lambda i: o.linear_dgrad(xs[i % R], w2, gmul=res[i % R], gmul_is_grad=True)
        out = fn(0)
        out = out[0] if isinstance(out, tuple) else out
        if ref is None: ref = out.clone()
        same = torch.equal(out, ref)
        row.append("%3d%%: %6.1f us%s" % (pct, timeit(fn), "" if same else " DIFF"))
    print("%-48s %s" % (name, "   ".join(row)))
o.set_option("stagger", -1)
