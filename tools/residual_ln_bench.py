#!/usr/bin/env python3
"""Development timing: where should a pre-LN block's residual add live?
   A  (today)   x1 = Linear(a) + x in the GEMM epilogue;  h = LayerNorm(x1)
   B            y  = Linear(a);                           h, x1 = LayerNorm(y + x) with the add (and the x1 output) inside LayerNorm
on the proj (K = D) and fc2 (K = 4 D) layers of the encoder and decoder of BASELINE configs[1].   python3 tools/residual_ln_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o

dev = torch.device("cuda:0")


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


for rep in range(2):
    for name, M, D, K in (("enc proj", 12800, 768, 768), ("enc fc2", 12800, 768, 3072), ("dec proj", 50432, 512, 512), ("dec fc2", 50432, 512, 2048)):
        g = torch.Generator().manual_seed(1)
        a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
        w = (torch.randn(D, K, generator=g) * K ** -0.5).to(dev, torch.bfloat16)
        b = torch.randn(D, generator=g).to(dev)
        x = torch.randn(M, D, generator=g).to(dev, torch.bfloat16)
        gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)

        def A():
            x1 = o.linear_fwd(a, w, b, residual=x)
            return o.layernorm_fwd(x1, gam, bet, 1e-6)

        def B():
            y = o.linear_fwd(a, w, b)
            return o.layernorm_fwd(y, gam, bet, 1e-6, residual=x)
        ta, tb = timeit(A), timeit(B)
        tg, tgr = timeit(lambda: o.linear_fwd(a, w, b)), timeit(lambda: o.linear_fwd(a, w, b, residual=x))
        tl, tlr = timeit(lambda: o.layernorm_fwd(x, gam, bet, 1e-6)), timeit(lambda: o.layernorm_fwd(x, gam, bet, 1e-6, residual=a[:, :D].contiguous() if K == D else x))
        print("%-8s M=%d D=%d K=%d:  A (add in GEMM) %6.1f us   B (add in LayerNorm) %6.1f us   | GEMM %5.1f  GEMM+res %5.1f  LN %5.1f  LN+res %5.1f" % (name, M, D, K, ta, tb, tg, tgr, tl, tlr), flush=True)
