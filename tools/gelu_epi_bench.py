#!/usr/bin/env python3
"""Development timing of the GELU epilogues: forward act = 1 (save the pre-activation) against act = 2 (save gelu'), and the data gradient
multiplying by gelu'(pre) against multiplying by the saved derivative, on the fc1 / fc2 shapes of BASELINE configs[1].
    python3 tools/gelu_epi_bench.py"""
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
    for name, M, D, Hd in (("enc", 12800, 768, 3072), ("bert", 32768, 768, 1536), ("dec", 50432, 512, 2048)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(M, D, generator=g).to(dev, torch.bfloat16)
        w1 = (torch.randn(Hd, D, generator=g) * D ** -0.5).to(dev, torch.bfloat16)
        b1 = torch.randn(Hd, generator=g).to(dev)
        w2 = (torch.randn(D, Hd, generator=g) * Hd ** -0.5).to(dev, torch.bfloat16)
        dy = torch.randn(M, D, generator=g).to(dev, torch.bfloat16)
        _, pre = o.linear_fwd(x, w1, b1, act=1, save_pre=True)
        _, der = o.linear_fwd(x, w1, b1, act=2, save_pre=True)
        t = [timeit(lambda: o.linear_fwd(x, w1, b1)), timeit(lambda: o.linear_fwd(x, w1, b1, act=1, save_pre=True)), timeit(lambda: o.linear_fwd(x, w1, b1, act=2, save_pre=True)),
             timeit(lambda: o.linear_dgrad(dy, w2)), timeit(lambda: o.linear_dgrad(dy, w2, gmul=pre)), timeit(lambda: o.linear_dgrad(dy, w2, gmul=der, gmul_is_grad=True))]
        u = torch.randn(M, Hd, generator=g).to(dev, torch.bfloat16)
        b2 = torch.randn(D, generator=g).to(dev)
        dpre = torch.randn(M, Hd, generator=g).to(dev, torch.bfloat16)
        t2 = [timeit(lambda: o.linear_fwd(u, w2, b2)), timeit(lambda: o.linear_fwd(u, w2, b2, residual=x)),
              timeit(lambda: o.linear_dgrad(dpre, w1)), timeit(lambda: o.linear_dgrad(dpre, w1, residual=x))]
        print("%-5s fc2 fwd: plain %6.1f  + residual %6.1f us   |  fc1 dgrad: plain %6.1f  + residual %6.1f us" % ((name,) + tuple(t2)), flush=True)
        print("%-5s fc1 fwd: plain %6.1f  gelu+pre %6.1f  gelu+gelu' %6.1f us   |  fc2 dgrad: plain %6.1f  * gelu'(pre) %6.1f  * saved gelu' %6.1f us" % ((name,) + tuple(t)), flush=True)
