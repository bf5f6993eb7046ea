#!/usr/bin/env python3
"""What the fused epilogues of the forward / data-gradient GEMMs cost against the plain bf16 output on the hot path's shapes.
To keep the operands out of the 256 MB Infinity Cache between repetitions (as inside a training step) every repetition uses its
own copy of the activations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
NCOPY = 6

def timeit(fn, n=NCOPY, rounds=3):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rounds):
        for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * rounds) * 1e3

print("%-22s | %8s %8s %8s | %8s %8s %8s" % ("shape (M,N,K)", "fwd", "+resid", "+gelu/pre", "dgrad", "+resid", "*gelu'"))
for name, M, N, K in [("enc proj", 12800, 768, 768), ("enc fc1", 12800, 3072, 768), ("enc fc2", 12800, 768, 3072), ("dec fc1", 50432, 2048, 512),
                      ("dec fc2", 50432, 512, 2048), ("bert inter", 32768, 1536, 768), ("bert out", 32768, 768, 1536)]:
    xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NCOPY)]
    rs = [torch.randn(M, N, device=dev).bfloat16() for _ in range(NCOPY)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev)
    t0 = timeit(lambda i: o.linear_fwd(xs[i], w, b))
    t1 = timeit(lambda i: o.linear_fwd(xs[i], w, b, residual=rs[i]))
    t2 = timeit(lambda i: o.linear_fwd(xs[i], w, b, act=1, save_pre=True))
    # data gradient: dy [M,N] -> dx [M,K]
    rk = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NCOPY)]
    d0 = timeit(lambda i: o.linear_dgrad(rs[i], w))
    d1 = timeit(lambda i: o.linear_dgrad(rs[i], w, residual=rk[i]))
    d2 = timeit(lambda i: o.linear_dgrad(rs[i], w, gmul=rk[i]))
    print("%-10s %5d %5d %4d | %6.1fus %6.1fus %6.1fus | %6.1fus %6.1fus %6.1fus" % (name, M, N, K, t0, t1, t2, d0, d1, d2))
    del xs, rs, rk
