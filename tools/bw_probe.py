#!/usr/bin/env python3
"""HBM fill / copy bandwidth reference (torch kernels), to judge the GEMM epilogue's write stream."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (64, 256, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    f = t(lambda: x.zero_()); c = t(lambda: y.copy_(x)); r = t(lambda: x.sum())
    print("%5d MB: fill %.2f TB/s  copy(read+write) %.2f TB/s  read(sum) %.2f TB/s" % (mb, 2 * n / f / 1e12, 4 * n / c / 1e12, 2 * n / r / 1e12))
