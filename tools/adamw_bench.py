#!/usr/bin/env python3
"""Development timing of the fused AdamW over an arena of the model's size (183 M parameters, random state), in both 16-bit builds, with
the host-side step arguments and with the device-side control block of the dynamic loss scaler (`ctl`): python3 tools/adamw_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import _lib, hip_ops as o

dev = torch.device("cuda:0")
n = 183173120
p, g, m = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-2, torch.randn(n, device=dev) * 1e-3
v = torch.rand(n, device=dev) * 1e-4
grp = torch.zeros(n // 64, device=dev, dtype=torch.uint8)
ss = torch.zeros(1, device=dev)
ctl = torch.tensor([1.0, 0.0, 0.1, 4.0], device=dev)
for half, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    _lib.set_half(half)
    p16 = torch.empty(n, device=dev, dtype=dt)
    for name, kw in (("host step, fused grad-norm", dict(grad_sumsq=ss)), ("host step", dict()), ("device ctl", dict(ctl=ctl)), ("device ctl, no 16-bit shadow", dict(ctl=ctl, shadow=False))):
        sh = kw.pop("shadow", True)
        def run(step):
            o.adamw_grouped(p, g, m, v, p16 if sh else None, grp, [1e-4, 1e-4], [0.05, 0.0], 0.9, 0.95, 1e-8, step, 1.0, **kw)
        best = 1e9
        for rep in range(3):
            run(1); torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(10):
                run(2 + i)
            e.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) / 10)
        print("%-5s %-32s %.3f ms  (%.2f TB/s at 30 B per parameter)" % (half, name, best, n * 30 / best / 1e9), flush=True)
