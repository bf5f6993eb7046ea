#!/usr/bin/env python3
"""Development timing of the fused AdamW over an arena of the model's size (183 M parameters, random state): python3 tools/adamw_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
n = 183173120
p, g, m = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-2, torch.randn(n, device=dev) * 1e-3
v = torch.rand(n, device=dev) * 1e-4
p16 = torch.empty(n, device=dev, dtype=torch.bfloat16)
grp = torch.zeros(n // 64, device=dev, dtype=torch.uint8)
ss = torch.zeros(1, device=dev)
from ecamp_amd import hip_ops as o
def run(step):
    o.adamw_grouped(p, g, m, v, p16, grp, [1e-4, 1e-4], [0.05, 0.0], 0.9, 0.95, 1e-8, step, 1.0, ss)
for rep in range(3):
    run(1); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10):
        run(2 + i)
    e.record(); torch.cuda.synchronize()
    t = a.elapsed_time(e) / 10
    print("adamw_grouped %.3f ms  (%.2f TB/s at 30 B per parameter)" % (t, n * 30 / t / 1e9), flush=True)
