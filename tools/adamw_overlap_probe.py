#!/usr/bin/env python3
"""Development probe: can the optimizer pass (1.1 ms, HBM-bound) hide under the NEXT step's encoder forward (12 blocks whose 150- to
600-tile GEMMs leave 12-41 % of the CUs idle)?  Times the encoder forward alone, AdamW alone, one after the other, and AdamW on a
second stream in `chunks` slices beside the encoder forward."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim, hip_ops as ops
from ecamp_amd.module import model_ecamp
from ecamp_amd.functions import VitBlockFn
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev); A = m.prepare(); m.eval()
opt = optim.FusedAdamW(optim.add_weight_decay(m, 0.05), lr=1.5e-4, betas=(0.9, 0.95)); opt._bind()
A.flat_g.normal_(0, 1e-3)
B, T, D = 256, 50, 768
x = torch.randn(B * T, D, device=dev).bfloat16()
side = torch.cuda.Stream()

def enc():
    y = x
    for blk in m.blocks:
        y = VitBlockFn.apply(y, blk, m, B, T, m.num_heads)
    return y

def adamw(lo=0, hi=None):
    hi = A.total if hi is None else hi
    g0 = opt.param_groups[0]
    ops.adamw_grouped(A.flat_p[lo:hi], A.flat_g[lo:hi], opt._m[lo:hi], opt._v[lo:hi], A.flat_p16[lo:hi], opt._table[lo // 64:hi // 64],
                      [g["lr"] for g in opt.param_groups], [g["weight_decay"] for g in opt.param_groups], g0["betas"][0], g0["betas"][1], g0["eps"], 1)

def both_serial():
    adamw(); enc()

def both_overlap(chunks):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    step = (A.total // chunks + 63) // 64 * 64
    with torch.cuda.stream(side):
        for lo in range(0, A.total, step):
            adamw(lo, min(A.total, lo + step))
    enc()
    cur.wait_stream(side)

def timeit(fn, n=10):
    with torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

high = torch.cuda.Stream(priority=-1)
def in_high(fn):
    def run():
        with torch.cuda.stream(high):
            fn()
    return run
for r in range(2):
    print("encoder on a HIGH-priority stream, AdamW on a normal one: 1 launch %.3f ms, 8 slices %.3f, 32 slices %.3f, 128 slices %.3f | all normal priority, 128 slices %.3f" %
          (timeit(in_high(lambda: both_overlap(1))), timeit(in_high(lambda: both_overlap(8))), timeit(in_high(lambda: both_overlap(32))),
           timeit(in_high(lambda: both_overlap(128))), timeit(lambda: both_overlap(128))))
    print("encoder forward %.3f ms | AdamW %.3f ms | one after the other %.3f ms | AdamW beside the encoder: 1 launch %.3f ms, 8 slices %.3f ms, 32 slices %.3f ms" %
          (timeit(enc), timeit(adamw), timeit(both_serial), timeit(lambda: both_overlap(1)), timeit(lambda: both_overlap(8)), timeit(lambda: both_overlap(32))))
