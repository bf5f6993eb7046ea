#!/usr/bin/env python3
"""Feasibility probe for HIP-graph capture of the training step: capture ONE steady-state step -- forward, backward on
the side streams, fused AdamW -- with torch.cuda.graph and replay it.  The captured step freezes everything the kernels take by value
(Philox offsets, learning rate, Adam's step count): this measures capturability and the host cost of a replay, it is not a training mode.
    python tools/graph_probe.py [--batch 32] [--steps 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev)
model.prepare(); model.train()
opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
scaler = NativeScalerWithGradNormCount()
batch = synthetic_batch(args.batch, 128, 448, seed=0, device=dev)


def step():
    mim, res, mlm = model(batch)
    scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()
    return mim, res, mlm


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n   # host enqueue ms per step, wall ms per step


for _ in range(6):
    out = step()
enq, wall = timed(step, args.steps)
print("B=%d eager : host enqueue %.2f ms/step, wall %.2f ms/step = %.0f pairs/s" % (args.batch, enq, wall, args.batch / wall * 1e3))
g = torch.cuda.CUDAGraph()
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = step()
except Exception as e:
    print("capture FAILED:", type(e).__name__, str(e)[:600])
    sys.exit(0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
enq, wall = timed(g.replay, args.steps)
print("B=%d graph : host enqueue %.2f ms/step, wall %.2f ms/step = %.0f pairs/s   losses %s" %
      (args.batch, enq, wall, args.batch / wall * 1e3, [round(float(t), 4) for t in out]))
