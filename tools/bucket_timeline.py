#!/usr/bin/env python3
"""When does backward finish each gradient bucket?  One GPU, no communication: the reducer's launch points are stamped with events
(the moment every stream that writes gradients has reached the point where the bucket's collective would start), and the exposed
tail of the exchange is priced for 2 / 4 / 8 GPUs with a serial communication stream at the per-GPU xGMI rates of DESIGN.md 7.
usage: bucket_timeline.py [bucket_mb [tail_bucket_mb [tail_span_mb]]]   (defaults: the wrapper's)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim, hip_ops
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.parallel import DistributedDataParallel
from ecamp_amd.util.misc import NativeScalerWithGradNormCount

dev = torch.device("cuda:0")
torch.manual_seed(0)
kw = {}
if len(sys.argv) > 1:
    kw["bucket_cap_mb"] = float(sys.argv[1])
if len(sys.argv) > 2:
    kw["tail_bucket_mb"] = float(sys.argv[2]) if float(sys.argv[2]) > 0 else None
if len(sys.argv) > 3:
    kw["tail_span_mb"] = float(sys.argv[3])
model = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev)
ddp = DistributedDataParallel(model, **kw)
red = ddp.reducer
model.train()
opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
scaler = NativeScalerWithGradNormCount()
batch = synthetic_batch(256, 128, 448, seed=0, device=dev)
probe = torch.cuda.Stream()
stamps = []

def stamp(tag):
    cur = torch.cuda.current_stream()
    probe.wait_stream(cur)
    if red.main_stream is not None:
        probe.wait_stream(red.main_stream)
    for s in hip_ops.side_streams(dev):
        probe.wait_stream(s)
    e = torch.cuda.Event(enable_timing=True)
    e.record(probe)
    stamps.append((tag, e))

orig_launch = red._launch
def launch(b):
    if not red.launched[b]:
        stamp(b)
    orig_launch(b)
red._launch = launch
orig_fin = red.finalize
def fin():
    if red.dirty:
        orig_fin(); stamp("end")
    else:
        orig_fin()
red.finalize = fin

def step():
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    mim, res, mlm = ddp(batch)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    return e0, e1, e2

for _ in range(4): step(); stamps.clear()
e0, e1, e2 = step()
torch.cuda.synchronize()
names = model.arena.names if hasattr(model.arena, "names") else None
t_end = [e for t, e in stamps if t == "end"][0]
print("forward %.2f ms, backward (to the last gradient) %.2f ms, step %.2f ms; %d buckets, largest %.1f MiB, last %.1f MiB  %s" %
      (e0.elapsed_time(e1), e1.elapsed_time(t_end), e0.elapsed_time(e2), len(red.buckets), max(hi - lo for lo, hi, _ in red.buckets) * 4 / 2 ** 20,
       (red.buckets[-1][1] - red.buckets[-1][0]) * 4 / 2 ** 20, kw))
rows = []
for tag, e in stamps:
    if tag == "end": continue
    lo, hi, slots = red.buckets[tag]
    rows.append((tag, (hi - lo) * 4 / 1e6, e1.elapsed_time(e), e.elapsed_time(t_end), slots))
for tag, mb, t, left, slots in rows:
    nm = ""
    if names is not None:
        nm = "%s ... %s" % (names[slots[0]], names[slots[-1]])
    print("bucket %2d %7.1f MB  complete %6.2f ms into backward, %6.2f ms before its end   %s" % (tag, mb, t, left, nm))
bw_end = e1.elapsed_time(t_end)
# per-GPU send rate of an all-reduce over the fully connected xGMI mesh: (N-1) links x ~76 GB/s per direction at 70 % efficiency
for N in (2, 4, 8):
    rate = (N - 1) * 76e9 * 0.7
    t = 0.0
    for tag, mb, ready, left, _ in rows:
        dur = 2.0 * (N - 1) / N * mb * 1e6 / rate * 1e3 + 0.05
        t = max(t, ready) + dur
    print("N=%d: exchange %.2f ms of link time in all, finishes %.2f ms after the last gradient (exposed)" %
          (N, sum(2.0 * (N - 1) / N * mb * 1e6 / rate * 1e3 + 0.05 for _, mb, _, _, _ in rows), max(0.0, t - bw_end)))
