#!/usr/bin/env python3
"""Development probe: how does the training step react to a small co-resident kernel on another stream (a stand-in for RCCL's
all-reduce workgroups during backward)?  usage: hog_probe.py [n_hog_blocks]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev); model.prepare(); model.train()
opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
scaler = NativeScalerWithGradNormCount()
batch = synthetic_batch(256, 128, 448, seed=0, device=dev)
def step():
    mim, res, mlm = model(batch)
    scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()
if os.environ.get("P8_WGRAD") == "0":
    from ecamp_amd import hip_ops
    hip_ops.set_option("p8_wgrad", 0)
if os.environ.get("P8_RESERVE"):
    from ecamp_amd import hip_ops
    hip_ops.set_option("p8_wgrad_reserve_cus", int(os.environ["P8_RESERVE"]))
for _ in range(3): step()
torch.cuda.synchronize()
import ctypes
from ecamp_amd import _lib
lib = _lib.load()
def run(n, hog_streams, blocks=1, threads=64):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in hog_streams:
        lib.ecamp_dev_spin(blocks, threads, int(2.0e9 * 0.4), ctypes.c_void_p(s.cuda_stream))   # ~0.2-0.4 s spin
    for _ in range(n): step()
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return dt
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("p8_wgrad:", os.environ.get("P8_WGRAD", "1"), "reserve:", os.environ.get("P8_RESERVE", "0"))
print("no hog      : %.2f ms/step" % run(5, []))
s1 = torch.cuda.Stream()
print("1 x 1 wave        : %.2f ms/step" % run(5, [s1]))
print("1 kernel, %2d x 256: %.2f ms/step" % (nh, run(5, [s1], nh, 256)))
print("1 kernel, %2d x 256: %.2f ms/step" % (4 * nh, run(5, [s1], 4 * nh, 256)))
