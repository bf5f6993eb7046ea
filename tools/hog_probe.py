#!/usr/bin/env python3
"""Development probe: how does the training step react to a small co-resident kernel on another stream (a stand-in for RCCL's
all-reduce workgroups during backward)?  usage: hog_probe.py [n_hog_blocks] | hog_probe.py bwd   (env P8_RESERVE, Q8_BWD_GRID, P8_WGRAD)"""
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
if os.environ.get("Q8_BWD_GRID"):
    from ecamp_amd import hip_ops
    hip_ops.set_option("q8_bwd_grid", int(os.environ["Q8_BWD_GRID"]))
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
def run_bwd_hog(n, blocks, ms=10.0, threads=256):
    """The co-tenant only during backward, as the all-reduce is: it starts when the forward pass has finished (event) and spins `ms`."""
    side = torch.cuda.Stream()
    cyc = int(2.0e9 * 0.4 * ms / 300.0)
    def one():
        mim, res, mlm = model(batch)
        if blocks > 0:
            e = torch.cuda.Event(); e.record()
            side.wait_event(e)
            lib.ecamp_dev_spin(blocks, threads, cyc, ctypes.c_void_p(side.cuda_stream))
        scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
    one(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("p8_wgrad:", os.environ.get("P8_WGRAD", "1"), "reserve:", os.environ.get("P8_RESERVE", "0"), "q8_bwd_grid:", os.environ.get("Q8_BWD_GRID", "0"))
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    # calibrate the spin: one block alone
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.ecamp_dev_spin(1, 64, int(2.0e9 * 0.4 * 10.0 / 300.0), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); torch.cuda.synchronize()
    print("spin calibrated for 10 ms lasts %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    th = int(os.environ.get("HOG_THREADS", "256"))   # 1024: at most two co-tenant workgroups fit a CU, so n of them hold >= n/2 CUs
    for blocks in (0, 16, 32, 48, 64, 96, 128):
        print("co-tenant of %3d x %d threads for ~10 ms of backward: %.2f ms/step" % (blocks, th, run_bwd_hog(10, blocks, threads=th)))
    sys.exit(0)
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("no hog      : %.2f ms/step" % run(5, []))
s1 = torch.cuda.Stream()
print("1 x 1 wave        : %.2f ms/step" % run(5, [s1]))
print("1 kernel, %2d x 256: %.2f ms/step" % (nh, run(5, [s1], nh, 256)))
print("1 kernel, %2d x 256: %.2f ms/step" % (4 * nh, run(5, [s1], 4 * nh, 256)))
