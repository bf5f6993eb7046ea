#!/usr/bin/env python3
"""SR head micro-benchmark (B=256, 224^2 -> 448^2): f32 VALU stencils (mode 0) vs bf16 matrix cores (mode 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, R, win = 256, 224, 12
g = torch.Generator().manual_seed(0)
pred = torch.randn(B, 3, R, R, generator=g).to(dev); big = torch.randn(B, 3, 2 * R, 2 * R, generator=g).to(dev)
col = torch.randint(0, 3, (B,), generator=g).to(dev); row = torch.randint(0, 3, (B,), generator=g).to(dev)
w = [(torch.randn(3, 3, 3, 3, generator=g) * 0.3).to(dev), (torch.randn(3, generator=g) * 0.1).to(dev),
     (torch.randn(3, 3, 3, 3, generator=g) * 0.3).to(dev), (torch.randn(3, generator=g) * 0.1).to(dev)]
res = {}
for mode in (0, 1):
    s = torch.zeros(1, device=dev)
    o.sr_fwd(pred, big, col, row, *w, s, 32, win, mode)
    gw = torch.zeros(168, device=dev)
    try:
        dsr = o.sr_bwd(pred, big, col, row, *w, gw, 32, win, mode)
    except Exception as e:
        dsr = None
    res[mode] = (s.item(), gw.clone(), dsr)
    tf = timeit(lambda: o.sr_fwd(pred, big, col, row, *w, s, 32, win, mode))
    tb = timeit(lambda: o.sr_bwd(pred, big, col, row, *w, gw, 32, win, mode))
    print("mode %d: fwd %7.1f us  bwd %7.1f us  loss_sum %.6e" % (mode, tf, tb, res[mode][0]))
print("loss rel diff %.3e" % (abs(res[1][0] - res[0][0]) / abs(res[0][0])))
print("gw rel diff %.3e  dsr rel diff %.3e" % (float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max()),
                                                float((res[1][2] - res[0][2]).abs().max() / res[0][2].abs().max())))
