#!/usr/bin/env python3
"""HBM-bound image-side kernels at configs[1] sizes: microseconds and effective GB/s (algorithmic bytes / time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, R, p = 256, 224, 16
L = (R // p) ** 2
big = torch.randn(B, 3, 2 * R, 2 * R, device=dev)
t = timeit(lambda: o.bicubic_resize(big, R, R))
print("bicubic 448->224           %7.1f us  %6.0f GB/s" % (t, (big.numel() * 4 + B * 3 * R * R * 4) / t / 1e3))
imgs = o.bicubic_resize(big, R, R)
pred = torch.randn(B * (L + 1), p * p * 3, device=dev).bfloat16()
mask = (torch.rand(B, L, device=dev) < 0.75).float()
s = o.zeros((2,), dev)
t = timeit(lambda: o.unpatchify_mim(pred, imgs, mask, s[0:], B, R, p))
print("unpatchify + masked MSE     %7.1f us  %6.0f GB/s" % (t, (pred.numel() * 2 + imgs.numel() * 8) / t / 1e3))
pred_img = o.unpatchify_mim(pred, imgs, mask, s[0:], B, R, p)
dsr = torch.randn(B, 3, R, R, device=dev)
gm = torch.tensor([1e-3, 1e-3], device=dev)
t = timeit(lambda: o.img_loss_bwd(pred_img, imgs, mask, dsr, gm, B, R, p, torch.bfloat16))
print("image-loss backward         %7.1f us  %6.0f GB/s" % (t, (imgs.numel() * 12 + pred.numel() * 2) / t / 1e3))
