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
# report-side embedding backward (LayerNorm backward + scatter into word / position / type rows)
Bt, S, H = 256, 128, 768
de = torch.randn(Bt * S, H, device=dev).bfloat16(); z = torch.randn(Bt * S, H, device=dev).bfloat16()
mean = torch.zeros(Bt * S, device=dev); rstd = torch.ones(Bt * S, device=dev); gamma = torch.ones(H, device=dev)
lens = torch.randint(32, 129, (Bt,), device=dev)
ids = torch.randint(5, 30000, (Bt, S), device=dev); ids[torch.arange(S, device=dev)[None] >= lens[:, None]] = 0
ids[:, 0] = 2; ids[torch.rand(Bt, S, device=dev) < 0.3] = 3
ty = torch.zeros(Bt, S, dtype=torch.int64, device=dev)
gw = torch.zeros(30000, H, device=dev); gp = torch.zeros(512, H, device=dev); gt = torch.zeros(2, H, device=dev); gg = torch.zeros(H, device=dev); gb = torch.zeros(H, device=dev)
t = timeit(lambda: o.bert_embed_bwd(de, z, mean, rstd, gamma, ids, ty, gw, gp, gt, gg, gb, Bt, S, H))
print("report embedding backward   %7.1f us  %6.0f GB/s" % (t, (de.numel() * 4 + de.numel() * 4) / t / 1e3))
# vocabulary cross-entropy, forward + gradient in place over the bf16 logits
M, V = 32768, 30000
logits = torch.randn(M, V, device=dev).bfloat16(); labels = torch.randint(0, V, (M,), device=dev); wts = torch.ones(M, device=dev); ls = o.zeros((1,), dev)
src = logits.clone()
def ce():
    logits.copy_(src); o.ce_fwd_bwd_(logits, labels, wts, ls)
tc = timeit(lambda: logits.copy_(src)); t = timeit(ce) - tc
print("weighted CE fwd+bwd (in place) %6.1f us  %6.0f GB/s (one read + one write of %d MB)" % (t, 2 * logits.numel() * 2 / t / 1e3, logits.numel() * 2 / 1e6))
