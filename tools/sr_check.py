#!/usr/bin/env python3
"""SR head modes against torch autograd on a small case: d loss / d pred_img and the conv weight gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
B, R, win = 3, 64, 3
g = torch.Generator().manual_seed(0)
pimg = torch.randn(B, 3, R, R, generator=g); big = torch.randn(B, 3, 2 * R, 2 * R, generator=g)
col, row = torch.tensor([0, 1, 1]), torch.tensor([1, 0, 1])
ws = [(torch.randn(3, 3, 3, 3, generator=g) * 0.3), (torch.randn(3, generator=g) * 0.1), (torch.randn(3, 3, 3, 3, generator=g) * 0.3), (torch.randn(3, generator=g) * 0.1)]
pr = pimg.clone().requires_grad_(True); wr = [w.clone().requires_grad_(True) for w in ws]
u = F.interpolate(pr, scale_factor=2, mode="bilinear", align_corners=False)
sr = F.relu(F.conv2d(F.relu(F.conv2d(u, wr[0], wr[1], padding=1)), wr[2], wr[3], padding=1) + u)
G = 2 * R // 32
sm = torch.zeros(B, G, G)
for i in range(B): sm[i, col[i]:col[i] + win, row[i]:row[i] + win] = 1
spm = torch.kron(sm, torch.ones(32, 32))[:, None].expand(-1, 3, -1, -1)
loss = 0.5 * ((sr * spm - big * spm) ** 2).sum()
loss.backward()
wd = [w.to(dev).contiguous() for w in ws]
for mode in (0, 1):
    s = torch.zeros(1, device=dev)
    o.sr_fwd(pimg.to(dev), big.to(dev), col.to(dev), row.to(dev), *wd, s, 32, win, mode)
    gw = torch.zeros(168, device=dev)
    dsr = o.sr_bwd(pimg.to(dev), big.to(dev), col.to(dev), row.to(dev), *wd, gw, 32, win, mode).cpu()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print("mode %d: loss rel %.2e | dsr rel %.2e | dW1 %.2e db1 %.2e dW2 %.2e db2 %.2e" % (
        mode, abs(s.item() - 2 * loss.item()) / (2 * loss.item()), rel(dsr, pr.grad), rel(gw[0:81].cpu(), wr[0].grad.view(-1)),
        rel(gw[81:84].cpu(), wr[1].grad), rel(gw[84:165].cpu(), wr[2].grad.view(-1)), rel(gw[165:168].cpu(), wr[3].grad)))
    if mode == 1:
        d = (dsr - pr.grad).abs()
        idx = d.flatten().argmax().item()
        b_, c_, y_, x_ = idx // (3 * R * R), (idx // (R * R)) % 3, (idx // R) % R, idx % R
        print("  worst at b=%d c=%d y=%d x=%d: got %.4f want %.4f ; mean abs err %.3e (mean |grad| %.3e)" % (b_, c_, y_, x_, dsr.flatten()[idx], pr.grad.flatten()[idx], d.mean(), pr.grad.abs().mean()))
