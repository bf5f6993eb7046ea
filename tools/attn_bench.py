#!/usr/bin/env python3
"""Attention micro-benchmark on the hot path's shapes: us, TFLOP/s (4*B*H*Tq*Tk*hd fwd, 2.5x bwd) and GB/s of the minimal traffic."""
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

# name, B, H, T, hd, key-mask?, dropout
SHAPES = [("enc  T=50  hd=64 ", 256, 12, 50, 64, False, 0.0), ("dec  T=197 hd=32 ", 256, 16, 197, 32, False, 0.0),
          ("bert S=128 hd=128", 256, 6, 128, 128, True, 0.1), ("bert no-drop     ", 256, 6, 128, 128, True, 0.0)]
for name, B, H, T, hd, km, p in SHAPES:
    D = H * hd
    qkv = torch.randn(B * T, 3 * D, device=dev).bfloat16()
    f = qkv.view(-1)
    st = (T * 3 * D, 3 * D, hd)
    mask = (torch.arange(T, device=dev)[None] < torch.randint(T // 4, T + 1, (B, 1), device=dev)).int().contiguous() if km else None
    out, lse, bits = o.attn_fwd(f, f[D:], f[2 * D:], B, H, T, T, hd, st, st, st, hd ** -0.5, mask, p, 1, 2, want_mask=True)
    do = torch.randn_like(out)
    dqkv = torch.empty_like(qkv); df = dqkv.view(-1)
    t1 = timeit(lambda: o.attn_fwd(f, f[D:], f[2 * D:], B, H, T, T, hd, st, st, st, hd ** -0.5, mask, p, 1, 2, want_mask=True))
    t2 = timeit(lambda: o.attn_bwd(f, f[D:], f[2 * D:], out, do, lse, df, df[D:], df[2 * D:], B, H, T, T, hd, st, st, st, st, st, st, hd ** -0.5, mask, p, 1, 2, drop_bits=bits))
    if bits is not None:   # the same backward pass regenerating the mask from the Philox counters (three evaluations per score in all)
        t3 = timeit(lambda: o.attn_bwd(f, f[D:], f[2 * D:], out, do, lse, df, df[D:], df[2 * D:], B, H, T, T, hd, st, st, st, st, st, st, hd ** -0.5, mask, p, 1, 2))
        name = name + " (bwd with Philox regenerated: %.1f us)" % t3
    fl = 4.0 * B * H * T * T * hd
    e = B * T * D * 2
    print("%s | fwd %6.1fus %5.0f TF %5.0f GB/s(4 tensors) | bwd %6.1fus %5.0f TF %5.0f GB/s(8 tensors)" %
          (name, t1, fl / t1 / 1e6, 4 * e / t1 / 1e3, t2, 2.5 * fl / t2 / 1e6, 8 * e / t2 / 1e3))
