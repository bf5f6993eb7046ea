#!/usr/bin/env python3
"""fp8 forward GEMM (quantise activations + v_mfma_scale 16x16x128) against the bf16 forward on the ViT shapes of configs[4] (B=512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o
dev = torch.device("cuda:0")
SHAPES = [("enc qkv", 25600, 2304, 768), ("enc proj", 25600, 768, 768), ("enc fc1", 25600, 3072, 768), ("enc fc2", 25600, 768, 3072),
          ("dec qkv", 100864, 1536, 512), ("dec fc1", 100864, 2048, 512), ("dec fc2", 100864, 512, 2048)]
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("%-10s %7s %5s %5s | %14s %14s %14s" % ("shape", "M", "N", "K", "bf16 fwd", "fp8 gemm only", "fp8 incl. quant"))
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev)
    w8, ws = o.quantize_fp8(w); x8, xs = o.quantize_fp8(x); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t0 = timeit(lambda: o.linear_fwd(x, w, b))
    t1 = timeit(lambda: o.call("ecamp_gemm_fp8", o.ptr(x8), o.ptr(w8), o.ptr(y), M, N, K, K, K, N, o.ptr(xs), o.ptr(ws), o.ptr(b), o.ptr(None), 0, o.ptr(None), N, 0, o.ptr(None), o.ptr(None), o.ptr(None), o.stream()))
    t2 = timeit(lambda: o.linear_fwd_fp8(x, w8, ws, b))
    print("%-10s %7d %5d %5d | %5.0f TF %4.0fus %5.0f TF %4.0fus %5.0f TF %4.0fus" % (name, M, N, K, fl / t0 / 1e9, t0 * 1e3, fl / t1 / 1e9, t1 * 1e3, fl / t2 / 1e9, t2 * 1e3))
