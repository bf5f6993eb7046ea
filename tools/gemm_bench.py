#!/usr/bin/env python3
"""Per-shape GEMM micro-benchmark on the hot path's shapes (fwd / dgrad / wgrad): TFLOP/s from HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o

dev = torch.device("cuda:0")
SHAPES = [("enc qkv", 12800, 2304, 768), ("enc proj", 12800, 768, 768), ("enc fc1", 12800, 3072, 768), ("enc fc2", 12800, 768, 3072),
          ("dec qkv", 50432, 1536, 512), ("dec proj", 50432, 512, 512), ("dec fc1", 50432, 2048, 512), ("dec fc2", 50432, 512, 2048),
          ("bert qkv", 32768, 2304, 768), ("bert dense", 32768, 768, 768), ("bert inter", 32768, 1536, 768), ("bert out", 32768, 768, 1536),
          ("vocab", 32768, 30000, 768)]
LIB = "--lib" in sys.argv   # also time the vendor library (torch.matmul -> hipBLASLt) on the same shapes, as a yardstick only
args = [a for a in sys.argv[1:] if a != "--lib"]
if args:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in args)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("%-12s %7s %6s %6s | %9s %9s %9s   (TFLOP/s; us)" % ("shape", "M", "N", "K", "fwd", "dgrad", "wgrad"))
tot = [0.0, 0.0, 0.0]
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    b = torch.randn(N, device=dev)
    gw = torch.zeros(N, K, device=dev)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: o.linear_fwd(x, w, b))
    t2 = timeit(lambda: o.linear_dgrad(dy, w))
    t3 = timeit(lambda: o.linear_wgrad(dy, x, gw))
    tot[0] += t1; tot[1] += t2; tot[2] += t3
    print("%-12s %7d %6d %6d | %5.0f %4.0fus %5.0f %4.0fus %5.0f %4.0fus" % (name, M, N, K, fl / t1 / 1e9, t1 * 1e3, fl / t2 / 1e9, t2 * 1e3, fl / t3 / 1e9, t3 * 1e3))
    if LIB:
        bb = b.bfloat16()
        l1 = timeit(lambda: torch.nn.functional.linear(x, w, bb))
        l2 = timeit(lambda: torch.matmul(dy, w))
        l3 = timeit(lambda: torch.matmul(dy.t(), x))
        print("%-12s %21s | %5.0f %4.0fus %5.0f %4.0fus %5.0f %4.0fus   <- hipBLASLt via torch (bf16 out, no fused epilogue work)"
              % ("", "", fl / l1 / 1e9, l1 * 1e3, fl / l2 / 1e9, l2 * 1e3, fl / l3 / 1e9, l3 * 1e3))
print("sum ms: fwd %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
