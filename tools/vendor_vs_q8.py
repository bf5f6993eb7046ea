#!/usr/bin/env python3
"""Yardstick run for tools/vendor_pmc.sh: ONE shape, the persistent Q8 kernel and the vendor library (torch -> hipBLASLt; never
linked or called by the product) on the same random operands, forward (x @ w^T + b) and data-gradient (dy @ w) forms.

    python3 tools/vendor_vs_q8.py "enc fc1" [--n 6]

Prints HIP-event times; under rocprofv3 the kernel trace / counter CSVs carry the per-kernel figures (tools/vendor_pmc_table.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o

SHAPES = {"enc qkv": (12800, 2304, 768), "enc fc1": (12800, 3072, 768), "bert inter": (32768, 1536, 768), "vocab": (32768, 30000, 768),
          "enc fc2": (12800, 768, 3072), "dec fc1": (50432, 2048, 512), "sq4096": (4096, 4096, 4096)}
name = sys.argv[1]
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 6
M, N, K = SHAPES[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
dy = torch.randn(M, N, device=dev).bfloat16()
b = torch.randn(N, device=dev)
bb = b.bfloat16()


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


fl = 2.0 * M * N * K
rows = [("q8 fwd", lambda: o.linear_fwd(x, w, b)), ("lib fwd", lambda: torch.nn.functional.linear(x, w, bb)),
        ("q8 dgrad", lambda: o.linear_dgrad(dy, w)), ("lib dgrad", lambda: torch.matmul(dy, w))]
for tag, fn in rows:
    t = timeit(fn)
    print("%-10s %-10s M=%d N=%d K=%d  %8.1f us  %6.0f TF" % (name, tag, M, N, K, t, fl / t / 1e6), flush=True)
