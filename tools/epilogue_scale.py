#!/usr/bin/env python3
"""Is the epilogue of the persistent GEMM bound per CU or chip-wide?

    python tools/epilogue_scale.py > gpurun_out/epilogue_scale.txt

The data-gradient form of the eight-wave kernel on 1024 output tiles (M = 16384, N = 4096) with G = 256 / 128 / 64 / 32 workgroups
(option "q8_bwd_grid": every workgroup walks 1024 / G tiles) and K = 512 / 1024 / 2048 / 4096.  Per workgroup and tile the time is
a + b x (K / 64): b = one K tile of the loop, a = everything that happens once per tile (cross-tile prologue, the 128 KB of stores).
If `a` falls when fewer CUs store at the same time, the stores are bound by something the CUs share (L2 write-back, the fabric, HBM) and
workgroups that reach their epilogues at different times would see a faster one; if it does not, the limit is the CU's own store path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops

dev = torch.device("cuda:0")
bf = torch.bfloat16
M, N = 16384, 4096
TILES = (M // 256) * (N // 256)


def run(G, K, epi, n=12):
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(bf)
    A, Bm, C = rnd(M, K), rnd(K, N) * (K ** -0.5), torch.empty(M, N, device=dev, dtype=bf)
    res = rnd(M, N) if epi == 2 else None
    gm = rnd(M, N) if epi == 3 else None
    hip_ops.set_option("q8_bwd_grid", G)
    call = lambda: hip_ops.gemm(A, Bm, C, M, N, K, True, K, False, N, N, residual=res, ldr=N, gmul=gm, ldg=N, act=2 if epi == 3 else 0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    hip_ops.set_option("q16_mode", 0)
    hip_ops.set_option("q8_mode", 2)
    Ks = (512, 1024, 2048, 4096)
    for epi in (0, 2, 3):
        print("# data-gradient form, %s epilogue, %d tiles of 256 x 256; us per launch | us per workgroup and tile" % ({0: "plain", 2: "residual", 3: "x saved gelu'"}[epi], TILES))
        print("%5s | %s | %9s %9s" % ("G", " ".join("K=%-14d" % k for k in Ks), "a (us)", "b (us/Kt)"))
        for G in ((256, 64) if os.environ.get("EPI_SCALE_SHORT") else (256, 128, 64, 32)):
            ts = [run(G, K, epi) for K in Ks]
            per = [t * G / TILES for t in ts]
            # least squares a + b * ktiles
            xs = [k / 64 for k in Ks]
            n = len(xs)
            sx, sy = sum(xs), sum(per)
            sxx, sxy = sum(x * x for x in xs), sum(x * y for x, y in zip(xs, per))
            b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
            a = (sy - b * sx) / n
            print("%5d | %s | %9.2f %9.3f" % (G, " ".join("%7.1f %6.2f" % (t, p) for t, p in zip(ts, per)), a, b))
    hip_ops.set_option("q8_bwd_grid", 0)


if __name__ == "__main__":
    main()
