#!/usr/bin/env python3
"""Every GEMM class of the bench step, inside the step and alone:

    python tools/gemm_in_step.py > gpurun_out/r05_gemm_in_step_vs_lab.txt

IN-STEP: the serialized bench step (configs[1], B=256, S=128, bf16; weight gradients and the image-decoder branch on the main stream, so every
launch runs alone behind the kernel it depends on) with the library's HIP-event brackets tagged per (form, epilogue, M, N, K)
(ecamp_prof_dump).  LAB: the same ecamp_gemm call repeated back to back, (a) on the same buffers ("warm": operands in L2 / Infinity Cache, no
dependent predecessor) and (b) rotating through copies of its activations that exceed the 256 MB Infinity Cache ("cold": operands from HBM
as in the step, still no dependent predecessor).  Columns: tiles of 256 x 256 (x split), rounds on 256 CUs and the share of the last
round that is idle; in-step minus lab-cold = what the dependent launch costs (ramp from a cold L2 behind a kernel boundary + drain);
lab-cold minus lab-warm = what operands from HBM instead of the caches cost; `quant` = time a perfectly balanced launch would save
(us x idle share of the rounds)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import _lib, hip_ops, optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount

dev = torch.device("cuda:0")
lib = _lib.load()


def in_step(steps=3):
    torch.manual_seed(42)
    model = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev)
    model.prepare()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount()
    batch = synthetic_batch(256, 128, 448, seed=0, device=dev)
    model.train()
    opt.zero_grad()

    def step():
        mim, res, mlm = model(batch)
        scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()

    hip_ops.OVERLAP_WGRAD = False
    hip_ops.OVERLAP_BRANCHES = False
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib.ecamp_prof_collect(-1, None, None, None)
    lib.ecamp_prof_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    lib.ecamp_prof_enable(0)
    n = lib.ecamp_prof_dump(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    lib.ecamp_prof_dump(buf, len(buf))
    rows = {}
    for line in buf.value.decode().splitlines():
        tag, cnt, ms, fl = line.split()
        rows[tag] = (int(cnt) / steps, 1e3 * float(ms) / int(cnt), float(fl) / int(cnt))
    del model, opt, batch
    torch.cuda.empty_cache()
    return rows


def lab(tag, warm, n=20, q16=None):
    """The call of `tag` alone, back to back.  q16: None = the kernel family of the tag, 0 / 1 = force the eight-wave / the four-wave kernel."""
    kind, form, epi, M, N, K, sp = tag.split(":")
    if kind not in ("q8", "q16"):
        return None
    hip_ops.set_option("q16_mode", (1 if kind == "q16" else 0) if q16 is None else q16)
    M, N, K, e = int(M), int(N), int(K), int(epi[1:])
    split = int(sp[1:]) if sp[0] == "s" else 1
    bf = torch.bfloat16
    # enough copies of the activation operands to exceed the Infinity Cache when `warm` is False
    per = (M * K + M * N) * 2 if form != "w" else (K * M + K * N) * 2
    nc = 1 if warm else max(2, int(400e6 // per) + 1)
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(bf)
    if form == "f":      # y[M,N] = x[M,K] w[N,K]^T
        A = [rnd(M, K) for _ in range(nc)]; Bm = rnd(N, K) * (K ** -0.5); C = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(nc)]
        bias = torch.zeros(N, device=dev); res = [rnd(M, N) for _ in range(nc)] if e == 2 else None
        pre = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(nc)] if e == 1 else None
        call = lambda i: hip_ops.gemm(A[i], Bm, C[i], M, N, K, True, K, True, K, N, bias=bias, residual=res[i] if res else None, ldr=N,
                                      pre_out=pre[i] if pre else None, ldp=N, act=2 if e == 1 else 0)
    elif form == "d":    # dx[M,N] = dy[M,K] w[K,N]
        A = [rnd(M, K) for _ in range(nc)]; Bm = rnd(K, N) * (K ** -0.5); C = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(nc)]
        res = [rnd(M, N) for _ in range(nc)] if e in (2, 3) else None
        gm = [rnd(M, N) for _ in range(nc)] if e == 3 else None
        call = lambda i: hip_ops.gemm(A[i], Bm, C[i], M, N, K, True, K, False, N, N, residual=res[i] if res else None, ldr=N,
                                      gmul=gm[i] if gm else None, ldg=N, act=2 if e == 3 else 0)
    else:                # dw[M,N] (f32) = dy[K,M]^T x[K,N]
        A = [rnd(K, M) for _ in range(nc)]; Bm = [rnd(K, N) for _ in range(nc)]; C = torch.zeros(M, N, device=dev)
        rs = torch.zeros(M, device=dev)
        call = lambda i: hip_ops.gemm(A[i], Bm[i], C, M, N, K, False, M, False, N, N, out_f32=True, accumulate=False, split_k=split, rowsum=rs)
    for i in range(nc):
        call(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        call(i % nc)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    rows = in_step()
    hip_ops.set_option("q8_mode", -1)
    only = os.environ.get("GEMM_IN_STEP_ONLY")   # e.g. "q16": lab columns for these tags only (shorter run)
    if only:
        rows = {k: v for k, v in rows.items() if k.startswith(only)}
    print("# GEMM classes of the bench step: in-step (serialized, behind their dependent predecessor) against the same call alone (back to back)")
    print("%-34s %5s %6s %6s %5s | %8s %8s %8s | %7s %7s %7s | %6s" % ("form:epi:M:N:K:split", "n/stp", "tiles", "rounds", "idle", "in-step", "lab cold", "lab warm",
                                                                 "dep us", "hbm us", "quant", "TF in"))
    tot = {"in": 0.0, "cold": 0.0, "warm": 0.0, "quant": 0.0, "n": 0.0}
    for tag, (nps, us, fl) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        parts = tag.split(":")
        if parts[0] not in ("q8", "q16"):
            print("%-34s %5.1f %6s %6s %5s | %8.1f %8s %8s |" % (tag[:34], nps, "-", "-", "-", us, "-", "-"))
            continue
        M, N, K = int(parts[3]), int(parts[4]), int(parts[5])
        split = int(parts[6][1:]) if parts[6][0] == "s" else 1
        tn = int(parts[6][1:]) if parts[6][0] == "w" else 256
        tiles = ((M + 255) // 256) * ((N + tn - 1) // tn) * split
        rounds = tiles / 256.0
        full = -(-tiles // 256)
        idle = 1.0 - rounds / full
        cold, warm = lab(tag, False), lab(tag, True)
        other = ""
        if parts[0] == "q16":   # the same call on the eight-wave kernel, for comparison
            other = "   [eight-wave kernel alone: cold %.1f warm %.1f]" % (lab(tag, False, q16=0), lab(tag, True, q16=0))
        quant = us * idle
        print("%-34s %5.1f %6d %6.2f %4.0f%% | %8.1f %8.1f %8.1f | %7.1f %7.1f %7.1f | %6.0f" % (tag[:34], nps, tiles, rounds, 100 * idle, us, cold, warm, us - cold, cold - warm,
                                                                                         quant, fl / us / 1e6) + other)
        tot["in"] += nps * us; tot["cold"] += nps * cold; tot["warm"] += nps * warm; tot["quant"] += nps * quant; tot["n"] += nps
    print("# per step: %.0f tagged launches; in-step %.2f ms; the same calls alone: cold %.2f ms, warm %.2f ms; dependent-launch cost %.2f ms; "
          "operands from HBM %.2f ms; partial last rounds %.2f ms" % (tot["n"], tot["in"] / 1e3, tot["cold"] / 1e3, tot["warm"] / 1e3,
                                                                        (tot["in"] - tot["cold"]) / 1e3, (tot["cold"] - tot["warm"]) / 1e3, tot["quant"] / 1e3))


if __name__ == "__main__":
    main()
