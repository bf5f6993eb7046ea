#!/usr/bin/env python3
"""BASELINE.json configs[4] (ViT-B/16, B = 512 per GPU, fp8-e4m3 forward GEMMs with bf16 gradients) against the bf16 step of the same size:
ms per step, socket power (rocm-smi, sampled once a second while the steps run) and joules per step.

    python tools/fp8_energy.py [--seconds 12]
"""
import argparse, json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda:0")


def power_sampler(stop, out):
    while not stop.is_set():
        try:
            s = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            w = re.search(r"Socket[^\n]*?([0-9]+\.[0-9]+)", s)
            c = re.search(r"sclk[^\n]*?\(([0-9]+)Mhz\)", s)
            if w:
                out.append((float(w.group(1)), float(c.group(1)) if c else 0.0))
        except Exception:
            pass
        stop.wait(1.0)


def run(name, **kw):
    torch.manual_seed(0)
    model = model_ecamp.ecamp(compute_dtype=torch.bfloat16, **kw).to(dev)
    model.prepare()
    model.train()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount()
    batch = synthetic_batch(args.batch, 128, 448, seed=0, device=dev)

    def step():
        mim, res, mlm = model(batch)
        scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        return mim, res, mlm
    for _ in range(8):
        out = step()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=power_sampler, args=(stop, samples), daemon=True)
    t0 = time.perf_counter()
    n = 0
    th.start()
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(4):
            out = step()
        n += 4
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    stop.set()
    th.join(timeout=15)
    samples = samples[2:] or samples     # the first samples still see the idle socket
    w = sum(s[0] for s in samples) / max(len(samples), 1)
    mhz = sum(s[1] for s in samples) / max(len(samples), 1)
    r = {"config": name, "pairs_per_gpu": args.batch, "steps": n, "ms_per_step": round(1e3 * dt, 2), "pairs_per_s": round(args.batch / dt, 1),
         "socket_watts": round(w, 0), "shader_mhz": round(mhz, 0), "joules_per_step": round(w * dt, 1), "joules_per_pair": round(w * dt / args.batch, 4),
         "power_samples": len(samples), "losses": [round(float(t), 4) for t in out]}
    print(json.dumps(r), flush=True)
    del model, opt, batch
    torch.cuda.empty_cache()
    return r


a = run("bf16, B=%d" % args.batch)
b = run("fp8 forward (e4m3 GEMM operands, bf16 gradients), B=%d" % args.batch, fp8_forward=True)
print(json.dumps({"fp8_over_bf16_speed": round(a["ms_per_step"] / b["ms_per_step"], 4), "fp8_over_bf16_joules": round(b["joules_per_step"] / a["joules_per_step"], 4)}))
