#!/usr/bin/env python3
"""`--amp fp16` soak: N training steps over a rotating set of synthetic batches (different report lengths, masks and images every step) with the
dynamic loss scaler; prints the scaler's state, the losses and whether every parameter is finite -- the run that found the padded-key NaN of the
attention backward (profiles/r06_soak_300_steps.txt).   python tools/fp16_soak.py [--steps 600] [--batch 256] [--model ecamp] [--lr 1.5e-4]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=600); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--model", default="ecamp")
ap.add_argument("--lr", type=float, default=1.5e-4); ap.add_argument("--nbatches", type=int, default=12); ap.add_argument("--growth_interval", type=int, default=200); ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
big = 896 if "448" in args.model else 448
model = getattr(model_ecamp, args.model)(compute_dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16).to(dev); model.prepare(); model.train()
opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=args.lr, betas=(0.9, 0.95))
scaler = NativeScalerWithGradNormCount(dynamic=True, growth_interval=args.growth_interval)     # (200, not 2000: the scale climbs within the run; 10: it hunts at half's ceiling)
batches = [synthetic_batch(args.batch, 128, big, seed=100 + i, device=dev) for i in range(args.nbatches)]
log = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(args.steps):
    mim, res, mlm = model(batches[i % len(batches)])
    n = scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()
    if i % 50 == 49 or i == args.steps - 1:
        log.append({"step": i + 1, "losses": [round(float(x), 4) for x in (mim, res, mlm)], "norm": round(float(n), 4), "scale": scaler.get_scale(), "skipped": scaler.skipped_steps})
        print(json.dumps(log[-1]), flush=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
print(json.dumps({"model": args.model, "dtype": args.dtype, "pairs_per_gpu": args.batch, "steps": args.steps, "ms_per_step_incl_readbacks": round(1e3 * dt / args.steps, 2),
                  "steps_taken": opt.steps_taken, "skipped": scaler.skipped_steps, "final_scale": scaler.get_scale(), "all_parameters_finite": finite}), flush=True)
