#!/usr/bin/env python3
"""Loss / gradient-norm trajectories of the same training run (same weights, batch, masks and dropout streams) in the three activation
formats: f32 (parity mode = the oracle's arithmetic), bfloat16 (the benchmarked mode) and IEEE half with the reference's dynamic loss
scaling (`--amp fp16`, libecamp_hip_f16.so).  Says how far each 16-bit format walks from the f32 run, step by step.
    python tools/dtype_trajectory.py [--steps 12] [--batch 64]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=12); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--modes", default="f32,bf16,f16,f16-noscale"); args = ap.parse_args()
dev = torch.device("cuda:0")
runs = {}
for mode in args.modes.split(","):
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f16-noscale": torch.float16}[mode]
    torch.manual_seed(0)
    model = model_ecamp.ecamp(compute_dtype=dt).to(dev); model.prepare(); model.train()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount(dynamic=(mode == "f16"))
    batch = synthetic_batch(args.batch, 128, 448, seed=0, device=dev)
    rows = []
    for i in range(args.steps):
        mim, res, mlm = model(batch)
        n = scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        rows.append([float(mim.detach()), float(res.detach()), float(mlm.detach()), float(n)])
    runs[mode] = rows
    print(json.dumps({"mode": mode, "scale": scaler.get_scale(), "skipped": scaler.skipped_steps,
                      "rows_mim_res_mlm_norm": [[round(v, 4) for v in r] for r in rows]}), flush=True)
    del model, opt, batch; torch.cuda.empty_cache()
if "f32" in runs:
    for mode, rows in runs.items():
        if mode == "f32":
            continue
        dev_ = [max(abs(a - b) / abs(b) for a, b in zip(r, q)) for r, q in zip(rows, runs["f32"])]
        print(json.dumps({"mode": mode, "max_rel_deviation_from_f32_per_step": [round(d, 5) for d in dev_]}), flush=True)
