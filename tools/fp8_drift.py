#!/usr/bin/env python3
"""BASELINE.json configs[4]: loss drift of the fp8-forward mode against bf16 from identical initial weights, data and RNG streams.
   python tools/fp8_drift.py [--steps 200] [--batch 512] [--out profiles/r01_fp8_drift.json]
Prints per-20-step losses of both runs and writes the summary (max / final |loss_fp8 - loss_bf16| / loss_bf16 per loss, ms/step)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
batches = [synthetic_batch(args.batch, 128, 448, seed=s, device=dev) for s in range(4)]
hist, ms = {}, {}
state = None
for fp8 in (False, True):
    torch.manual_seed(42)
    model = model_ecamp.ecamp(compute_dtype=torch.bfloat16, fp8_forward=fp8)
    if state is None:
        state = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(state)
    model.to(dev).train()
    opt = optim.FusedAdamW(optim.add_weight_decay(model, 0.05), lr=1.5e-4, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount()
    losses = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(args.steps):
        mim, res, mlm = model(batches[i % len(batches)])
        scaler(mim + res + mlm, opt, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        losses.append(torch.stack([mim.detach(), res.detach(), mlm.detach()]))
    torch.cuda.synchronize()
    ms[fp8] = 1e3 * (time.perf_counter() - t0) / args.steps
    hist[fp8] = torch.stack(losses).float().cpu()
    del model, opt
rel = (hist[True] - hist[False]).abs() / hist[False].abs()
for i in range(0, args.steps, max(1, args.steps // 10)):
    print("step %4d  bf16 %s  fp8 %s  rel %s" % (i, [round(v, 4) for v in hist[False][i].tolist()], [round(v, 4) for v in hist[True][i].tolist()],
                                                 ["%.2e" % v for v in rel[i].tolist()]))
tail = slice(max(0, args.steps - 20), args.steps)
res = {"config": "ViT-B/16 + reference BERT, B=%d, S=128, %d optimizer steps, 4 synthetic batches cycled, lr 1.5e-4, dropout on (same Philox streams)" % (args.batch, args.steps),
       "losses": ["mim", "res", "mlm"],
       "rel_drift_max": [round(v, 5) for v in rel.max(0).values.tolist()],
       "rel_drift_mean_last20": [round(v, 5) for v in rel[tail].mean(0).tolist()],
       "final_bf16": [round(v, 5) for v in hist[False][-1].tolist()], "final_fp8": [round(v, 5) for v in hist[True][-1].tolist()],
       "ms_per_step_bf16": round(ms[False], 2), "ms_per_step_fp8": round(ms[True], 2)}
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
