#!/usr/bin/env python3
"""Loss drift of the saved-derivative GELU (ecamp_gemm act = 2: the forward saves gelu'(x) in bf16, the backward multiplies by it) against
the recomputed derivative (act = 1) from identical initial weights, data and RNG streams -- and, as the yardstick for what a difference
of that size means, act = 1 against act = 1 with a different dropout seed is NOT used: the two runs here differ ONLY in that rounding.
   python tools/gelu_drift.py [--steps 200] [--batch 256] [--out profiles/r04_gelu_saved_grad_drift.json]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import optim
from ecamp_amd.data import synthetic_batch
from ecamp_amd.module import model_ecamp
from ecamp_amd.util.misc import NativeScalerWithGradNormCount

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
batches = [synthetic_batch(args.batch, 128, 448, seed=s, device=dev) for s in range(4)]
hist, ms = {}, {}
state = None
for act in (1, 2):
    torch.manual_seed(42)
    model = model_ecamp.ecamp(compute_dtype=torch.bfloat16)
    if state is None:
        state = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(state)
    model.gelu_act = act
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
    ms[act] = 1e3 * (time.perf_counter() - t0) / args.steps
    hist[act] = torch.stack(losses).float().cpu()
    del model, opt
rel = (hist[2] - hist[1]).abs() / hist[1].abs()
for i in range(0, args.steps, max(1, args.steps // 10)):
    print("step %4d  act=1 %s  act=2 %s  rel %s" % (i, [round(v, 4) for v in hist[1][i].tolist()], [round(v, 4) for v in hist[2][i].tolist()],
                                                    ["%.2e" % v for v in rel[i].tolist()]))
tail = slice(max(0, args.steps - 20), args.steps)
res = {"config": "ViT-B/16 + reference BERT, B=%d, S=128, %d optimizer steps, 4 synthetic batches cycled, lr 1.5e-4, dropout on (same Philox streams); the runs differ only in ecamp_gemm act = 1 (gelu' recomputed from the bf16 pre-activation) vs act = 2 (gelu' saved in bf16 by the forward pass)" % (args.batch, args.steps),
       "losses": ["mim", "res", "mlm"],
       "rel_diff_max": [round(v, 6) for v in rel.max(0).values.tolist()],
       "rel_diff_mean_last20": [round(v, 6) for v in rel[tail].mean(0).tolist()],
       "final_act1": [round(v, 5) for v in hist[1][-1].tolist()], "final_act2": [round(v, 5) for v in hist[2][-1].tolist()],
       "ms_per_step_act1": round(ms[1], 2), "ms_per_step_act2": round(ms[2], 2)}
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
