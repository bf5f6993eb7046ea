#!/usr/bin/env python3
"""Random (batch, report length, mask ratio, dropout on / off) through the whole tiny model in the three activation formats against the oracle
on the host (oracle = test infrastructure; this tool is one): losses and every parameter's gradient.  python tools/fuzz_model.py [--cases 12]"""
import argparse, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ecamp_amd import hip_ops, _lib
from ecamp_amd.module import model_ecamp as me
from oracle import ecamp_oracle as orc
from oracle import recipe
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=12); ap.add_argument("--seed", type=int, default=0); args = ap.parse_args()
dev = torch.device("cuda:0")
torch.set_num_threads(16)
rng = random.Random(args.seed)
cfg = orc.cfg_tiny()
state = recipe.recipe_state(cfg, seed=0)
TOL = {torch.float32: (2e-4, 1e-3, 1e-3), torch.bfloat16: (3e-2, 1.5e-2, 6e-2), torch.float16: (1e-3, 3e-3, 1e-2)}   # losses, median / worst gradient
fails = 0
for c in range(args.cases):
    B, S = rng.randint(1, 6), rng.choice([rng.randint(4, 256), rng.choice([8, 64, 128, 256])])
    mr = rng.choice([0.25, 0.5, 0.75, 0.9]); train = rng.random() < 0.5
    batch = recipe.recipe_batch(cfg, B, S, seed=100 + c)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=100 + c)
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        model = me.ecamp_tiny(compute_dtype=dtype); model.load_state_dict(state); model.to(dev)
        model.train() if train else model.eval()
        model.prepare(); model._rng_trace = []
        out = model(batch, mask_ratio=mr, noise=noise)
        trace = list(model._rng_trace); model._rng_trace = None
        ls = 65536.0 if dtype == torch.float16 else 1.0
        (sum(out) * ls).backward(); torch.cuda.synchronize()
        it = iter(trace)
        def replay(shape, p):
            seed, off = next(it)
            return hip_ops.dropout_mask(shape, dev, p, seed, off).float().cpu()
        P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), state), cfg)
        ref = orc.forward(P, cfg, batch, mr, noise, train=replay if train else False)
        sum(ref).backward()
        lt, mt, wt = TOL[dtype]
        le = max(abs(a.item() - b.item()) / abs(b.item()) for a, b in zip(out, ref))
        gmax = max(t.grad.norm().item() for t in P.values() if t.grad is not None)
        errs = {}
        for n, prm in model.named_parameters():
            if not prm.requires_grad or P[n].grad is None: continue
            gr = P[n].grad
            if dtype != torch.float32 and gr.norm().item() < 1e-3 * gmax: continue
            errs[n] = (prm.grad.float().cpu() / ls - gr).norm().item() / (gr.norm().item() + 1e-5 * gmax)
        worst = max(errs, key=errs.get); med = float(np.median(list(errs.values())))
        ok = le < lt and med < mt and errs[worst] < wt
        fails += 0 if ok else 1
        print("%s B=%d S=%3d mask_ratio=%.2f %-5s %-8s losses %.1e  gradients median %.1e worst %.1e (%s)" % ("ok  " if ok else "FAIL", B, S, mr, "train" if train else "eval",
              str(dtype).split(".")[-1], le, med, errs[worst], worst), flush=True)
        del model
_lib.set_half("bf16")
print("fuzz_model: %d cases x 3 formats, %d failures" % (args.cases, fails), flush=True)
sys.exit(1 if fails else 0)
