#!/usr/bin/env python3
"""The full-size model with the production kernel selection at batch sizes / report lengths that are not multiples of anything (edge tiles of the
persistent GEMMs, ragged item tables of the grouped weight gradients), bf16 and fp16, against the oracle on the host.  python tools/odd_batch.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from ecamp_amd import _lib
from ecamp_amd.module import model_ecamp as me
from oracle import ecamp_oracle as orc
from oracle import recipe
dev = torch.device("cuda:0")
torch.set_num_threads(min(os.cpu_count() or 1, 64))
cfg = orc.cfg_base()
state = recipe.recipe_state(cfg, seed=0)
for B, S in ((100, 128), (37, 100), (130, 77)):
    batch = recipe.recipe_batch(cfg, B, S, seed=31); noise = recipe.recipe_noise(B, cfg.num_patches, seed=31)
    runs = {}
    for dtype, scale in ((torch.bfloat16, 1.0), (torch.float16, 65536.0)):
        _lib.set_half(dtype); lib = _lib.load()
        model = me.ecamp(compute_dtype=dtype); model.load_state_dict(state, strict=True); model.to(dev).eval()
        q0, w0, s0 = int(lib.ecamp_gemm_q8_launches()), int(lib.ecamp_wgrad_group_launches()), int(lib.ecamp_gemm_q16_launches())
        out = model(batch, mask_ratio=0.75, noise=noise); (sum(out) * scale).backward(); torch.cuda.synchronize()
        params = dict(model.named_parameters())
        runs[dtype] = (np.array([t.item() for t in out]), {n: params[n].grad.double().norm().item() / scale for n in orc.trainable_names(cfg) if params[n].grad is not None},
                       int(lib.ecamp_gemm_q8_launches()) - q0, int(lib.ecamp_wgrad_group_launches()) - w0, int(lib.ecamp_gemm_q16_launches()) - s0)
        del model, out, params; torch.cuda.empty_cache()
    P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), state), cfg)
    ref = orc.forward(P, cfg, batch, 0.75, noise); sum(ref).backward()
    want = np.array([t.item() for t in ref])
    for dtype in runs:
        got, gn, nq, nw, n16 = runs[dtype]
        names = [n for n in gn if P[n].grad is not None]
        rn = np.array([P[n].grad.double().norm().item() for n in names]); hn = np.array([gn[n] for n in names])
        big = rn > 1e-3 * rn.max(); e = np.abs(hn - rn)[big] / rn[big]
        print("B=%d S=%d %s: q8 %d (q16 %d) grouped wgrad %d; loss rel %s; grad-norm median %.2e max %.2e (%s)" % (B, S, str(dtype).split('.')[-1], nq, n16, nw,
              np.round(np.abs(got - want) / want, 6), np.median(e), e.max(), np.array(names)[big][int(e.argmax())]), flush=True)
_lib.set_half("bf16")
