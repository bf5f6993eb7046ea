#!/usr/bin/env python3
"""Random model GEOMETRY (image size, patch size, widths, head dimensions, decoder, MLP ratio, BERT depth) through the whole model in the three
activation formats against the oracle built for the same configuration: the image-side kernels (im2col of visible patches, unshuffle, unpatchify,
the SR head's windows, bicubic) and the attention kernels see token counts and widths no named configuration has.  A geometry the library refuses
with an error counts as refused, a silent mismatch as a failure.   python tools/fuzz_geometry.py [--cases 8] [--seed 0]"""
import argparse, os, random, sys
from functools import partial
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn
from ecamp_amd import _lib
from ecamp_amd.module import model_ecamp as me
from ecamp_amd.module.bert_config import BertConfig
from oracle import ecamp_oracle as orc
from oracle import recipe
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=8); ap.add_argument("--seed", type=int, default=0); args = ap.parse_args()
dev = torch.device("cuda:0")
torch.set_num_threads(32)
rng = random.Random(args.seed)
TOL = {torch.float32: (2e-4, 1e-3, 2e-3), torch.bfloat16: (3e-2, 1.5e-2, 6e-2), torch.float16: (1e-3, 3e-3, 1e-2)}
fails = refused = 0
for c in range(args.cases):
    img, patch = rng.choice([(224, 16), (256, 16), (336, 16), (448, 16), (288, 16), (400, 16)])
    hd, H = rng.choice([32, 64, 128]), rng.randint(1, 6)
    hdd, Hd = rng.choice([32, 64, 128]), rng.randint(1, 6)
    depth, ddepth, mlp, bl = rng.randint(1, 2), rng.randint(1, 2), rng.choice([2.0, 4.0]), rng.randint(1, 2)
    grid = img // patch
    # the report side: hidden width / heads / intermediate width / vocabulary other than the reference's 768 / 6 / 1536 / 30000 half of the time
    bh, bH = rng.choice([(768, 6), (768, 6), (384, 6), (512, 4), (256, 8), (640, 5), (1024, 8)])
    binter, vocab = rng.choice([2, 4]) * bh if rng.random() < 0.5 else 1536, rng.choice([30000, 30000, 1000, 8000, 30528])
    cfg = orc.Cfg(img_size=img, patch_size=patch, embed_dim=hd * H, depth=depth, num_heads=H, decoder_embed_dim=hdd * Hd, decoder_depth=ddepth,
                  decoder_num_heads=Hd, mlp_ratio=mlp, sr_window=(12 * grid) // 14,
                  bert=orc.BertCfg(num_hidden_layers=bl, hidden_size=bh, num_attention_heads=bH, intermediate_size=binter, vocab_size=vocab))
    B, S = rng.randint(1, 4), rng.randint(8, 160)
    mr = rng.choice([0.5, 0.75])
    state = recipe.recipe_state(cfg, seed=c)
    batch = recipe.recipe_batch(cfg, B, S, seed=200 + c)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=200 + c)
    tag = "img %d patch %d enc %dx%d (hd %d) depth %d dec %dx%d (hd %d) depth %d mlp %.0f bert %d x %d/%d/%d vocab %d B %d S %d mask %.2f" % (img, patch, H, hd, hd, depth, Hd, hdd, hdd, ddepth, mlp, bl, bh, bH, binter, vocab, B, S, mr)
    P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), state), cfg)
    ref = orc.forward(P, cfg, batch, mr, noise)
    sum(ref).backward()
    gmax = max(t.grad.norm().item() for t in P.values() if t.grad is not None)
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        try:
            model = me.ECAMP(img_size=img, patch_size=patch, in_chans=3, embed_dim=hd * H, depth=depth, num_heads=H, decoder_embed_dim=hdd * Hd, decoder_depth=ddepth,
                             decoder_num_heads=Hd, mlp_ratio=mlp, norm_layer=partial(nn.LayerNorm, eps=1e-6), bert_config=BertConfig(num_hidden_layers=bl, hidden_size=bh, num_attention_heads=bH, intermediate_size=binter, vocab_size=vocab), compute_dtype=dtype)
            model.load_state_dict(state, strict=True); model.to(dev).eval()
            out = model(batch, mask_ratio=mr, noise=noise)
            ls = 65536.0 if dtype == torch.float16 else 1.0
            (sum(out) * ls).backward(); torch.cuda.synchronize()
        except (_lib.EcampHipError, ValueError) as e:
            refused += 1; print("refused  %s %s: %s" % (tag, str(dtype).split(".")[-1], str(e)[:140]), flush=True); continue
        lt, mt, wt = TOL[dtype]
        le = max(abs(a.item() - b.item()) / abs(b.item()) for a, b in zip(out, ref))
        errs = {}
        for n, prm in model.named_parameters():
            if not prm.requires_grad or P[n].grad is None: continue
            gr = P[n].grad
            if dtype != torch.float32 and gr.norm().item() < 1e-3 * gmax: continue
            errs[n] = (prm.grad.float().cpu() / ls - gr).norm().item() / (gr.norm().item() + 1e-5 * gmax)
        worst = max(errs, key=errs.get); med = float(np.median(list(errs.values())))
        ok = le < lt and med < mt and errs[worst] < wt
        fails += 0 if ok else 1
        print("%s %s %-8s losses %.1e gradients median %.1e worst %.1e (%s)" % ("ok  " if ok else "FAIL", tag, str(dtype).split(".")[-1], le, med, errs[worst], worst), flush=True)
        del model
_lib.set_half("bf16")
print("fuzz_geometry: %d geometries x 3 formats, %d refused with an error, %d failures" % (args.cases, refused, fails), flush=True)
sys.exit(1 if fails else 0)
