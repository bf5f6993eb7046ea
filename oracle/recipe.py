"""TEST INFRASTRUCTURE ONLY.  Deterministic, torch-RNG-independent recipes for weights, batches and
masking noise, so that the reference (in the authoring container), the oracle and the HIP path
(on the GPU box) can all build *identical* inputs without committing 183 M floats.

numpy `Generator(PCG64(seed))` streams are used throughout; seeds derive from the tensor's
state-dict key (crc32) so the recipe is independent of registration order.
"""
import zlib

import numpy as np
import torch

from . import ecamp_oracle as orc


def _rng(key, seed):
    return np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) + 7919 * seed) & 0xFFFFFFFF))


def recipe_state(cfg, seed=0):
    """Reference-style state dict (incl. the tied `decoder.bias` alias) with recipe weights.

    Scales are chosen so activations stay O(1) through 12+4+7 layers: linear weights
    ~N(0, 2/(fan_in+fan_out)), LN gamma 1+0.1n, LN beta / biases 0.02n, embeddings 0.02n,
    3x3 convs 0.2n; frozen sin-cos tables as the model computes them.
    """
    shapes = orc.param_shapes(cfg)
    st = {}
    for k, (shp, _tr) in shapes.items():
        if k in orc.TIED:
            continue
        if k == "pos_embed":
            st[k] = orc.sincos_2d(cfg.embed_dim, cfg.grid)
            continue
        if k == "decoder_pos_embed":
            st[k] = orc.sincos_2d(cfg.decoder_embed_dim, cfg.grid)
            continue
        g = _rng(k, seed)
        n = g.standard_normal(size=shp, dtype=np.float32)
        if k.startswith("super_res") and len(shp) == 4:
            w = 0.2 * n
        elif len(shp) == 4:  # patch-embed conv, viewed (out, in*p*p)
            fan_in = shp[1] * shp[2] * shp[3]
            w = n * np.float32(np.sqrt(2.0 / (fan_in + shp[0])))
        elif len(shp) == 2 and "embeddings" in k:
            w = 0.02 * n
        elif len(shp) == 2:
            w = n * np.float32(np.sqrt(2.0 / (shp[0] + shp[1])))
        elif len(shp) == 1 and k.endswith("weight"):  # LayerNorm gamma
            w = 1.0 + 0.1 * n
        else:  # biases, cls_token, mask_token
            w = 0.02 * n
        st[k] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    for alias, src in orc.TIED.items():
        st[alias] = st[src]
    return st


def recipe_batch(cfg, B, S, seed=0, big=None):
    """Synthetic batch with the schema of pretrain_datasets.py:228-237."""
    g = np.random.Generator(np.random.PCG64(1234 + seed))
    R = 2 * cfg.img_size if big is None else big
    image = g.standard_normal(size=(B, 3, R, R), dtype=np.float32)
    labels = g.integers(5, cfg.bert.vocab_size, size=(B, S), dtype=np.int64)
    lens = g.integers(max(2, S // 4), S + 1, size=(B,))
    am = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    labels = labels * am  # PAD = 0
    labels[:, 0] = 2  # CLS
    ids = labels.copy()
    mask_here = (g.random(size=(B, S)) < 0.5) & (am == 1)
    mask_here[:, 0] = False
    ids[mask_here] = 3  # MASK
    weights = np.ones((B, S), dtype=np.float32)
    dim = g.random(size=(B, S)) < 0.10
    weights[dim] = 0.05
    # row re-normalisation in the spirit of pretrain_datasets.py:177-184 (masked positions expanded)
    for b in range(B):
        dcnt = int(dim[b].sum())
        mcnt = int(mask_here[b].sum())
        ldm = int((dim[b] & mask_here[b]).sum())
        if mcnt > 0 and dcnt > 0:
            weights[b, mask_here[b]] *= np.float32((0.95 * (dcnt - ldm) + mcnt) / (mcnt - 0.95 * ldm))
    column = g.integers(0, 3, size=(B,), dtype=np.int64)
    row = g.integers(0, 3, size=(B,), dtype=np.int64)
    t = torch.from_numpy
    return dict(image=t(image), ids=t(ids), labels=t(labels), attention_mask=t(am),
                type_ids=torch.zeros(B, S, dtype=torch.int64), weights=t(weights), column=t(column), row=t(row))


def recipe_noise(B, L, seed=0):
    """Masking noise standing in for torch.rand(N, L) (model_ecamp.py:177); tie-free by construction."""
    g = np.random.Generator(np.random.PCG64(4242 + seed))
    n = np.stack([g.permutation(L) for _ in range(B)]).astype(np.float32)
    return torch.from_numpy((n + 0.5) / L)
