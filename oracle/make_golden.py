"""TEST INFRASTRUCTURE ONLY.  Generates `tests/golden/*.npz` by running the REFERENCE's own code
(`/root/reference/ECAMP/Pre-training`, imported through `oracle/ref_shim.py`) on recipe inputs,
and checks `oracle/ecamp_oracle.py` against it while doing so (the oracle is "pinned" only if this
script exits 0).  Runs in the authoring container only; the GPU box has no `/root/reference`.

    python -m oracle.make_golden            # all configs
    python -m oracle.make_golden tiny       # one config

What is recorded per config (all float64/float32 numpy, < 1 MB each):
  losses (mim, res, mlm); masking ints (ids_keep, ids_restore, mask); digests (L2 norm, mean,
  strided sample) of imgs, latent, pred, pred_img, sr, fused, seq_out, logits; per-parameter gradient
  L2 norms for EVERY trainable tensor + strided samples for a named subset; global grad-norm
  (misc.get_grad_norm_); for `tiny`: an accum_iter=2 AdamW engine step (post-step parameter
  digests) and the lr-schedule table from util/lr_sched.py with run.sh's arguments.
"""
import os
import sys
import time
import types

import numpy as np
import torch

from . import ecamp_oracle as orc
from . import recipe
from . import ref_shim

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CONFIGS = {
    # name: (cfg factory, tiny flag for the shim, B, S)
    "tiny_b4_s128": (orc.cfg_tiny, True, 4, 128),
    "base_b2_s128": (orc.cfg_base, False, 2, 128),
    "base_b2_s256": (orc.cfg_base, False, 2, 256),
}
GRAD_SAMPLE_KEYS = [
    "cls_token", "mask_token", "patch_embed.proj.weight", "patch_embed.proj.bias",
    "blocks.0.attn.qkv.weight", "blocks.0.norm1.weight", "blocks.11.mlp.fc2.weight", "norm.bias",
    "decoder_embed.weight", "decoder_blocks.3.attn.proj.weight", "decoder_pred.bias",
    "super_res.conv1.weight", "super_res.conv2.bias", "bert_mlp.weight",
    "bert_encoder.model.bert.embeddings.word_embeddings.weight",
    "bert_encoder.model.bert.embeddings.position_embeddings.weight",
    "bert_encoder.model.bert.embeddings.LayerNorm.weight",
    "bert_encoder.model.bert.context_fusion_layer.cross_self_attention.key.weight",
    "bert_encoder.model.bert.context_fusion_layer.gap_mlp.weight",
    "bert_encoder.model.bert.context_fusion_layer.out_layer.LayerNorm.bias",
    "bert_encoder.model.bert.encoder.layer.1.attention.self.query.weight",
    "bert_encoder.model.bert.encoder.layer.0.output.dense.bias",
    "bert_encoder.model.cls.predictions.bias", "bert_encoder.model.cls.predictions.decoder.weight",
    "bert_encoder.model.cls.predictions.transform.LayerNorm.weight",
]
NSAMP = 1024


def digest(t):
    """(norm, mean, strided sample) -- the same function is used by the parity tests."""
    f = t.detach().to(torch.float64).flatten()
    step = max(1, f.numel() // NSAMP)
    return np.array([f.norm().item(), f.mean().item()]), f[::step][:NSAMP].numpy().astype(np.float64)


def put(d, name, t):
    d[name + "/nm"], d[name + "/s"] = digest(t)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


class _PatchRand:
    """Replace torch.rand by the recipe noise for the duration of one reference forward."""

    def __init__(self, noise):
        self.noise = noise

    def __enter__(self):
        self._orig = torch.rand
        torch.rand = lambda *a, **k: self.noise.clone()

    def __exit__(self, *a):
        torch.rand = self._orig


def run_reference(name, cfg, tiny, B, S):
    torch.manual_seed(0)
    model = ref_shim.build_reference_model(tiny=tiny)
    state = recipe.recipe_state(cfg, seed=0)
    ref_keys = list(model.state_dict().keys())
    assert sorted(ref_keys) == sorted(state.keys()), (set(ref_keys) ^ set(state.keys()))
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(state[k].shape), k
    model.load_state_dict(state, strict=True)
    # the sin-cos tables the reference computed itself must equal the oracle's (a22)
    model.eval()
    batch = recipe.recipe_batch(cfg, B, S, seed=0)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=0)

    cap = {}
    bert = model.bert_encoder.model
    h1 = bert.bert.context_fusion_layer.register_forward_hook(lambda m, i, o: cap.__setitem__("fused", o[0]))
    h2 = bert.cls.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o))
    h3 = bert.bert.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("seq_out", o[0]))
    h4 = model.super_res.register_forward_hook(lambda m, i, o: cap.__setitem__("sr", o))
    orig_enc = model.image_encoder
    orig_dec = model.image_decoder

    def enc(x, r):
        out = orig_enc(x, r)
        cap["imgs"], cap["latent"], cap["mask"], cap["ids_restore"], cap["ids_keep"] = x, out[0], out[1], out[2], out[3]
        return out

    def dec(x, r):
        out = orig_dec(x, r)
        cap["pred"] = out
        return out

    model.image_encoder, model.image_decoder = enc, dec
    t0 = time.time()
    with _PatchRand(noise):
        mim, res, mlm = model(batch)
    (mim + res + mlm).backward()
    print("  reference fwd+bwd %.1fs  losses %.6f %.6f %.6f" % (time.time() - t0, mim.item(), res.item(), mlm.item()))
    for h in (h1, h2, h3, h4):
        h.remove()
    model.image_encoder, model.image_decoder = orig_enc, orig_dec
    cap["pred_img"] = model.unpatchify(cap["pred"])
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    import util.misc as misc
    gn = misc.get_grad_norm_([p for p in model.parameters()])
    return model, batch, noise, (mim, res, mlm), cap, grads, gn


def run_oracle(cfg, batch, noise):
    P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), recipe.recipe_state(cfg, seed=0)), cfg)
    t0 = time.time()
    (mim, res, mlm), aux = orc.forward(P, cfg, batch, 0.75, noise, train=False, return_aux=True)
    (mim + res + mlm).backward()
    print("  oracle    fwd+bwd %.1fs  losses %.6f %.6f %.6f" % (time.time() - t0, mim.item(), res.item(), mlm.item()))
    grads = {k: P[k].grad for k in orc.trainable_names(cfg)}
    return P, (mim, res, mlm), aux, grads


def engine_step_reference(model, cfg, B, S):
    """a1-a4 golden: accum_iter=2, AdamW(lr, betas=(0.9,0.95)), wd groups from timm add_weight_decay."""
    import timm.optim.optim_factory as of
    import util.misc as misc
    model.zero_grad(set_to_none=True)
    groups = of.add_weight_decay(model, 0.05)
    opt = torch.optim.AdamW(groups, lr=1.5e-4, betas=(0.9, 0.95))
    logged = []
    for it in range(2):
        batch = recipe.recipe_batch(cfg, B, S, seed=10 + it)
        noise = recipe.recipe_noise(B, cfg.num_patches, seed=10 + it)
        with _PatchRand(noise):
            mim, res, mlm = model(batch)
        logged.append([mim.item(), res.item(), mlm.item()])
        ((mim + res + mlm) / 2).backward()
    norm = misc.get_grad_norm_(model.parameters())
    opt.step()
    return np.array(logged), float(norm), {n: p.detach() for n, p in model.named_parameters()}


def engine_step_oracle(cfg, B, S):
    P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), recipe.recipe_state(cfg, seed=0)), cfg)
    logged = []
    for it in range(2):
        batch = recipe.recipe_batch(cfg, B, S, seed=10 + it)
        noise = recipe.recipe_noise(B, cfg.num_patches, seed=10 + it)
        mim, res, mlm = orc.forward(P, cfg, batch, 0.75, noise)
        logged.append([mim.item(), res.item(), mlm.item()])
        ((mim + res + mlm) / 2).backward()
    names = [k for k in orc.trainable_names(cfg) if P[k].grad is not None]
    norm = orc.grad_norm([P[k].grad for k in names])
    no_decay, decay = orc.weight_decay_groups(cfg)
    with torch.no_grad():
        for k in names:
            orc.adamw_step(P[k], P[k].grad, torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1, 1.5e-4,
                           0.0 if k in set(no_decay) else 0.05)
    return np.array(logged), float(norm), P


def lr_table():
    install_args = types.SimpleNamespace(lr=1.5e-4, min_lr=0.0, warmup_epochs=40, max_epoch=200)
    import util.lr_sched as lr_sched
    grp = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]
    opt = types.SimpleNamespace(param_groups=grp)
    eps = np.array([0.0, 0.5, 1.0, 7.25, 39.99, 40.0, 40.5, 60.0, 100.0, 119.999, 120.0, 150.0, 199.0, 200.0])
    ref = np.array([lr_sched.adjust_learning_rate(opt, float(e), install_args) for e in eps])
    mine = np.array([orc.adjust_learning_rate(float(e), 1.5e-4, 0.0, 40, 200) for e in eps])
    assert np.allclose(ref, mine, rtol=1e-14, atol=0), (ref, mine)
    assert grp[1]["lr"] == ref[-1] * 0.5
    return eps, ref


def main(which):
    ref_shim.install()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    worst = 0.0
    for name, (mk, tiny, B, S) in CONFIGS.items():
        if which and name not in which and name.split("_")[0] not in which:
            continue
        cfg = mk()
        print("[%s] B=%d S=%d" % (name, B, S))
        model, batch, noise, rl, cap, rg, rgn = run_reference(name, cfg, tiny, B, S)
        P, ol, aux, og = run_oracle(cfg, batch, noise)

        d = {"meta/B": np.array(B), "meta/S": np.array(S)}
        d["losses"] = np.array([x.item() for x in rl], dtype=np.float64)
        errs = {"losses": rel([x.item() for x in ol], d["losses"])}
        for k in ("ids_keep", "ids_restore"):
            d[k] = cap[k].numpy().astype(np.int32)
            assert (aux[k].numpy() == cap[k].numpy()).all(), k
        d["mask"] = cap["mask"].numpy().astype(np.float32)
        assert (aux["mask"].numpy() == d["mask"]).all()
        for k in ("imgs", "latent", "pred", "pred_img", "sr", "fused", "seq_out", "logits"):
            put(d, "act/" + k, cap[k])
            nm, s = digest(aux[k])
            errs["act/" + k] = max(rel(nm[0], d["act/" + k + "/nm"][0]), rel(s, d["act/" + k + "/s"]))
        # sin-cos tables computed by the reference's own util/pos_embed.py
        sd = model.state_dict()
        errs["pos_embed"] = rel(orc.sincos_2d(cfg.embed_dim, cfg.grid), sd["pos_embed"])
        errs["decoder_pos_embed"] = rel(orc.sincos_2d(cfg.decoder_embed_dim, cfg.grid), sd["decoder_pos_embed"])
        put(d, "tab/pos_embed", sd["pos_embed"])
        put(d, "tab/decoder_pos_embed", sd["decoder_pos_embed"])
        # gradients
        names = orc.trainable_names(cfg)
        assert sorted(rg.keys()) == sorted(names), set(rg.keys()) ^ set(names)
        none_ref = sorted(n for n in names if rg[n] is None)
        assert none_ref == sorted(orc.UNUSED), none_ref
        assert sorted(n for n in names if og[n] is None) == none_ref
        gnames = [n for n in names if rg[n] is not None]
        d["grad/names"] = np.array(gnames)
        d["grad/norms"] = np.array([rg[n].double().norm().item() for n in gnames])
        onorms = np.array([og[n].double().norm().item() for n in gnames])
        errs["grad/norms"] = float(np.abs(onorms - d["grad/norms"]).max() / d["grad/norms"].max())
        errs["grad/norms_each"] = float((np.abs(onorms - d["grad/norms"]) / (d["grad/norms"] + 1e-12)).max())
        for n in GRAD_SAMPLE_KEYS:
            put(d, "grad/" + n, rg[n])
            nm, s = digest(og[n])
            errs["grad/" + n] = rel(s, d["grad/" + n + "/s"])
        d["grad/global_norm"] = np.array(float(rgn))
        errs["grad/global_norm"] = rel(float(orc.grad_norm([og[n] for n in gnames])), float(rgn))
        # weight-decay group membership (a4)
        import timm.optim.optim_factory as of
        groups = of.add_weight_decay(model, 0.05)
        id2n = {id(p): n for n, p in model.named_parameters()}
        nd_ref = sorted(id2n[id(p)] for p in groups[0]["params"])
        dc_ref = sorted(id2n[id(p)] for p in groups[1]["params"])
        nd, dc = orc.weight_decay_groups(cfg)
        assert sorted(nd) == nd_ref and sorted(dc) == dc_ref
        d["wd/no_decay"], d["wd/decay"] = np.array(nd_ref), np.array(dc_ref)

        if tiny:
            lg, norm, newp = engine_step_reference(model, cfg, B, S)
            olg, onorm, oP = engine_step_oracle(cfg, B, S)
            d["engine/logged"], d["engine/grad_norm"] = lg, np.array(norm)
            errs["engine/logged"] = rel(olg, lg)
            errs["engine/grad_norm"] = rel(onorm, norm)
            for n in GRAD_SAMPLE_KEYS:
                put(d, "engine/param/" + n, newp[n])
                # compare the UPDATE (p_new - p_old), the part AdamW computes
                old = recipe.recipe_state(cfg, seed=0)[n].double()
                du_ref = (newp[n].double() - old).flatten()
                du_orc = (oP[n].detach().double() - old).flatten()
                errs["engine/param/" + n] = rel(du_orc.numpy(), du_ref.numpy())
            eps, lrs = lr_table()
            d["lr/epochs"], d["lr/values"] = eps, lrs

        bad = {k: v for k, v in errs.items() if v > 2e-5}
        for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:6]:
            print("    oracle-vs-reference rel err %-70s %.2e" % (k, v))
        worst = max(worst, max(errs.values()))
        if bad:
            print("ORACLE MISMATCH:", bad)
            sys.exit(1)
        d["meta/oracle_vs_reference_worst_rel"] = np.array(max(errs.values()))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print("  wrote %s (%d arrays)" % (name + ".npz", len(d)))
        del model, P
    print("oracle pinned against the reference; worst rel err %.2e" % worst)


if __name__ == "__main__":
    main(sys.argv[1:])
