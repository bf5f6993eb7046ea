#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Golden vector for SURVEY.md 8(f) f4: the reference's *Visualization* model
(/root/reference/Visualization/module, forward(imgs, text_ids, attention_mask, type_ids, mask_ratio=0) -> fusion
cross-attention probabilities [B, 6, S, 196]) run here on CPU through oracle/ref_shim.py, compared with
oracle/ecamp_oracle.forward_visualization, and written to tests/golden/vis_base_b2_s128.npz.

    python oracle/make_golden_vis.py        # needs /root/reference (authoring container only)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ecamp_oracle as orc  # noqa: E402
from oracle import recipe, ref_shim  # noqa: E402
from oracle.make_golden import _PatchRand, digest, rel  # noqa: E402

VIS_ROOT = "/root/reference/Visualization"
OUT = os.path.join(ROOT, "tests", "golden", "vis_base_b2_s128.npz")


def build_vis_model():
    ref_shim.install()  # stubs + the Pre-training path; now swap the `module` package for the Visualization one
    for k in [k for k in sys.modules if k == "module" or k.startswith("module.") or k == "util" or k.startswith("util.")]:
        del sys.modules[k]
    sys.path.insert(0, VIS_ROOT)
    import module.bert_config as bc
    for k, v in dict(is_decoder=False, add_cross_attention=False, chunk_size_feed_forward=0, tie_word_embeddings=True,
                     output_attentions=False, output_hidden_states=False, use_return_dict=True, return_dict=True).items():
        if not hasattr(bc.BertConfig, k):
            setattr(bc.BertConfig, k, v)
    import module.model_ecamp as me
    assert me.__file__.startswith(VIS_ROOT), me.__file__
    model = me.ecamp(norm_pix_loss=True)
    pred = model.bert_encoder.model.cls.predictions
    pred.decoder.bias = pred.bias
    for m in model.modules():
        cfg = getattr(m, "config", None)
        if cfg is not None and hasattr(cfg, "_attn_implementation"):
            try:
                cfg._attn_implementation = "eager"
            except Exception:
                pass
    return model


def main():
    torch.set_num_threads(8)
    cfg = orc.cfg_base()
    B, S = 2, 128
    model = build_vis_model()
    state = recipe.recipe_state(cfg, seed=0)
    model.load_state_dict(state, strict=True)
    model.eval()
    batch = recipe.recipe_batch(cfg, B, S, seed=0)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=0)
    imgs = orc.bicubic_resize(batch["image"], cfg.img_size)  # the Visualization model takes the 224^2 image directly
    with torch.no_grad(), _PatchRand(noise):
        ref = model(imgs, batch["ids"], batch["attention_mask"], batch["type_ids"])
    assert tuple(ref.shape) == (B, cfg.bert.num_attention_heads, S, cfg.num_patches), ref.shape
    P = orc.load_state(orc.new_params(cfg, requires_grad=False), state)
    with torch.no_grad():
        mine, ids_keep = orc.forward_visualization(P, cfg, imgs, batch["ids"], batch["attention_mask"], batch["type_ids"], 0.0, noise)
    nm_r, s_r = digest(ref)
    nm_o, s_o = digest(mine)
    err = max(rel(nm_o[0], nm_r[0]), rel(s_o, s_r), rel(mine[:, :, 4].numpy(), ref[:, :, 4].numpy()))
    print("oracle vs reference Visualization forward: max rel err %.3e" % err)
    assert err < 1e-5, err
    np.savez_compressed(OUT, **{"meta/B": np.array(B), "meta/S": np.array(S), "probs/nm": nm_r, "probs/s": s_r,
                                "probs_tok4": ref[:, :, 4].numpy().astype(np.float64), "ids_keep": ids_keep.numpy().astype(np.int32),
                                "rowsum_err": np.array(float((ref.sum(-1) - 1).abs().max()))})
    print("wrote", OUT)


if __name__ == "__main__":
    main()
