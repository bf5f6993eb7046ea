"""not gpu: the oracle (oracle/ecamp_oracle.py) against the golden vectors captured from the REFERENCE's own code
(tests/golden/*.npz, written by oracle/make_golden.py in the authoring container).  If /root/reference is present the
generator itself re-checks oracle-vs-reference live; here only committed data is needed."""
import os

import numpy as np
import pytest
import torch

from oracle import ecamp_oracle as orc
from oracle import recipe
from oracle.make_golden import GRAD_SAMPLE_KEYS, digest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("name,mk", [("tiny_b4_s128", orc.cfg_tiny), ("base_b2_s128", orc.cfg_base)])
def test_oracle_reproduces_reference_vectors(name, mk):
    torch.set_num_threads(8)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = mk()
    B, S = int(g["meta/B"]), int(g["meta/S"])
    P = orc.set_requires_grad(orc.load_state(orc.new_params(cfg), recipe.recipe_state(cfg, seed=0)), cfg)
    (mim, res, mlm), aux = orc.forward(P, cfg, recipe.recipe_batch(cfg, B, S, seed=0), 0.75, recipe.recipe_noise(B, cfg.num_patches, seed=0),
                                       return_aux=True)
    assert rel([mim.item(), res.item(), mlm.item()], g["losses"]) < 1e-6
    assert (aux["ids_keep"].numpy() == g["ids_keep"]).all() and (aux["ids_restore"].numpy() == g["ids_restore"]).all()
    assert (aux["mask"].numpy() == g["mask"]).all()
    for k in ("imgs", "latent", "pred", "pred_img", "sr", "fused", "seq_out", "logits"):
        nm, s = digest(aux[k])
        assert rel(nm[0], g["act/%s/nm" % k][0]) < 1e-5 and rel(s, g["act/%s/s" % k]) < 1e-5, k
    (mim + res + mlm).backward()
    names = list(g["grad/names"])
    norms = np.array([P[n].grad.double().norm().item() for n in names])
    assert (np.abs(norms - g["grad/norms"]) / (g["grad/norms"] + 1e-9)).max() < 1e-4
    for n in GRAD_SAMPLE_KEYS:
        assert rel(digest(P[n].grad)[1], g["grad/%s/s" % n]) < 1e-4, n
    for n in orc.UNUSED:
        assert P[n].grad is None  # the pooler never reaches a loss (bert_modeling.py:144)
    assert rel(float(orc.grad_norm([P[n].grad for n in names])), float(g["grad/global_norm"])) < 1e-5


def test_oracle_host_arithmetic_matches_reference():
    g = np.load(os.path.join(GOLD, "tiny_b4_s128.npz"))
    mine = [orc.adjust_learning_rate(float(e), 1.5e-4, 0.0, 40, 200) for e in g["lr/epochs"]]
    assert np.allclose(mine, g["lr/values"], rtol=1e-14, atol=0)
    nd, dc = orc.weight_decay_groups(orc.cfg_tiny())
    assert sorted(nd) == sorted(g["wd/no_decay"]) and sorted(dc) == sorted(g["wd/decay"])
    for dim, key in ((192, "pos_embed"), (512, "decoder_pos_embed")):
        nm, s = digest(orc.sincos_2d(dim, 14))
        assert rel(s, g["tab/%s/s" % key]) < 1e-7
    assert float(g["meta/oracle_vs_reference_worst_rel"]) < 2e-5  # recorded when the vectors were generated


def test_golden_generator_skips_cleanly_without_reference():
    from oracle import ref_shim
    if ref_shim.reference_available():
        pytest.skip("reference checkout present (authoring container)")
    with pytest.raises(RuntimeError):
        ref_shim.install()


def test_oracle_visualization_forward_matches_reference_vectors():
    """oracle.forward_visualization vs the cross-attention probabilities captured from the reference's
    Visualization model (tests/golden/vis_base_b2_s128.npz, written by oracle/make_golden_vis.py)."""
    torch.set_num_threads(8)
    g = np.load(os.path.join(GOLD, "vis_base_b2_s128.npz"))
    cfg = orc.cfg_base()
    B, S = int(g["meta/B"]), int(g["meta/S"])
    P = orc.load_state(orc.new_params(cfg, requires_grad=False), recipe.recipe_state(cfg, seed=0))
    batch = recipe.recipe_batch(cfg, B, S, seed=0)
    imgs = orc.bicubic_resize(batch["image"], cfg.img_size)
    with torch.no_grad():
        probs, ids_keep = orc.forward_visualization(P, cfg, imgs, batch["ids"], batch["attention_mask"], batch["type_ids"], 0.0,
                                                    recipe.recipe_noise(B, cfg.num_patches, seed=0))
    assert tuple(probs.shape) == (B, cfg.bert.num_attention_heads, S, cfg.num_patches)
    assert (ids_keep.numpy() == g["ids_keep"]).all()
    nm, s = digest(probs)
    assert rel(nm[0], g["probs/nm"][0]) < 1e-5 and rel(s, g["probs/s"]) < 1e-5
    assert rel(probs[:, :, 4].numpy(), g["probs_tok4"]) < 1e-5  # the row main_visualization.py:153-154 plots
    assert float((probs.sum(-1) - 1).abs().max()) < 1e-5
