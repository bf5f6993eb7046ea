"""the batched entity-aware masker and loss-weight computation (ecamp_amd/module/pretrain_datasets.py) against
vectors captured from the REFERENCE's own ContextBertDataset code (tests/golden/data_pipeline.npz, oracle/make_golden_data.py).
Integer outputs must be bit-exact; the f32 weights too (same f64 arithmetic, same single rounding)."""
import os
import random

import numpy as np
import pytest
import torch

from ecamp_amd.module import pretrain_datasets as pdm

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz")
REF_TOK = "/root/reference/ECAMP/Pre-training/dataset/mimic_wordpiece.json"


def _gold():
    g = np.load(GOLD, allow_pickle=False)
    V = int(g["vocab_size"])
    vocab = pdm.MaskVocab(torch.from_numpy(np.unpackbits(g["is_entity"])[:V].astype(bool)),
                          torch.from_numpy(np.unpackbits(g["is_subword"])[:V].astype(bool)))
    return g, vocab


def _stream(rng, L):
    return torch.tensor([[rng.random() for _ in range(2 * L)]], dtype=torch.float64)


def _mask_cases(device):
    g, vocab = _gold()
    vocab = vocab.to(device)
    toks = torch.from_numpy(g["cm_tokens"]).to(device)
    L = toks.shape[1]
    streams = torch.cat([_stream(random.Random(int(s)), L) for s in g["cm_seed"]], 0).to(device)
    masked, mask_pos = pdm.context_mask(toks, vocab, streams)   # ONE batched call over all 64 cases
    assert (masked.cpu().numpy() == g["cm_masked"]).all()
    assert (mask_pos.cpu().numpy() == g["cm_mask_pos"]).all()
    # a case where nothing is valid (position 1 is already PAD) consumes nothing and changes nothing
    t = torch.zeros((1, L), dtype=torch.int64, device=device)
    t[0, 0] = 2
    m, p = pdm.context_mask(t, vocab, streams[:1])
    assert (m == t).all() and not p.any()


def _item_cases(device):
    g, vocab = _gold()
    vocab = vocab.to(device)
    ids = torch.from_numpy(g["item_ids"]).to(device)
    L = ids.shape[1]
    streams = []
    for seed, idx in zip(g["item_seed"], g["item_index"]):
        rng = random.Random(int(seed))
        pdm.assemble_report(str(g["item_reports"][idx]), str(g["item_llm"][idx]), rng)   # the draws __getitem__ makes before masking
        streams.append(_stream(rng, L))
    streams = torch.cat(streams, 0).to(device)
    masked, mask_pos = pdm.context_mask(ids, vocab, streams)
    weights = pdm.template_weights(ids, mask_pos)
    assert (masked.cpu().numpy() == g["item_masked"]).all()
    w, gw = weights.cpu().numpy(), g["item_weights"]
    assert w.dtype == np.float32 and np.array_equal(w, gw), float(np.abs(w - gw).max())
    # the fixtures exercise both re-normalisation branches and the template down-weighting
    assert (gw == np.float32(0.05)).any() and (gw > 1).any()


def test_context_mask_matches_reference_loop():
    _mask_cases("cpu")


def test_item_masking_and_weights_match_reference():
    _item_cases("cpu")


@pytest.mark.gpu
def test_masker_on_device_matches_reference(dev):
    _mask_cases(dev)
    _item_cases(dev)
    m = pdm.DeviceMasker(_gold()[1], dev, seed=3)
    ids = torch.from_numpy(_gold()[0]["item_ids"]).to(dev)
    masked, w = m(ids)
    assert masked.shape == ids.shape and w.shape == ids.shape and w.dtype == torch.float32
    assert ((masked == ids) | (masked == pdm.MASK)).all()


@pytest.mark.skipif(not os.path.exists(REF_TOK), reason="the reference tokenizer file is only present in the authoring container")
def test_text_item_with_tokenizer_matches_reference_items():
    """Whole text half of __getitem__ (LLM splice, tokenisation, masking, weights) from the report strings."""
    import tokenizers
    g, _ = _gold()
    ds = object.__new__(pdm.ContextBertDataset)
    ds.max_caption_length = g["item_ids"].shape[1]
    ds.tokenizer = tokenizers.Tokenizer.from_file(REF_TOK)
    ds.tokenizer.enable_truncation(max_length=ds.max_caption_length)
    ds.tokenizer.enable_padding(length=ds.max_caption_length)
    ds.vocab = pdm.MaskVocab.from_tokenizer(ds.tokenizer)
    ds.report_list, ds.llm_out_list = [str(s) for s in g["item_reports"]], [str(s) for s in g["item_llm"]]
    for k, (seed, idx) in enumerate(zip(g["item_seed"], g["item_index"])):
        ids, am, ty, masked, w = ds.text_item(int(idx), rng=random.Random(int(seed)))
        assert (ids[0].numpy() == g["item_ids"][k]).all() and (am[0].numpy() == g["item_attn"][k]).all()
        assert (masked[0].numpy() == g["item_masked"][k]).all() and np.array_equal(w[0].numpy(), g["item_weights"][k])
    batch = ds.collate_fn([(torch.zeros(3, 2, 2), ids, am, ty, masked, w, torch.tensor([1]), torch.tensor([2]))])
    assert batch["ids"].shape == (1, ds.max_caption_length) and batch["column"].shape == (1,)   # no .squeeze() at B == 1


def test_uint8_image_item_is_the_f32_item_before_normalisation():
    """ContextBertDataset(image_u8=True): the image half of an item stops before ToTensor / Normalize (pretrain_datasets.py:50-52) and
    returns the uint8 grayscale crop; normalising it reproduces the default f32 item bit for bit (same crop, same flip: the transform
    draws from the torch RNG exactly as before).  measure_item_rate reports a positive items/s figure."""
    import numpy as np
    import torch
    from PIL import Image
    from ecamp_amd.data import normalise_u8
    from ecamp_amd.module.pretrain_datasets import default_image_transform, measure_item_rate
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, (300, 260, 3), dtype=np.uint8), "RGB")
    torch.manual_seed(5)
    f = default_image_transform(64)(img)
    torch.manual_seed(5)
    u = default_image_transform(64, image_u8=True)(img)
    assert f.shape == (3, 64, 64) and f.dtype == torch.float32 and u.shape == (64, 64) and u.dtype == torch.uint8
    assert torch.equal(normalise_u8(u[None])[0], f)

    class Fake:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            return default_image_transform(64, image_u8=True)(img)

    import random
    import numpy as np
    random.seed(7); torch.manual_seed(7); np.random.seed(7)
    before = (random.getstate(), torch.get_rng_state().clone(), np.random.get_state()[1].copy())
    assert measure_item_rate(Fake(), seconds=0.2, max_items=16) > 0
    # the probe runs a machine-speed-dependent number of items; it must leave every generator where it found it
    assert random.getstate() == before[0] and torch.equal(torch.get_rng_state(), before[1]) and (np.random.get_state()[1] == before[2]).all()
