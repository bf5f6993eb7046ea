#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Golden vectors for SURVEY.md 8(f) f2: the reference's own `ContextBertDataset._context_mask`
and `__getitem__` (ECAMP/Pre-training/module/pretrain_datasets.py:60-199) run here through oracle/ref_shim.py on (a) synthetic
radiology-style reports written for this test and (b) random token sequences with hand-placed edge cases, with Python's `random`
seeded per case.  Writes tests/golden/data_pipeline.npz: inputs (token ids, the uniforms the reference consumed, the two
per-token-id flags the masker needs) and expected outputs (masked ids, context positions, loss weights).

    python oracle/make_golden_data.py        # needs /root/reference (authoring container only)
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "data_pipeline.npz")
L = 256

FINDINGS = ["there is no evidence of pneumothorax", "there is no pleural effusion", "small left pleural effusion is unchanged",
            "the cardiomediastinal silhouette is within normal limits", "mild cardiomegaly with pulmonary vascular congestion",
            "bibasilar atelectasis without focal consolidation", "there is no focal consolidation", "right lower lobe opacity concerning for pneumonia",
            "the lungs are hyperinflated consistent with emphysema", "a right internal jugular catheter terminates in the superior vena cava",
            "healed left rib fracture", "no acute osseous abnormality", "interstitial edema has improved", "tortuous thoracic aorta",
            "stable granuloma in the left upper lobe", "low lung volumes", "the patient is status post median sternotomy",
            "hilar contours are unremarkable", "there is no evidence of pulmonary embolism", "moderate hiatal hernia",
            "perihilar opacities may reflect mild pulmonary vascular engorgement", "no pneumothorax", "increased thickening of the minor fissure"]
LLM = ["pneumothorax absent; effusion small.", "cardiomegaly, congestion.", "", "opacity in the right lower lobe, possible pneumonia.",
       "no acute cardiopulmonary process.", "atelectasis; edema improved."]


def reports(n, rng):
    out = []
    for i in range(n):
        k = rng.randint(1, 9) if i % 7 else 60  # every 7th report is long enough to be truncated at 256 tokens
        out.append((". ".join(rng.choice(FINDINGS) for _ in range(k)) + ".", rng.choice(LLM)))
    return out


def main():
    ref_shim.install()
    import module.pretrain_datasets as pd_ref
    import tokenizers
    tok = tokenizers.Tokenizer.from_file(os.path.join(ref_shim.REF_ROOT, "dataset", "mimic_wordpiece.json"))
    vocab = tok.get_vocab()
    V = max(vocab.values()) + 1
    is_ent, is_sub = np.zeros(V, dtype=bool), np.zeros(V, dtype=bool)
    for w, i in vocab.items():
        is_ent[i] = w in pd_ref.entities
        is_sub[i] = w[0:2] == "##"
    ds = object.__new__(pd_ref.ContextBertDataset)   # bypass __init__ (CSV files, torchvision transforms)
    ds.max_caption_length, ds.tokenizer = L, tok
    ds.idxtoword = {v: k for k, v in vocab.items()}
    ds.transform = lambda img: torch.zeros(3, 2, 2)
    pd_ref.pil_loader = lambda path: None
    rng = random.Random(20250101)
    reps = reports(40, rng)
    ds.images_list = ["x"] * len(reps)
    ds.report_list = [r for r, _ in reps]
    ds.llm_out_list = [l for _, l in reps]
    ds.attn_i_list, ds.attn_j_list = [i % 3 for i in range(len(reps))], [(i // 3) % 3 for i in range(len(reps))]

    item = {k: [] for k in ("seed", "ids", "masked", "weights", "attn", "index")}
    for idx in range(len(reps)):
        for rep in range(2):
            seed = 1000 * idx + rep
            random.seed(seed)
            _, ids, am, ty, masked, weights, col, row = ds[idx]
            item["seed"].append(seed); item["index"].append(idx)
            item["ids"].append(ids[0].numpy()); item["masked"].append(masked[0].numpy()); item["weights"].append(weights[0].numpy().copy())
            item["attn"].append(am[0].numpy())

    # direct _context_mask calls on random / adversarial sequences
    ent_ids = np.nonzero(is_ent)[0]
    sub_ids = np.nonzero(is_sub)[0]
    plain = np.array([i for i in range(20, 2000) if not is_ent[i] and not is_sub[i]])
    cm = {k: [] for k in ("seed", "tokens", "masked", "mask_pos")}
    g = np.random.Generator(np.random.PCG64(7))
    for case in range(64):
        n = int(g.integers(1, L - 1)) if case % 8 else L - 1      # valid length (position 0 = CLS); every 8th: no padding at all
        t = np.zeros(L, dtype=np.int64)
        t[0] = 2
        body = g.choice(plain, size=L)
        kind = g.random(L)
        body = np.where(kind < 0.12, g.choice(ent_ids, size=L), body)
        body = np.where((kind >= 0.12) & (kind < 0.30), g.choice(sub_ids, size=L), body)
        body = np.where((kind >= 0.30) & (kind < 0.36), 16, body)
        if case % 5 == 0:
            body = np.where(np.isin(body, ent_ids), g.choice(plain, size=L), body)   # no entity at all -> 75 % branch
        t[1:n + 1] = body[1:n + 1]
        if case % 9 == 1 and n > 4:
            t[1], t[2] = ent_ids[0], sub_ids[0]                                       # entity at position 1, subword right after
        if case % 9 == 2 and n > 6:
            t[3] = 3                                                                  # a literal [MASK] in the input
        seed = 500000 + case
        random.seed(seed)
        masked, mp = ds._context_mask(torch.tensor(t)[None])
        cm["seed"].append(seed); cm["tokens"].append(t); cm["masked"].append(masked[0].numpy())
        m = np.zeros(L, dtype=bool); m[mp] = True
        cm["mask_pos"].append(m)

    np.savez_compressed(OUT, is_entity=np.packbits(is_ent), is_subword=np.packbits(is_sub), vocab_size=np.array(V),
                        item_seed=np.array(item["seed"]), item_index=np.array(item["index"]), item_ids=np.stack(item["ids"]),
                        item_masked=np.stack(item["masked"]), item_weights=np.stack(item["weights"]), item_attn=np.stack(item["attn"]),
                        item_reports=np.array(ds.report_list), item_llm=np.array(ds.llm_out_list),
                        cm_seed=np.array(cm["seed"]), cm_tokens=np.stack(cm["tokens"]), cm_masked=np.stack(cm["masked"]),
                        cm_mask_pos=np.stack(cm["mask_pos"]))
    print("wrote", OUT, "items", len(item["seed"]), "mask cases", len(cm["seed"]))


if __name__ == "__main__":
    main()
