"""ContextBertDataset -- the reference's data pipeline (ECAMP/Pre-training/module/pretrain_datasets.py:34-239), SURVEY.md 8(f) f2.

What is different from the reference, and why:
  * entity-aware masking (`_context_mask`, :60-110) and the template down-weighting / re-normalisation (:141-184) are
    restated as BATCHED TENSOR code (`context_mask`, `template_weights`) that runs on the host or on the GPU, instead of a
    Python loop over 256 positions per sample: at ~5 k pairs/s/GPU x 8 GPUs the per-token loop cannot feed the trainer.
    Given the same sequential stream of uniforms (the reference consumes `random.random()` in position order) the result is
    bit-identical to the reference loop -- checked against vectors captured from the reference (tests/golden/data_pipeline.npz).
  * the reference's collate_fn `.squeeze()`s a batch of one down to 1-D (:218-225); ours keeps the batch dimension.
Same constructor, same item tuple, same batch dict schema (:228-237).
"""
import os
import random

import numpy as np
from typing import List, Tuple

import torch
from torch.utils.data import Dataset

# pretrain_datasets.py:17-24 -- the 44 entities whose context is kept, and the token ids of the down-weighted templates
entities = ['abnormality', 'abscess', 'aerate', 'aorta', 'atelectasis', 'bronchiectasis', 'calcification', 'cardiomediastinal',
            'cardiomegaly', 'catheter', 'chf', 'collapse', 'congestion', 'consolidation', 'contour', 'COPD',
            'deformity', 'dilation', 'distention', 'edema', 'effusion', 'embolism', 'emphysema', 'engorgement',
            'fibrosis', 'fracture', 'granuloma', 'hernia', 'hilar', 'hyperinflate', 'hemidiaphragm', 'infiltrate',
            'mass', 'nodule', 'obscure', 'opacity', 'perihilar', 'pneumonia', 'pneumothorax', 'sarcoidosis',
            'silhouette', 'thickening', 'tuberculosis', 'vasculature']
template1 = [219, 149, 152, 422, 158]  # "there is no evidence of"
template2 = [219, 149, 152]            # "there is no"
PAD, MASK, PERIOD = 0, 3, 16            # token ids the reference hard-codes (:74,77,93)


class MaskVocab:
    """Per-token-id flags the masker needs: `word in entities` (:68,86) and `word[0:2] == '##'` (:77,81)."""

    def __init__(self, is_entity: torch.Tensor, is_subword: torch.Tensor):
        self.is_entity, self.is_subword = is_entity.bool(), is_subword.bool()

    @classmethod
    def from_tokenizer(cls, tokenizer):
        vocab = tokenizer.get_vocab()
        n = max(vocab.values()) + 1
        ent, sub = torch.zeros(n, dtype=torch.bool), torch.zeros(n, dtype=torch.bool)
        es = set(entities)
        for w, i in vocab.items():
            ent[i] = w in es
            sub[i] = w[0:2] == '##'
        return cls(ent, sub)

    def to(self, device):
        return MaskVocab(self.is_entity.to(device), self.is_subword.to(device))


def context_mask(tokens: torch.Tensor, vocab: MaskVocab, stream: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched `_context_mask` (pretrain_datasets.py:60-110).

    tokens [B, L] int64; stream [B, >= 2L-2] float64 uniforms, consumed per row in the order the reference calls
    `random.random()`: one per visited non-subword position (ascending), then one per entity position (ascending).
    -> (masked_tokens [B, L], mask_pos [B, L] bool: the entity-context positions the reference returns as a list).
    The loop's data dependencies, all resolved without a loop:
      * it stops at the first PAD at or after position 1 and never visits position L-1 (:73-75);
      * a subword token copies the masked state of the token before it -- by induction the pass-1 state of its word's first
        token (the 75 % entity masking of the second loop happens later and is NOT propagated) (:77-82);
      * `i not in mask_pos` (:104) is always true when position i is visited (an entity only registers positions before
        itself), so context positions are protected from nothing; they only carry the expanded loss weight;
      * the assignment at :95-96 is dead (it sits under `if word in entities` and tests `word not in entities`)."""
    B, L = tokens.shape
    dev = tokens.device
    idx = torch.arange(L, device=dev)[None, :]
    inner = (idx >= 1) & (idx <= L - 2)
    is_pad = (tokens == PAD) & inner
    first_pad = torch.where(is_pad.any(1), is_pad.to(torch.int64).argmax(1), torch.full((B,), L, device=dev, dtype=torch.int64))
    valid = inner & (idx < first_pad[:, None])
    sub = vocab.is_subword[tokens]
    ent_tok = vocab.is_entity[tokens]
    entity_exist = (ent_tok & inner).any(1)                       # :67-70 scans every position
    head = valid & ~sub                                           # positions that draw a number in the first loop
    ent = head & ent_tok                                          # entity_pos
    k1 = head.to(torch.int64).cumsum(1) - 1
    prob1 = stream.gather(1, k1.clamp_min(0))
    m_head = head & torch.where(entity_exist[:, None], (prob1 < 0.7) & ~ent, prob1 < 0.75)    # :98-105
    state = m_head | (tokens == MASK)                             # "masked_tokens[i-1] == 3" as seen by the next subword
    h_idx = torch.where(~sub, idx.expand(B, L), torch.zeros_like(tokens)).cummax(1).values   # first token of the word at i
    m_sub = valid & sub & state.gather(1, h_idx)                  # :77-79
    n1 = head.to(torch.int64).sum(1)
    k2 = n1[:, None] + ent.to(torch.int64).cumsum(1) - 1
    prob2 = stream.gather(1, k2.clamp(0, stream.shape[1] - 1))
    m_ent = ent & (prob2 < 0.75)                                  # :107-111
    masked = torch.where(m_head | m_sub | m_ent, torch.full_like(tokens, MASK), tokens)
    not_period = tokens != PERIOD
    mask_pos = torch.zeros_like(valid)
    mask_pos[:, 1:L - 1] |= ent[:, 2:L] & not_period[:, 1:L - 1]      # i-1 for an entity at i >= 2
    mask_pos[:, 1:L - 2] |= ent[:, 3:L] & not_period[:, 1:L - 2]      # i-2 for an entity at i >= 3
    return masked, mask_pos


def template_weights(ids: torch.Tensor, mask_pos: torch.Tensor) -> torch.Tensor:
    """Per-token loss weights (pretrain_datasets.py:141-184): 0.05 on "there is no [evidence of]" templates, the entity-context
    positions scaled up so that the row's total weight is preserved, or every weight scaled when no context exists.
    ids [B, L] ORIGINAL token ids, mask_pos [B, L] bool -> f32 [B, L].  The templates cannot overlap themselves or each other
    (token 219 only opens them), so the reference's skip-ahead scan equals independent matching at every start i < L-4."""
    B, L = ids.shape
    dev = ids.device
    n = L - 4

    def match(t):
        m = torch.ones((B, n), dtype=torch.bool, device=dev)
        for k, v in enumerate(t):
            m &= ids[:, k:k + n] == v
        return m

    m1 = match(template1)
    m2 = match(template2) & ~m1
    dim = torch.zeros((B, L), dtype=torch.bool, device=dev)
    for k in range(5):
        dim[:, k:k + n] |= m1
    for k in range(3):
        dim[:, k:k + n] |= m2
    weights = torch.where(dim, torch.full((B, L), 0.05, device=dev), torch.ones((B, L), device=dev))
    dcnt = dim.sum(1).to(torch.float64)
    mcnt = mask_pos.sum(1).to(torch.float64)
    ldm = (mask_pos & dim).sum(1).to(torch.float64)
    case_a = (mcnt > 0) & (dcnt > 0)
    case_b = ~case_a & (dcnt > 0)
    one = torch.ones_like(dcnt)
    exp_a = (0.95 * (dcnt - ldm) + mcnt) / torch.where(case_a, mcnt - 0.95 * ldm, one)           # :177-178
    exp_b = float(L) / (float(L) - 0.95 * dcnt)                                                  # :182-183
    sa = torch.where(case_a, exp_a, one).to(torch.float32)[:, None]
    sb = torch.where(case_b, exp_b, one).to(torch.float32)[:, None]
    weights = torch.where(mask_pos, weights * sa, weights)
    return weights * sb


def assemble_report(report: str, llm_output: str, rng) -> str:
    """pretrain_datasets.py:116-132: with probability 0.8 splice the LLM summary between two sentences of the report."""
    parts = report.split('.')
    n = len(parts)
    sent = ""
    if rng.random() < 0.8:
        location = rng.randint(0, n)
        for i in range(0, location):
            sent += parts[i]
            sent += "."
        sent += llm_output
        for i in range(location, n):
            sent += parts[i]
            sent += "."
    else:
        sent = report
    sent = sent.replace("..", ".")
    return '[CLS] ' + sent


def pil_loader(path: str):
    from PIL import Image
    with open(path, 'rb') as f:
        img = Image.open(f)
        return img.convert('RGB')


def random_resized_crop_params(width, height, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision 0.14.1 `RandomResizedCrop.get_params` (transforms/transforms.py; the reference's first transform,
    pretrain_datasets.py:48) restated: (top i, left j, h, w) of the crop, drawn from torch's GLOBAL generator with the same calls in
    the same order, so a worker seeded like the reference's draws the reference's boxes.  Two details that matter for equal boxes:
    the log-ratio bounds are float32 (`torch.log(torch.tensor(ratio))`) and the aspect ratio is exp'ed in float32 before `.item()`;
    after ten misses the fallback is the ratio-clamped CENTRAL crop (whole width at ratio 3/4 for a narrow image, whole height at
    4/3 for a wide one, else the whole image) -- rounds 1-5 took a min-side square there."""
    import math
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def random_flip():
    """torchvision `RandomHorizontalFlip.forward`: `torch.rand(1) < p` with p = 0.5 (pretrain_datasets.py:49)."""
    return bool(torch.rand(1) < 0.5)


def default_image_transform(size=448, image_u8=False):
    """pretrain_datasets.py:47-52 (RandomResizedCrop(448, scale=(0.2,1), bicubic) / flip / grayscale x3 / normalise) with PIL and
    torch only (torchvision is not a dependency).  Draws from the torch RNG as torchvision's transforms do
    (`random_resized_crop_params`, `random_flip`), so Python's `random` stream -- which the text half of the item consumes -- is left
    exactly as in the reference.
    image_u8: stop before ToTensor / Normalize and return the grayscale crop itself, uint8 [size, size] -- one byte per pixel instead
    of twelve through the DataLoader, pinned memory, PCIe and HBM; the model's kernels normalise it on the fly to the same bits."""
    from PIL import Image

    def tf(img):
        w, h = img.size
        i, j, ch, cw = random_resized_crop_params(w, h)
        img = img.crop((j, i, j + cw, i + ch)).resize((size, size), Image.BICUBIC)   # torchvision F.resized_crop on a PIL image
        if random_flip():
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if image_u8:
            return torch.from_numpy(np.array(img.convert('L'), dtype=np.uint8))
        g = np.asarray(img.convert('L'), dtype=np.float32) / 255.0
        t = torch.from_numpy((g - 0.4721) / 0.3037)
        return t[None].expand(3, size, size).contiguous()

    return tf


# ---- the image half on the device (csrc/augment.hip) --------------------------------------------------------------
# At 7 k pairs/s per GPU the per-sample PIL path above cannot feed the trainer (tools/loader_rate.py: a full-size MIMIC-CXR JPEG costs a
# worker tens of milliseconds to decode and resample).  The path below keeps the reference's RESULT and moves the work: radiographs are
# decoded ONCE, offline, into uint8 grayscale shards (`U8ShardWriter`); a loader worker only draws the crop box and the flip (the same
# torch-RNG calls in the same order) and copies the box's bytes; the device resamples every crop of the batch to 448 x 448 with
# Pillow's exact integer arithmetic (`DeviceAugmenter` -> ecamp_resample_crops_u8) and hands the model the uint8 [B, 448, 448] schema
# it already reads.  Byte for byte the `image_u8` item of `default_image_transform` on the same stored pixels (tests/test_augment.py).


class U8ShardWriter:
    """Pre-decoded radiographs, back to back in one flat uint8 file + an index [N, 3] = (byte offset, H, W) (`<path>.idx.npy`).
    `add` takes a PIL image or a uint8 [H, W] array; colour inputs go through PIL's convert('L') -- for MIMIC-CXR-JPG (grayscale JPEGs
    the reference opens as RGB, pretrain_datasets.py:28-31) that is the identity on the common channel.  `max_side`: optionally
    shrink the stored image so that its longer side is at most that many pixels (Pillow bicubic) -- a LOSSY choice of the user
    (the reference crops from the full-size image); without it the crops are the reference's pixels."""

    def __init__(self, path, max_side=None):
        self.path, self.max_side = path, max_side
        self.f = open(path, "wb")
        self.index = []

    def add(self, img):
        from PIL import Image
        if not isinstance(img, Image.Image):
            img = Image.fromarray(np.ascontiguousarray(img, dtype=np.uint8), "L")
        img = img.convert("L")
        if self.max_side and max(img.size) > self.max_side:
            r = self.max_side / float(max(img.size))
            img = img.resize((max(1, int(round(img.size[0] * r))), max(1, int(round(img.size[1] * r)))), Image.BICUBIC)
        a = np.asarray(img, dtype=np.uint8)
        self.index.append((self.f.tell(), a.shape[0], a.shape[1]))
        self.f.write(a.tobytes())
        return len(self.index) - 1

    def close(self):
        self.f.close()
        np.save(self.path + ".idx.npy", np.asarray(self.index, dtype=np.int64).reshape(-1, 3))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class U8ShardReader:
    """Memory-mapped view of a `U8ShardWriter` file: `reader[i]` -> uint8 [H, W] (no copy, no decode)."""

    def __init__(self, path):
        self.index = np.load(path + ".idx.npy")
        self.data = np.memmap(path, dtype=np.uint8, mode="r")

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        off, h, w = (int(v) for v in self.index[i])
        return self.data[off:off + h * w].reshape(h, w)


def device_crop_item(image):
    """The loader worker's share of the device path: draw the crop box and the flip exactly as `default_image_transform` does (same
    torch-RNG calls, same order) and return the BOX's bytes, contiguous uint8 [h, w], + the flip.  `image`: uint8 [H, W] (a shard
    view) or a PIL image."""
    if not isinstance(image, np.ndarray):
        image = np.asarray(image.convert("L"), dtype=np.uint8)
    H, W = image.shape
    i, j, h, w = random_resized_crop_params(W, H)
    flip = random_flip()
    return np.ascontiguousarray(image[i:i + h, j:j + w]), flip


def crops_meta(table):
    """(largest side, sum of h, largest h) of a crop table: what sizes the device call's tap tables and workspace.  Plain Python ints, taken
    on the host where the table is made, so that the training step never reads the table back from the device."""
    return int(max(table[:, 1].max(), table[:, 2].max())), int(table[:, 1].sum()), int(table[:, 1].max())


def pack_crops(items, pin=False):
    """Batch of `device_crop_item` results -> (flat uint8 tensor of all crops back to back, int64 table [B, 6] for
    ecamp_resample_crops_u8: byte offset, h, w, flip, first intermediate row, 0)."""
    sizes = [c.shape for c, _ in items]
    total = sum(h * w for h, w in sizes)
    flat = torch.empty((total,), dtype=torch.uint8, pin_memory=pin)
    table = torch.zeros((len(items), 6), dtype=torch.int64, pin_memory=pin)
    fa = flat.numpy()
    off = row = 0
    for n, (c, flip) in enumerate(items):
        h, w = c.shape
        fa[off:off + h * w] = c.reshape(-1)
        table[n, 0], table[n, 1], table[n, 2], table[n, 3], table[n, 4] = off, h, w, int(flip), row
        off += h * w
        row += h
    return flat, table


class DeviceAugmenter:
    """crops (flat uint8 + table, host or device) -> uint8 [B, size, size] on the device: RandomResizedCrop's resize + flip + Grayscale,
    Pillow-exact (csrc/augment.hip).  The workspace grows to the largest batch seen and is reused (kernels never allocate)."""

    def __init__(self, device, size=448):
        self.device, self.size = torch.device(device), size
        self.ws = None
        self.err = torch.zeros((1,), dtype=torch.int32, device=self.device)

    @staticmethod
    def taps(max_side, size):
        """Pillow's ksize for the batch's largest scale factor: 2 * ceil(2 * max(1, max_side / size)) + 1."""
        import math
        return 2 * int(math.ceil(2.0 * max(1.0, max_side / float(size)))) + 1

    def __call__(self, flat, table, meta=None, check=False):
        """meta = `crops_meta(table)` made on the host (the collate function's `image_meta`); without it a table that already lives on the
        device is read back (a host synchronisation: tests only)."""
        from .. import hip_ops as ops
        if meta is None:
            meta = crops_meta(table if table.device.type == "cpu" else table.cpu())
        B = table.shape[0]
        max_side, rows, max_h = (int(v) for v in meta)
        kmax = self.taps(max_side, self.size)
        need = ops.resample_crops_workspace_bytes(B, self.size, kmax, rows)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        out = torch.empty((B, self.size, self.size), dtype=torch.uint8, device=self.device)
        ops.resample_crops_u8(flat.to(self.device, non_blocking=True), table.to(self.device, non_blocking=True), out, self.size, kmax, rows, max_h,
                              self.ws, self.err)
        if check and int(self.err.item()) != 0:
            raise RuntimeError("ecamp_resample_crops_u8: a crop needed more taps than the table was sized for")
        return out


class ContextBertDataset(Dataset):
    """Same constructor and item tuple as the reference class (pretrain_datasets.py:34-199).  `data_root` holds
    `mimic_wordpiece.json`, `mimic-cxr-2.0.0-entity-llm.csv` (img_path, report, llm_output) and
    `mimic-cxr-2.0.0-attn-label.csv` (label_i, label_j)."""

    def __init__(self, data_root, max_caption_length: int = 256, transform=None, image_u8=False, image_shard=None):
        """image_shard: path of a `U8ShardWriter` file holding the radiographs of the CSV in row order, pre-decoded -- the DEVICE image
        pipeline: an item then carries the bytes of its crop box instead of a resampled image (`device_crop_item`), `collate_fn` packs
        them (`image_crops`, `image_table`) and `ECAMP.forward` resamples the batch on the GPU (same uint8 item, byte for byte)."""
        import tokenizers
        self.shard = U8ShardReader(image_shard) if image_shard else None
        self.max_caption_length = max_caption_length
        self.data_root = data_root
        self.images_list, self.report_list, self.llm_out_list, self.attn_i_list, self.attn_j_list = self.read_csv()
        self.tokenizer = tokenizers.Tokenizer.from_file(os.path.join(self.data_root, "mimic_wordpiece.json"))
        self.tokenizer.enable_truncation(max_length=self.max_caption_length)
        self.tokenizer.enable_padding(length=self.max_caption_length)
        self.vocab = MaskVocab.from_tokenizer(self.tokenizer)
        self.transform = transform if transform is not None else default_image_transform(448, image_u8=image_u8)

    def __len__(self):
        return len(self.images_list)

    def read_csv(self):
        import pandas as pd
        df = pd.read_csv(os.path.join(self.data_root, 'mimic-cxr-2.0.0-entity-llm.csv'), sep=',')
        df_attn = pd.read_csv(os.path.join(self.data_root, 'mimic-cxr-2.0.0-attn-label.csv'), sep=',')
        return df["img_path"], df["report"], df["llm_output"], df_attn["label_i"], df_attn["label_j"]

    def text_item(self, index, rng=random):
        """Everything of __getitem__ but the image (:113-190): ids, attention_mask, type_ids, masked_ids, weights (all [1, L])."""
        sent = assemble_report(self.report_list[index], self.llm_out_list[index], rng)
        encoded = self.tokenizer.encode(sent)
        ids = torch.tensor(encoded.ids).unsqueeze(0)
        attention_mask = torch.tensor(encoded.attention_mask).unsqueeze(0)
        type_ids = torch.tensor(encoded.type_ids).unsqueeze(0)
        L = ids.shape[1]
        stream = torch.tensor([[rng.random() for _ in range(2 * L)]], dtype=torch.float64)
        masked_ids, mask_pos = context_mask(ids, self.vocab, stream)
        weights = template_weights(ids, mask_pos)
        return ids, attention_mask, type_ids, masked_ids, weights

    def __getitem__(self, index):
        if getattr(self, "shard", None) is not None:
            image = device_crop_item(self.shard[index])          # (uint8 [h, w] crop bytes, flip): same RNG draws as self.transform
        else:
            image = self.transform(pil_loader(self.images_list[index]))
        ids, attention_mask, type_ids, masked_ids, weights = self.text_item(index)
        column = torch.tensor(self.attn_i_list[index]).unsqueeze(0)
        row = torch.tensor(self.attn_j_list[index]).unsqueeze(0)
        return image, ids, attention_mask, type_ids, masked_ids, weights, column, row

    def collate_fn(self, instances: List[Tuple]):
        cols = list(zip(*instances))
        st = lambda i: torch.cat(cols[i], 0)   # items are [1, L] / [1]: concatenating keeps the batch dimension at B == 1
        out = {"labels": st(1), "attention_mask": st(2), "type_ids": st(3), "ids": st(4), "weights": st(5), "column": st(6), "row": st(7)}
        if getattr(self, "shard", None) is not None:
            out["image_crops"], out["image_table"] = pack_crops(cols[0])
            out["image_meta"] = crops_meta(out["image_table"])   # Python ints: stay on the host through the prefetcher
        else:
            out["image"] = torch.stack(cols[0])
        return out


class DeviceMasker:
    """Mask-on-device path for pre-tokenised reports: `ids` [B, L] (already on the GPU) -> (masked ids, weights) with one
    batched call, uniforms from a torch generator.  Statistically the reference's masker; not the same random stream."""

    def __init__(self, vocab: MaskVocab, device, seed=0):
        self.vocab = vocab.to(device)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __call__(self, ids):
        B, L = ids.shape
        stream = torch.rand((B, 2 * L), generator=self.gen, device=ids.device, dtype=torch.float64)
        masked, mask_pos = context_mask(ids, self.vocab, stream)
        return masked, template_weights(ids, mask_pos)


def measure_item_rate(dataset, seconds=2.0, max_items=256):
    """Items per second ONE loader worker gets out of `dataset.__getitem__` (image decode + augmentation on PIL, tokenisation, the
    masking arithmetic): what `--num_workers` has to be multiplied with to feed a GPU.  At ~6.8 k pairs/s per MI355X a node of
    eight needs ~55 k items/s; main_pretrain.py prints this figure next to the step rate so that a loader-bound run is visible."""
    import time
    # The probe runs a wall-clock-bounded number of items, and every item draws from Python's `random` (text masking), torch's global
    # generator (crop / flip) and possibly numpy's: without the save / restore below, the state those generators are in when the model
    # is initialised and the loader is built would depend on how fast this machine is -- and a fixed --seed would no longer fix the
    # run (the reference derives item stream and initialisation from the seed alone, main_pretrain.py:188-191).
    state = (random.getstate(), torch.get_rng_state(), np.random.get_state())
    try:
        t0, n = time.time(), 0
        while n < max_items and (n < 4 or time.time() - t0 < seconds):
            dataset[n % len(dataset)]
            n += 1
        return n / max(time.time() - t0, 1e-9)
    finally:
        random.setstate(state[0])
        torch.set_rng_state(state[1])
        np.random.set_state(state[2])
