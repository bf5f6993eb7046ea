"""SURVEY.md 8(f) f2, image half (ECAMP/Pre-training/module/pretrain_datasets.py:47-52,113-115): the crop-box / flip draws, the host PIL
item, and the device path (offline uint8 shards -> per-sample crop bytes -> csrc/augment.hip) that replaces it -- byte for byte."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "image_transform.npz")


def _radiograph(W, H, seed):
    from oracle.make_golden_image import synthetic_radiograph
    return synthetic_radiograph(W, H, seed)


def test_crop_boxes_and_flips_match_the_torchvision_restatement():
    """`random_resized_crop_params` / `random_flip` draw what torchvision 0.14.1's RandomResizedCrop.get_params / RandomHorizontalFlip draw
    under the same torch seed (fixture from oracle/tv_transforms.py: first-try hits, misses, and the three fallbacks -- a narrow image
    keeps its whole WIDTH at ratio 3/4, a wide one its whole HEIGHT at 4/3; rounds 1-5 took a min-side square)."""
    from ecamp_amd.module import pretrain_datasets as pd
    g = np.load(GOLD)
    shapes = g["shapes"]
    seen_fallback = 0
    for n, seed, i, j, h, w, flip in g["params"]:
        W, H = (int(v) for v in shapes[n])
        torch.manual_seed(int(seed))
        got = pd.random_resized_crop_params(W, H) + (int(pd.random_flip()),)
        assert got == (i, j, h, w, flip), ((W, H), seed, got, (i, j, h, w, flip))
        seen_fallback += (W, H) in ((100, 2000), (2000, 100))
    assert seen_fallback == 12
    torch.manual_seed(0)
    assert pd.random_resized_crop_params(100, 2000)[2:] == (133, 100)      # whole width, h = round(100 / (3/4)), centred
    assert pd.random_resized_crop_params(2000, 100)[2:] == (100, 133)      # whole height, w = round(100 * 4/3)


def test_pillow_resample_restatement_equals_pillow():
    """oracle/tv_transforms.pillow_resize_u8 (the algorithm csrc/augment.hip implements) against the installed Pillow: equal bytes on
    down-scaling (widened support), up-scaling, identity and very anisotropic crops."""
    from PIL import Image
    from oracle import tv_transforms as tv
    rng = np.random.default_rng(3)
    for H, W in ((700, 900), (448, 448), (300, 200), (449, 447), (133, 100), (1500, 1100), (60, 2000)):
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        ref = np.array(Image.fromarray(img, "L").resize((448, 448), Image.BICUBIC))
        assert np.array_equal(tv.pillow_resize_u8(img, 448), ref), (H, W)
    img = _radiograph(640, 480, 1)
    rgb = Image.fromarray(np.stack([img] * 3, -1), "RGB").resize((448, 448), Image.BICUBIC).convert("L")   # the reference's order: RGB resize, then L
    assert np.array_equal(np.array(rgb), tv.pillow_resize_u8(img, 448))


def test_host_item_matches_the_reference_restatement_and_fixture():
    """`default_image_transform` (PIL) on a grayscale radiograph opened as RGB, like pil_loader does: the uint8 item and the normalised f32
    item equal the restated reference pipeline under the same torch seed, and the committed fixture (box, flip, crc32, samples)."""
    from PIL import Image
    from ecamp_amd.module import pretrain_datasets as pd
    from oracle import tv_transforms as tv
    g = np.load(GOLD)
    for n, (W, H, iseed, tseed, i, j, h, w, flip, crc) in enumerate(g["items"]):
        img = _radiograph(int(W), int(H), int(iseed))
        pil = Image.fromarray(img, "L").convert("RGB")
        torch.manual_seed(int(tseed))
        u8 = pd.default_image_transform(448, image_u8=True)(pil).numpy()
        torch.manual_seed(int(tseed))
        want, p = tv.reference_item_u8(img, 448)
        assert p == (i, j, h, w, flip)
        assert np.array_equal(u8, want) and zlib.crc32(u8.tobytes()) == int(crc)
        assert np.array_equal(u8[::37, ::41], g["item%d_sample" % n])
        torch.manual_seed(int(tseed))
        f32 = pd.default_image_transform(448)(pil)
        assert torch.equal(f32, tv.to_tensor_normalize(want))
        assert np.array_equal(f32[:, ::37, ::41].numpy(), g["item%d_f32_sample" % n])


def test_shards_and_crop_items_round_trip(tmp_path):
    """U8ShardWriter / U8ShardReader return the stored pixels; `device_crop_item` consumes the torch RNG exactly like the host transform
    (same box, same flip, generator left in the same state) and `pack_crops` lays the crops out as the kernel's table says."""
    from ecamp_amd.module import pretrain_datasets as pd
    imgs = [_radiograph(300 + 40 * k, 260 + 30 * k, k) for k in range(4)]
    path = os.path.join(tmp_path, "shard.u8")
    with pd.U8ShardWriter(path) as w:
        for im in imgs:
            w.add(im)
    rd = pd.U8ShardReader(path)
    assert len(rd) == 4 and all(np.array_equal(rd[k], imgs[k]) for k in range(4))
    items = []
    for k in range(4):
        torch.manual_seed(50 + k)
        crop, flip = pd.device_crop_item(rd[k])
        st = torch.get_rng_state()
        torch.manual_seed(50 + k)
        H, W = imgs[k].shape
        i, j, h, w_ = pd.random_resized_crop_params(W, H)
        assert flip == pd.random_flip() and torch.equal(st, torch.get_rng_state())
        assert np.array_equal(crop, imgs[k][i:i + h, j:j + w_])
        items.append((crop, flip))
    flat, table = pd.pack_crops(items)
    off = row = 0
    for k, (crop, flip) in enumerate(items):
        h, w_ = crop.shape
        assert table[k].tolist() == [off, h, w_, int(flip), row, 0]
        assert np.array_equal(flat[off:off + h * w_].numpy().reshape(h, w_), crop)
        off += h * w_
        row += h
    with pd.U8ShardWriter(os.path.join(tmp_path, "small.u8"), max_side=128) as w:
        w.add(imgs[3])
    assert max(pd.U8ShardReader(os.path.join(tmp_path, "small.u8"))[0].shape) == 128


def test_tap_count_formula_matches_pillow_coefficients():
    """`DeviceAugmenter.taps` (the size of the device's tap tables) equals the ksize of Pillow's precompute_coeffs for every input size, and
    the restated coefficients sum to 2^22 +- rounding per output pixel (22-bit fixed point, normalised)."""
    from ecamp_amd.module import pretrain_datasets as pd
    from oracle import tv_transforms as tv
    for n in (1, 2, 5, 37, 200, 447, 448, 449, 897, 1024, 2544, 3056, 9000):
        kk, bounds = tv.pillow_coeffs(n, 448)
        assert pd.DeviceAugmenter.taps(n, 448) == kk.shape[1], n
        assert (bounds[:, 1] >= 1).all() and (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= n).all()
        assert np.abs(kk.sum(1) - (1 << 22)).max() <= kk.shape[1], n


@pytest.mark.gpu
def test_device_augmenter_equals_the_host_pil_item(dev):
    """The device path against the host path on the same stored pixels and the same torch seed: ecamp_resample_crops_u8 returns the bytes
    PIL's crop().resize(BICUBIC) + flip returns -- big down-scaling (a 2544 x 3056 radiograph: 29 taps), up-scaling (a 300 x 200 one),
    both fallbacks, ragged crop sizes in one batch; and the model reads the result as its `image_u8` schema."""
    from PIL import Image
    from ecamp_amd.module import pretrain_datasets as pd
    shapes = [(2544, 3056), (900, 1100), (300, 200), (100, 2000), (2000, 100), (448, 448), (1024, 1024), (37, 41)]
    imgs = [_radiograph(W, H, 10 + n) for n, (W, H) in enumerate(shapes)]
    want, items = [], []
    tf = pd.default_image_transform(448, image_u8=True)
    for n, im in enumerate(imgs):
        torch.manual_seed(900 + n)
        want.append(tf(Image.fromarray(im, "L").convert("RGB")))
        torch.manual_seed(900 + n)
        items.append(pd.device_crop_item(im))
    assert any(f for _, f in items) and not all(f for _, f in items)
    flat, table = pd.pack_crops(items, pin=True)
    aug = pd.DeviceAugmenter(dev)
    got = aug(flat, table, check=True).cpu()
    for n in range(len(imgs)):
        assert torch.equal(got[n], want[n]), (shapes[n], int((got[n] != want[n]).sum()))
    got2 = aug(flat.to(dev), table.to(dev), check=True).cpu()      # inputs already on the device; workspace reused
    assert torch.equal(got, got2)


@pytest.mark.gpu
def test_device_augmenter_at_batch_size(dev):
    """B = 256 crops of pre-shrunk radiographs (longer side 1024, what a shard with max_side = 1024 holds) in one call: every item equals the
    host item; prints the device time next to what the host transform costs per item."""
    import time
    from PIL import Image
    from ecamp_amd.module import pretrain_datasets as pd
    base = [_radiograph(848, 1024, s) for s in range(8)]
    B = 256
    tf = pd.default_image_transform(448, image_u8=True)
    items = []
    torch.manual_seed(4242)
    for b in range(B):
        items.append(pd.device_crop_item(base[b % 8]))
    flat, table = pd.pack_crops(items, pin=True)
    aug = pd.DeviceAugmenter(dev)
    out = aug(flat, table, check=True)
    torch.cuda.synchronize()
    fd, td = flat.to(dev), table.to(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        aug(fd, td)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / 5 * 1e3
    torch.manual_seed(4242)
    t0 = time.perf_counter()
    want = [tf(Image.fromarray(base[b % 8], "L").convert("RGB")) for b in range(B)]
    host_ms = (time.perf_counter() - t0) * 1e3 / B
    got = out.cpu()
    bad = [b for b in range(B) if not torch.equal(got[b], want[b])]
    print("  device: %d crops (%.0f MB of crop bytes) -> [%d, 448, 448] in %.2f ms; host PIL: %.2f ms per item on one core" %
          (B, flat.numel() / 1e6, B, dev_ms, host_ms))
    assert not bad, bad[:8]


@pytest.mark.gpu
def test_model_forward_on_crops_equals_forward_on_the_host_items(dev):
    """`ECAMP.forward` on a batch that carries crops (`image_crops` / `image_table` / `image_meta`, the `--image_shard` collate) gives the
    losses of the same batch with the host-made uint8 items (`image`), through the prefetcher as well (the table and the crop bytes cross PCIe
    on the copy stream; the Python-int meta stays on the host: no read-back in the step)."""
    from PIL import Image
    from ecamp_amd.data import DevicePrefetcher
    from ecamp_amd.module import model_ecamp as me
    from ecamp_amd.module import pretrain_datasets as pd
    from oracle import ecamp_oracle as orc
    from oracle import recipe
    cfg = orc.cfg_tiny()
    B, S = 4, 64
    model = me.ecamp_tiny(compute_dtype=torch.bfloat16)
    model.load_state_dict(recipe.recipe_state(cfg, seed=0))
    model.to(dev).eval()
    batch = recipe.recipe_batch(cfg, B, S, seed=3)
    noise = recipe.recipe_noise(B, cfg.num_patches, seed=3)
    imgs = [_radiograph(500 + 60 * k, 420 + 90 * k, 30 + k) for k in range(B)]
    tf = pd.default_image_transform(448, image_u8=True)
    host_items, crops = [], []
    for k, im in enumerate(imgs):
        torch.manual_seed(700 + k)
        host_items.append(tf(Image.fromarray(im, "L").convert("RGB")))
        torch.manual_seed(700 + k)
        crops.append(pd.device_crop_item(im))
    rest = {k: v for k, v in batch.items() if k != "image"}
    a = model(dict(rest, image=torch.stack(host_items)), noise=noise)
    flat, table = pd.pack_crops(crops, pin=True)
    crop_batch = dict(rest, image_crops=flat, image_table=table, image_meta=pd.crops_meta(table))
    b = model(crop_batch, noise=noise)
    c = [model(x, noise=noise) for x in DevicePrefetcher([crop_batch, crop_batch], dev)][-1]
    torch.cuda.synchronize()
    for x, y, z in zip(a, b, c):      # same pixels in, same kernels: equal up to the summation order of the loss sums' f32 atomics
        assert abs(x.item() - y.item()) <= 1e-5 * abs(x.item()) and abs(x.item() - z.item()) <= 1e-5 * abs(x.item()), (x.item(), y.item(), z.item())


REF_TOK = "/root/reference/ECAMP/Pre-training/dataset/mimic_wordpiece.json"


@pytest.mark.skipif(not os.path.exists(REF_TOK), reason="the reference tokenizer file is only present in the authoring container")
def test_dataset_with_image_shard_hands_over_crops_of_the_same_items(tmp_path):
    """`ContextBertDataset(data_root, image_shard=...)` end to end on a four-row data root written here (the reference's CSV columns, JPEG files,
    its tokenizer): every item of the shard-backed dataset carries the crop whose Pillow resample (+ flip) IS the `image_u8` item of the
    file-backed dataset under the same seeds; the text half is identical; `collate_fn` packs crops, table and host-side meta."""
    import random
    import shutil
    import pandas as pd
    from PIL import Image
    from ecamp_amd.module import pretrain_datasets as pdm
    from oracle import tv_transforms as tv
    root = str(tmp_path)
    shutil.copy(REF_TOK, os.path.join(root, "mimic_wordpiece.json"))
    paths, raws = [], []
    for k in range(4):
        a = _radiograph(420 + 50 * k, 380 + 40 * k, 60 + k)
        p = os.path.join(root, "img%d.png" % k)      # PNG: lossless, so that the shard and the file hold the same pixels
        Image.fromarray(a, "L").save(p)
        paths.append(p)
        raws.append(a)
    pd.DataFrame({"img_path": paths, "report": ["there is no evidence of pneumothorax. small left pleural effusion."] * 4,
                  "llm_output": ["effusion small."] * 4}).to_csv(os.path.join(root, "mimic-cxr-2.0.0-entity-llm.csv"), index=False)
    pd.DataFrame({"label_i": [0, 1, 2, 1], "label_j": [2, 1, 0, 0]}).to_csv(os.path.join(root, "mimic-cxr-2.0.0-attn-label.csv"), index=False)
    shard = os.path.join(root, "images.u8")
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_image_shard.py"), "--data_path", root, "--out", shard, "--workers", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    rd = pdm.U8ShardReader(shard)
    assert len(rd) == 4 and all(np.array_equal(rd[k], raws[k]) for k in range(4))   # the tool decodes the CSV's files in row order, pixel for pixel
    ds_file = pdm.ContextBertDataset(root, max_caption_length=64, image_u8=True)
    ds_shard = pdm.ContextBertDataset(root, max_caption_length=64, image_shard=shard)
    items = []
    for k in range(4):
        random.seed(100 + k); torch.manual_seed(200 + k)
        a = ds_file[k]
        random.seed(100 + k); torch.manual_seed(200 + k)
        b = ds_shard[k]
        crop, flip = b[0]
        want = tv.pillow_resize_u8(crop, 448)
        if flip:
            want = want[:, ::-1]
        assert np.array_equal(a[0].numpy(), want), k
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)
        items.append(b)
    batch = ds_shard.collate_fn(items)
    assert "image" not in batch and batch["image_table"].shape == (4, 6) and batch["image_crops"].dtype == torch.uint8
    assert batch["image_meta"] == pdm.crops_meta(batch["image_table"]) and batch["ids"].shape == (4, 64)
    assert int(batch["image_table"][:, 1].mul(batch["image_table"][:, 2]).sum()) == batch["image_crops"].numel()
