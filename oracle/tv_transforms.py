"""TEST INFRASTRUCTURE ONLY (never imported by the product).  CPU restatement of the third-party arithmetic behind the reference's
image transform (ECAMP/Pre-training/module/pretrain_datasets.py:47-52):

    transforms.RandomResizedCrop(448, scale=(0.2, 1.0), interpolation=BICUBIC)  -> torchvision 0.14.1 (environment.yml:134)
    transforms.RandomHorizontalFlip(), Grayscale(3), ToTensor(), Normalize([0.4721], [0.3037])
    on PIL images                                                               -> Pillow 10.4.0 (environment.yml:85)

Neither torchvision nor that Pillow is in this image (Pillow here: 12.2.0, same resample): the functions below restate the PUBLISHED
algorithms -- torchvision/transforms/transforms.py `RandomResizedCrop.get_params`, `RandomHorizontalFlip.forward`;
Pillow src/libImaging/Resample.c `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`, and Convert.c's L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 -- as slow, literal loops.
Pinned: `pillow_resize_u8` equals the installed Pillow byte for byte (tests/test_augment.py); the crop-box draws have no installed
counterpart to compare with ("parity unpinned" for RandomResizedCrop.get_params: restated from the source, fixture from the restatement).
"""
import math

import numpy as np
import torch


def tv_get_params(width, height, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), gen=None):
    """torchvision 0.14.1 RandomResizedCrop.get_params(img, scale, ratio) -> (i, j, h, w); draws from torch's global generator."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))                       # float32, as in the source
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)                           # fallback to central crop
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w = width
        h = height
    i = (height - h) // 2
    j = (width - w) // 2
    return i, j, h, w


def tv_flip(p=0.5):
    """RandomHorizontalFlip.forward: `if torch.rand(1) < self.p`."""
    return bool(torch.rand(1) < p)


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc: (int taps [out, ksize], bounds [out, 2])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << 22)) if k < 0 else int(0.5 + k * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def pillow_resize_u8(img, out):
    """Image.fromarray(img, 'L').resize((out, out), BICUBIC) for uint8 [H, W]: horizontal pass to a uint8 intermediate, then vertical."""
    H, W = img.shape
    kx, bx = pillow_coeffs(W, out)
    ky, by = pillow_coeffs(H, out)
    a = img.astype(np.int64)
    tmp = np.zeros((H, out), np.uint8)
    for xx in range(out):
        xmin, n = bx[xx]
        tmp[:, xx] = np.clip(((1 << 21) + (a[:, xmin:xmin + n] * kx[xx, :n]).sum(1)) >> 22, 0, 255)
    t = tmp.astype(np.int64)
    res = np.zeros((out, out), np.uint8)
    for yy in range(out):
        ymin, n = by[yy]
        res[yy] = np.clip(((1 << 21) + (t[ymin:ymin + n, :] * ky[yy, :n, None]).sum(0)) >> 22, 0, 255)
    return res


def reference_item_u8(img, size=448):
    """The reference's transform up to (not including) ToTensor on a grayscale radiograph `img` (uint8 [H, W]; the reference opens it as
    RGB with three equal channels): crop -> resize -> flip -> L.  -> (uint8 [size, size], (i, j, h, w, flip))."""
    H, W = img.shape
    i, j, h, w = tv_get_params(W, H)
    out = pillow_resize_u8(np.ascontiguousarray(img[i:i + h, j:j + w]), size)
    flip = tv_flip()
    if flip:
        out = out[:, ::-1].copy()
    # Grayscale(3) of an RGB image with R = G = B = v: (19595 v + 38470 v + 7471 v + 0x8000) >> 16 = v
    return out, (i, j, h, w, int(flip))


def to_tensor_normalize(u8):
    """ToTensor + Normalize(mean=[0.4721], std=[0.3037]) of the 3-channel grayscale image, f32 arithmetic as torchvision's."""
    t = torch.from_numpy(np.ascontiguousarray(u8)).to(torch.float32).div(255)
    t = t[None].expand(3, -1, -1).clone()
    return t.sub_(torch.as_tensor([0.4721], dtype=torch.float32).view(-1, 1, 1)).div_(torch.as_tensor([0.3037], dtype=torch.float32).view(-1, 1, 1))
