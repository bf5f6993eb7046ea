"""Fixed 2-D sin-cos position table (ECAMP/Pre-training/util/pos_embed.py:20-67), init-time host code.
float64 numpy like the reference (the reference's `np.float` alias is gone in numpy >= 1.24)."""
import numpy as np


def _sincos_1d(dim, positions):
    """[M] positions -> [M, dim]: sin block then cos block, frequencies 1/10000^(i/(dim/2))  (pos_embed.py:49-67)."""
    assert dim % 2 == 0
    freq = 1.0 / np.power(10000.0, np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    angles = positions.reshape(-1).astype(np.float64)[:, None] * freq[None, :]
    return np.concatenate([np.sin(angles), np.cos(angles)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """[grid*grid (+1), embed_dim]; first half of the channels encodes the W coordinate ("w goes first",
    pos_embed.py:28-30,43-44), second half the H coordinate; the cls row is zeros."""
    assert embed_dim % 2 == 0
    coords = np.arange(grid_size, dtype=np.float32)
    ww, hh = np.meshgrid(coords, coords)  # ww[y, x] = x, hh[y, x] = y
    table = np.concatenate([_sincos_1d(embed_dim // 2, ww), _sincos_1d(embed_dim // 2, hh)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, embed_dim]), table], axis=0)
    return table
