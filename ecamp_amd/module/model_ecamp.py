"""ECAMP model -- drop-in for ECAMP/Pre-training/module/model_ecamp.py (ToniChopp/ECAMP), MI355X-native.

Same public surface: `ecamp(**kwargs)` factory (model_ecamp.py:328-333), `ECAMP.forward(batch, mask_ratio=0.75)
-> (mim_loss, res_loss, mlm_loss)` (:303-325), identical `state_dict()` keys, nn.Module semantics
(`parameters()`, `train()/eval()`, `load_state_dict`).  Everything between is different: the forward and the
backward are ~35 hand-written stages of HIP kernels (ecamp_amd/functions.py) over flat HBM arenas
(ecamp_amd/arena.py).  There is no CPU execution path: calling the model without an MI355X raises.

Lifted hard-coded constants of the reference (defaults unchanged): `.cuda()` -> the model's device; Resize([224,224])
-> `img_size`; SR window 12 -> `sr_window` (scaled with the patch grid); `BertConfig()` -> `bert_config`;
`bert_mlp` out features 768 -> `bert_config.hidden_size`.
"""
from functools import partial

import os

import torch
import torch.nn as nn

from ..util.pos_embed import get_2d_sincos_pos_embed
from .bert_config import BertConfig
from .bert_encoder import MultiModalBertEncoder


class PatchEmbed(nn.Module):
    """timm 0.4.12 PatchEmbed parameter container (`proj` = Conv2d k=stride=patch)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    """timm 0.4.12 Block parameter container (norm1, attn.qkv/proj, norm2, mlp.fc1/fc2)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        assert qkv_bias, "the reference always builds Blocks with qkv_bias=True"
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class InterpolateConvSuperResolution(nn.Module):
    """Parameter container of model_ecamp.py:28-46 (bilinear x2 -> conv -> ReLU -> conv -> +skip -> ReLU)."""

    def __init__(self, scale_factor, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        if (scale_factor, in_channels, out_channels, kernel_size, stride, padding) != (2, 3, 3, 3, 1, 1):
            raise ValueError("the HIP SR head implements the reference configuration (x2, 3->3, 3x3, stride 1, pad 1) only")
        self.scale_factor = scale_factor
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)


class ECAMP(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=768,
                 decoder_depth=4, decoder_num_heads=6, mlp_ratio=4.0, norm_layer=nn.LayerNorm, norm_pix_loss=False,
                 bert_config=None, compute_dtype=torch.bfloat16, sr_window=None, fp8_forward=False, gelu_saved_grad=None):
        super().__init__()
        if in_chans != 3 or patch_size != 16:
            # (the SR head's fused kernels are built for 32-px super-patches = patch 16, the only value the reference constructs,
            # model_ecamp.py:328-333; refused here instead of at the first forward's ecamp_sr_fwd)
            raise ValueError("in_chans must be 3 and patch_size 16")
        for d, h in ((embed_dim, num_heads), (decoder_embed_dim, decoder_num_heads)):
            if d % h != 0 or d // h not in (32, 64, 128):
                raise ValueError("head_dim must be 32, 64 or 128 (got %d/%d)" % (d, h))
        if compute_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("compute_dtype must be torch.float32 (parity mode), torch.bfloat16 or torch.float16 (the reference's autocast format)")
        self.img_size, self.patch_size, self.embed_dim = img_size, patch_size, embed_dim
        self.num_heads, self.decoder_embed_dim, self.decoder_num_heads = num_heads, decoder_embed_dim, decoder_num_heads
        self.compute_dtype = compute_dtype
        # BASELINE.json configs[4]: the forward GEMMs of the ViT blocks (qkv, proj, fc1, fc2; encoder and decoder) on e4m3 copies of
        # activations and weights (per-tensor scales), gradients through the bf16 path.  bf16 compute dtype only.
        if fp8_forward and compute_dtype != torch.bfloat16:
            raise ValueError("fp8_forward needs compute_dtype=torch.bfloat16")
        self.fp8_forward = bool(fp8_forward)
        # fp8_forward also covers the MLM head (transform dense + vocabulary decoder) when set; ECAMP_FP8_HEAD=0/1 overrides the default
        self.fp8_head = os.environ.get("ECAMP_FP8_HEAD", "0") != "0"
        # GELU of the MLP / FFN blocks (timm Mlp.act, HF BertIntermediate): 2 = the fc1 epilogue saves gelu'(pre-activation) instead of the
        # pre-activation and the fc2 data gradient multiplies by it (no erf / exp in the backward pass; one more bf16 rounding of the
        # derivative -- a departure from the reference's GeluBackward, which recomputes gelu' in f32 from the pre-activation; drift over 200
        # steps 2.3e-5, profiles/r04_gelu_saved_grad_drift.json); 1 = save the pre-activation and recompute gelu' in the backward epilogue.
        # bf16 mode only.  A constructor argument (`gelu_saved_grad`, main_pretrain.py --gelu_saved_grad, recorded in config.yaml); None =
        # the environment's ECAMP_GELU_SAVED_GRAD, default on.
        if gelu_saved_grad is None:
            gelu_saved_grad = os.environ.get("ECAMP_GELU_SAVED_GRAD", "1") != "0"
        self.gelu_saved_grad = bool(gelu_saved_grad) and compute_dtype in (torch.bfloat16, torch.float16)
        self.gelu_act = 2 if self.gelu_saved_grad else 1
        self.bert_config = bert_config if bert_config is not None else BertConfig()
        # image encoder (model_ecamp.py:58-69)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_patches = num_patches = self.patch_embed.num_patches
        grid = img_size // patch_size
        self.sr_window = sr_window if sr_window is not None else (12 * grid) // 14  # 12 of 14 super-patches (:208)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # image decoder + SR (model_ecamp.py:72-94)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                             for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.super_res = InterpolateConvSuperResolution(scale_factor=2, in_channels=3, out_channels=3, kernel_size=3, stride=1, padding=1)
        # report side (model_ecamp.py:96-100)
        self.bert_encoder = MultiModalBertEncoder(self.bert_config)
        self.bert_mlp = nn.Linear(embed_dim, self.bert_config.hidden_size, bias=True)
        self.norm_pix_loss = norm_pix_loss  # parsed, stored, never used -- exactly like the reference (:100)
        self.initialize_weights()
        # runtime state (not parameters)
        self.arena = None
        self.keep_aux = False
        self._aux = None
        self._aux_logits = None
        self._rng_seed = None
        self._rng_ctr = 0
        self._rng_trace = None
        self._norm_cache = {}
        self._register_load_state_dict_pre_hook(self._alias_old_keys)
        self.register_load_state_dict_post_hook(lambda mod, inc: mod.arena.sync_shadow() if mod.arena is not None else None)

    # ---------------------------------------------------------------------------------------------
    def initialize_weights(self):
        """model_ecamp.py:105-135 plus HF's embedding init (normal(0, initializer_range), PAD row zero)."""
        g = int(self.num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g, cls_token=True)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g, cls_token=True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=.02)
        torch.nn.init.normal_(self.mask_token, std=.02)
        for mod in self.modules():
            if isinstance(mod, nn.Embedding):
                mod.weight.data.normal_(mean=0.0, std=self.bert_config.initializer_range)
                if mod.padding_idx is not None:
                    mod.weight.data[mod.padding_idx].zero_()
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @staticmethod
    def _alias_old_keys(state_dict, prefix, *args):
        """Old checkpoints name the fusion layer `cross_attn_layer` (Visualization/main_visualization.py:91-92)."""
        for k in list(state_dict.keys()):
            if ".bert.cross_attn_layer." in k:
                state_dict[k.replace(".bert.cross_attn_layer.", ".bert.context_fusion_layer.")] = state_dict.pop(k)

    # ---------------------------------------------------------------------------------------------
    def arena_fuse_groups(self):
        groups = []
        for mod in self.modules():
            if hasattr(mod, "query") and hasattr(mod, "key") and hasattr(mod, "value"):
                groups.append([mod.query.weight, mod.key.weight, mod.value.weight])
                groups.append([mod.query.bias, mod.key.bias, mod.value.bias])
        return groups

    def _apply(self, fn, *a, **k):
        self.arena = None  # .to()/.float() re-materialise parameters: the arena is rebuilt lazily
        return super()._apply(fn, *a, **k)

    def prepare(self):
        """Move the parameters into the flat HBM arenas (idempotent).  Called lazily by forward() and by every public stage method, so it
        is also where the process is switched to the build of this model's 16-bit format (bf16: libecamp_hip.so; torch.float16, the
        reference's autocast format: libecamp_hip_f16.so) -- backward and the optimizer run under the same setting."""
        from .. import _lib
        _lib.set_half(self.compute_dtype)
        if self.arena is None:
            from ..arena import ParamArena
            _lib.load()
            self.arena = ParamArena(self, self.compute_dtype)
            self._rng_seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF
        else:
            self.arena.attach_grads()
        return self.arena

    def sync_params(self):
        """Call after editing parameters by hand (the bf16 shadow the GEMMs read is refreshed)."""
        if self.arena is not None:
            self.arena.sync_shadow()

    def next_rng(self):
        """(seed, offset) of the next Philox stream (masking noise, or one dropout site).  `_rng_trace` (a list, tests only) records the
        pairs in call order: the oracle replays the same masks through the development ABI `ecamp_dropout_mask`."""
        self._rng_ctr += 1
        if self._rng_trace is not None:
            self._rng_trace.append((self._rng_seed, self._rng_ctr))
        return self._rng_seed, self._rng_ctr

    def _loss_norm(self, n1, n2, dev):
        key = (n1, n2, str(dev))
        if key not in self._norm_cache:
            self._norm_cache[key] = torch.tensor([1.0 / n1, 1.0 / n2], dtype=torch.float32, device=dev)
        return self._norm_cache[key]

    # ---------------------------------------------------------------------------------------------
    # --- the reference's stage-wise public methods (model_ecamp.py:138-300).  `forward` below runs the same kernels with the
    # stages fused across these boundaries (no materialised masks, no unpatchified copy); these exist so that code written
    # against the reference's methods keeps working, and they are differentiable through the same hand-written backwards.
    def patchify(self, imgs):
        """imgs (N, 3, H, W) -> (N, L, (2p)**2 * 3): the reference patchifies with the SUPER-resolution patch (model_ecamp.py:138-150)."""
        p = self.patch_size * 2
        if imgs.shape[2] != imgs.shape[3] or imgs.shape[2] % p != 0:
            raise AssertionError("patchify: square image with a side divisible by %d expected, got %s" % (p, tuple(imgs.shape)))
        h = imgs.shape[2] // p
        return imgs.reshape(imgs.shape[0], 3, h, p, h, p).permute(0, 2, 4, 3, 5, 1).reshape(imgs.shape[0], h * h, p * p * 3)

    def unpatchify(self, x):
        """x (N, L, p*p*3) -> (N, 3, H, W)  (model_ecamp.py:153-165)."""
        p = self.patch_size
        h = int(x.shape[1] ** .5)
        if h * h != x.shape[1]:
            raise AssertionError("unpatchify: L=%d is not a square" % x.shape[1])
        return x.reshape(x.shape[0], h, h, p, p, 3).permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], 3, h * p, h * p)

    def random_masking(self, x, mask_ratio, noise=None):
        """x [N, L, D] -> (x_masked [N, len_keep, D], mask [N, L] (0 keep / 1 remove), ids_restore, ids_keep)  (model_ecamp.py:168-193).
        The noise comes from the model's Philox stream (or `noise`), the index sort from the `mask_indices` kernel."""
        from .. import hip_ops as ops
        self.prepare()
        N, L, D = x.shape
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            seed, off = self.next_rng()
            noise = ops.uniform((N, L), x.device, seed, off)
        ids_restore, ids_keep, mask = ops.mask_indices(noise.to(x.device, torch.float32).contiguous(), len_keep)
        ids_restore, ids_keep = ids_restore.long(), ids_keep.long()        # the kernels keep int32; the reference's API is int64
        return torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D)), mask, ids_restore, ids_keep

    def mask_2_pixel(self, mask, column, row):
        """mask [N, L] -> (pixel_mask [N,3,R,R], super_pixel_mask [N,3,2R,2R])  (model_ecamp.py:196-215; the first grid axis is
        the one the reference calls `column`).  Only for callers that want the masks: `forward_loss` never materialises them."""
        p, g, w = self.patch_size, int(mask.shape[1] ** .5), self.sr_window
        m2 = mask.reshape(mask.shape[0], g, g)
        idx = torch.arange(g, device=mask.device)
        c, r = column.view(-1, 1).to(mask.device), row.view(-1, 1).to(mask.device)
        sup = (((idx >= c) & (idx < c + w))[:, :, None] & ((idx >= r) & (idx < r + w))[:, None, :]).to(torch.float32)
        up = lambda t, k: t.repeat_interleave(k, 1).repeat_interleave(k, 2).unsqueeze(1).repeat(1, 3, 1, 1)
        return up(m2, p), up(sup, 2 * p)

    def image_encoder(self, x, mask_ratio, noise=None):
        """x [N,3,R,R] -> (latent [N, 1+len_keep, D], mask, ids_restore, ids_keep)  (model_ecamp.py:218-237)."""
        from ..functions import NormFn, StemFn, VitBlockFn
        A = self.prepare()
        x = x.to(A.device, dtype=torch.float32).contiguous()
        if x.shape[1:] != (3, self.img_size, self.img_size):
            raise ValueError("image_encoder expects [N,3,%d,%d], got %s" % (self.img_size, self.img_size, tuple(x.shape)))
        if noise is not None:
            noise = noise.to(A.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        t, _, mask, ids_restore, ids_keep = StemFn.apply(x, noise, self, mask_ratio, self.cls_token)
        T = ids_keep.shape[1] + 1
        for blk in self.blocks:
            t = VitBlockFn.apply(t, blk, self, B, T, self.num_heads)
        return NormFn.apply(t, self.norm, self).view(B, T, -1), mask, ids_restore.long(), ids_keep.long()   # int64 as in the reference

    def image_decoder(self, x, ids_restore):
        """latent [N, 1+len_keep, D], ids_restore [N, L] -> pred [N, L, p*p*3] (a view without the cls row)  (model_ecamp.py:240-264)."""
        from ..functions import DecHeadFn, DecStemFn, VitBlockFn
        A = self.prepare()
        B, T, D = x.shape
        L = self.num_patches
        ids_keep = torch.argsort(ids_restore.to(A.device), dim=1)[:, :T - 1].to(torch.int32).contiguous()   # ids_shuffle = inverse permutation
        ids_restore = ids_restore.to(A.device, torch.int32).contiguous()       # the index kernels read int32
        xd = DecStemFn.apply(x.to(self.compute_dtype).reshape(B * T, D), ids_restore, ids_keep, self, B)
        for blk in self.decoder_blocks:
            xd = VitBlockFn.apply(xd, blk, self, B, L + 1, self.decoder_num_heads)
        return DecHeadFn.apply(xd, self).view(B, L + 1, -1)[:, 1:, :]

    def forward_loss(self, imgs, big_imgs, pred, mask, column, row):
        """imgs [N,3,R,R], big_imgs [N,3,2R,2R], pred [N,L,p*p*3], mask [N,L] (1 = removed), column/row [N] -> (mim_loss, res_loss)
        (model_ecamp.py:276-300): unpatchify + masked MSE and SR head + windowed MSE in two fused kernels."""
        from ..functions import PixelLossFn
        A = self.prepare()
        dev = A.device
        B, L, PD = pred.shape
        full = torch.cat([pred.new_zeros(B, 1, PD), pred], 1).to(self.compute_dtype).reshape(B * (L + 1), PD)   # the kernels skip row 0
        f32 = lambda t: t.to(dev, dtype=torch.float32).contiguous()
        i64 = lambda t: t.to(dev, dtype=torch.int64).contiguous().view(-1)
        out = PixelLossFn.apply(full, f32(imgs), f32(big_imgs), f32(mask), i64(column), i64(row), self, B)
        return out[0], out[1]

    def forward_report_decoder(self, latent, ids_keep, caption_ids, labels, attention_mask, token_type_ids, weights, B=None, T=None):
        """model_ecamp.py:267-273 (`ids_keep` is unused there too).  `latent` is [B, T, D] as in the reference, or [B*T, D] with B, T."""
        from ..functions import ReportStemFn
        if latent.dim() == 3:
            B, T = latent.shape[:2]
            latent = latent.reshape(B * T, -1)
            dev = latent.device
            mv = lambda t, dt: t.to(dev, dtype=dt).contiguous()
            caption_ids, labels, attention_mask, token_type_ids = (mv(t, torch.int64) for t in (caption_ids, labels, attention_mask, token_type_ids))
            weights = mv(weights, torch.float32)
        lat, gap = ReportStemFn.apply(latent, self, B, T)
        out = self.bert_encoder(lat, gap, caption_ids, labels, attention_mask, token_type_ids, weights, self, B, T)
        return out.loss

    def forward(self, batch, mask_ratio=0.75, noise=None, image_side_only=False):
        """batch: dict with the schema of pretrain_datasets.py:228-237 (CPU or device tensors).
        noise: optional [B, L] masking noise standing in for torch.rand at model_ecamp.py:177 (parity tests).
        image_side_only (measurement aid, bench.py `vit_*`): stem -> encoder -> decoder -> image losses (model_ecamp.py:218-264,276-300),
        i.e. the "ViT-B/16 forward+backward" the north-star target is quoted on, without the report side -> (mim_loss, res_loss, None)."""
        from ..functions import DecStemFn, ImgLossFn, NormFn, StemFn, VitBlockFn
        A = self.prepare()
        dev = A.device
        if "image" not in batch and "image_crops" in batch:
            # device image pipeline (pretrain_datasets.DeviceAugmenter, csrc/augment.hip): the loader handed over the BYTES of each sample's
            # crop box + a table; RandomResizedCrop's resize, the flip and Grayscale run here, Pillow-exact -> the uint8 schema below
            if getattr(self, "_augmenter", None) is None or self._augmenter.device != dev or self._augmenter.size != 2 * self.img_size:
                from .pretrain_datasets import DeviceAugmenter
                self._augmenter = DeviceAugmenter(dev, size=2 * self.img_size)
            img = self._augmenter(batch["image_crops"], batch["image_table"], meta=batch.get("image_meta"))
        else:
            img = batch["image"]
        if img.dtype == torch.uint8:
            # compact schema (ecamp_amd.data / ContextBertDataset(image_u8=True)): the grayscale crop itself, [B, 2R, 2R] or [B, 1, 2R, 2R];
            # the bicubic and SR-loss kernels normalise it on the fly -- same bits as the f32 [B, 3, 2R, 2R] image, a twelfth of the bytes
            big = img.to(dev, non_blocking=True).reshape(img.shape[0], img.shape[-2], img.shape[-1]).contiguous()
        else:
            big = img.to(dev, dtype=torch.float32, non_blocking=True).contiguous()
        mv = lambda t, dt: t.to(dev, dtype=dt, non_blocking=True).contiguous()
        ids, labels = mv(batch["ids"], torch.int64), mv(batch["labels"], torch.int64)
        attention_mask, type_ids = mv(batch["attention_mask"], torch.int64), mv(batch["type_ids"], torch.int64)
        weights = mv(batch["weights"], torch.float32)
        column, row = mv(batch["column"], torch.int64).view(-1), mv(batch["row"], torch.int64).view(-1)
        if ids.dim() == 1:  # the reference's collate_fn .squeeze()s a batch of one (pretrain_datasets.py:218-225)
            ids, labels, attention_mask, type_ids, weights = (t.unsqueeze(0) for t in (ids, labels, attention_mask, type_ids, weights))
        B = big.shape[0]
        want = (2 * self.img_size, 2 * self.img_size) if big.dtype == torch.uint8 else (3, 2 * self.img_size, 2 * self.img_size)
        if tuple(big.shape[1:]) != want:
            raise ValueError("image must be f32 [B,3,%d,%d] or uint8 [B,%d,%d] (2x the encoder resolution), got %s %s"
                             % (2 * self.img_size, 2 * self.img_size, 2 * self.img_size, 2 * self.img_size, big.dtype, tuple(img.shape)))
        if noise is not None:
            noise = noise.to(dev, dtype=torch.float32).contiguous()

        x, imgs, mask, ids_restore, ids_keep = StemFn.apply(big, noise, self, mask_ratio, self.cls_token)   # global feature
        T = ids_keep.shape[1] + 1
        for blk in self.blocks:
            x = VitBlockFn.apply(x, blk, self, B, T, self.num_heads)
        latent = NormFn.apply(x, self.norm, self)
        def image_decoder():
            xd = DecStemFn.apply(latent, ids_restore, ids_keep, self, B)
            for blk in self.decoder_blocks:
                xd = VitBlockFn.apply(xd, blk, self, B, self.num_patches + 1, self.decoder_num_heads)
            return ImgLossFn.apply(xd, imgs, big, mask, column, row, self, B)

        from .. import hip_ops as ops
        if image_side_only:
            img_losses = image_decoder()
            return img_losses[0], img_losses[1], None
        if ops.OVERLAP_BRANCHES and latent.is_cuda:
            # the two consumers of `latent` on two streams (see hip_ops.branch_stream)
            main, bs = torch.cuda.current_stream(dev), ops.branch_stream(dev)
            bs.wait_stream(main)
            with torch.cuda.stream(bs):
                img_losses = image_decoder()
            for t in (latent, imgs, big, mask, ids_restore, ids_keep, column, row):
                t.record_stream(bs)   # (not hip_ops.hold: the branch's BACKWARD nodes read these again on `bs`, later than any fence taken here)
            mlm_loss = self.forward_report_decoder(latent, ids_keep, ids, labels, attention_mask, type_ids, weights, B, T)
            main.wait_stream(bs)
            for t in img_losses:   # three scalars allocated on the branch stream and read by the CALLER on the main one, later than any point this
                t.record_stream(main)   # function could fence: the allocator's own bookkeeping (no effect on its steady state at this size)
        else:
            img_losses = image_decoder()
            mlm_loss = self.forward_report_decoder(latent, ids_keep, ids, labels, attention_mask, type_ids, weights, B, T)
        if self.keep_aux:
            self._aux = dict(self._aux or {}, imgs=imgs, mask=mask, ids_restore=ids_restore, ids_keep=ids_keep,
                             latent=latent.view(B, T, -1), logits=self._aux_logits, **(getattr(self, "_aux_text", None) or {}))
        return img_losses[0], img_losses[1], mlm_loss


    @torch.no_grad()
    def forward_visualization(self, imgs, text_ids, attention_mask, type_ids, mask_ratio=0, noise=None):
        """The reference's Visualization model (Visualization/module/model_ecamp.py:308-319): encoder on the 224^2 image with
        `mask_ratio` (0: every patch kept, in argsort(noise) order as the reference does), then the fusion layer's
        cross-attention probabilities of the report tokens onto the image tokens (Visualization/module/context_fusion.py:45-57)
        -> f32 [B, heads, S, L_keep].  Evaluation semantics (no dropout) whatever `self.training` is."""
        import math

        from .. import hip_ops as ops
        from ..functions import BertEmbedFn, NormFn, ReportStemFn, StemFn, VitBlockFn, _self_attn_fwd
        A = self.prepare()
        dev = A.device
        imgs = imgs.to(dev, dtype=torch.float32, non_blocking=True).contiguous()
        if imgs.shape[1:] != (3, self.img_size, self.img_size):
            raise ValueError("imgs must be [B,3,%d,%d], got %s" % (self.img_size, self.img_size, tuple(imgs.shape)))
        mv = lambda t: t.to(dev, dtype=torch.int64, non_blocking=True).contiguous()
        ids, attention_mask, type_ids = mv(text_ids), mv(attention_mask), mv(type_ids)
        if ids.dim() == 1:
            ids, attention_mask, type_ids = ids[None], attention_mask[None], type_ids[None]
        B, S = ids.shape
        if noise is not None:
            noise = noise.to(dev, dtype=torch.float32).contiguous()
        x, _, _, _, ids_keep = StemFn.apply(imgs, noise, self, mask_ratio, self.cls_token)
        T = ids_keep.shape[1] + 1
        for blk in self.blocks:
            x = VitBlockFn.apply(x, blk, self, B, T, self.num_heads)
        latent = NormFn.apply(x, self.norm, self)
        lat, _ = ReportStemFn.apply(latent, self, B, T)
        bert = self.bert_encoder.model.bert
        e = BertEmbedFn.apply(ids, type_ids, bert.embeddings, self, 0.0, bert.embeddings.LayerNorm.weight)
        fl = bert.context_fusion_layer
        key_mask = attention_mask.to(torch.int32).contiguous()
        a1 = _self_attn_fwd(self, fl.attention.self, fl.attention.output, e, B, S, key_mask, 0.0, 0.0, [])
        ca = fl.cross_self_attention
        H = a1.shape[1]
        heads = self.bert_config.num_attention_heads
        hd = H // heads
        q = ops.linear_fwd(a1, A.w(ca.query.weight), ca.query.bias.data)
        k = ops.linear_fwd(lat, A.w(ca.key.weight), ca.key.bias.data)   # [B*T, H]; the cls row is skipped by a pointer offset
        return ops.attn_probs(q, k.view(-1)[H:], B, heads, S, T - 1, hd, (S * H, H, hd), (T * H, H, hd), 1.0 / math.sqrt(hd))


def ecamp(**kwargs):
    """ViT-B/16 encoder + 512-d/4-block decoder + reference BERT -- model_ecamp.py:328-333."""
    return ECAMP(patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=4,
                 decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def ecamp_tiny(**kwargs):
    """BASELINE.json configs[0]: ViT-Tiny/16 (D=192, 3 heads) + 2-layer BERT; decoder unchanged."""
    kwargs.setdefault("bert_config", BertConfig(num_hidden_layers=2))
    return ECAMP(patch_size=16, in_chans=3, embed_dim=192, depth=12, num_heads=3, decoder_embed_dim=512, decoder_depth=4,
                 decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def ecamp_large_448(**kwargs):
    """BASELINE.json configs[3]: ViT-L/16 at 448^2 encoder input (decoder sequence 785)."""
    return ECAMP(img_size=448, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                 decoder_depth=4, decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
