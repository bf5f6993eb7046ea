"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ecamp_amd`.

Shim that lets the *reference's own* pre-training code (`/root/reference/ECAMP/Pre-training`)
be imported and run on CPU in the authoring container, whose library versions differ from the
reference's pins (environment.yml:128-138: torch 1.13.1, timm 0.4.12, transformers 4.42.4,
torchvision 0.14.1).  It is used only by `oracle/make_golden.py` to generate the golden vectors
committed under `tests/golden/` and to validate `oracle/ecamp_oracle.py`.  Nothing here travels
to the GPU box as a dependency: `/root/reference` does not exist there and this module raises.

What is stubbed, and why:
  * `torch._six` (removed), `ipdb` (absent), `np.float` (removed in numpy 2)      -> trivial aliases
  * `torchvision.transforms.Resize` (absent)  -> F.interpolate(bicubic, align_corners=False,
    antialias=False), which is what torchvision 0.14.1 dispatches to for float tensors
  * `timm.models.vision_transformer.{PatchEmbed,Block}` (absent) -> restatement of timm 0.4.12
    (third-party, not vendored in the reference; call sites model_ecamp.py:19,60,66-68,80-82)
  * transformers 5.x vs 4.42.4 API drift: legacy `BertSelfAttention` (4.42.4 signature incl.
    the cross-attention mode used at context_fusion.py:45-53), `get_extended_attention_mask`,
    `get_head_mask`, `apply_chunking_to_forward` location, config attribute defaults, and the
    4.42.4 tying of `cls.predictions.decoder.bias` to `cls.predictions.bias`.
The arithmetic of BertEmbeddings / BertSelfOutput / BertIntermediate / BertOutput / BertLayer /
BertEncoder / BertPooler / BertOnlyMLMHead comes from the *installed* transformers (unchanged
arithmetic between 4.42.4 and 5.x).
"""
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference/ECAMP/Pre-training"


def reference_available():
    return os.path.isdir(REF_ROOT)


# ----------------------------------------------------------------------------------------------
# timm 0.4.12 restatement (vision_transformer.py: Mlp, Attention, Block; layers/patch_embed.py)
# ----------------------------------------------------------------------------------------------
class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class _Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        attn = self.attn_drop(attn)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        x = self.proj(x)
        return self.proj_drop(x)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x


class _PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size = (img_size, img_size)
        patch_size = (patch_size, patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1]
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


def _add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """timm 0.4.12 optim_factory.add_weight_decay (call site main_pretrain.py:253)."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


# ----------------------------------------------------------------------------------------------
# transformers 4.42.4 BertSelfAttention (eager, absolute positions) restatement
# ----------------------------------------------------------------------------------------------
class _LegacyBertSelfAttention(nn.Module):
    def __init__(self, config, position_embedding_type=None, **_ignored):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def _split(self, x):
        return x.view(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, **_ignored):
        q = self._split(self.query(hidden_states))
        if encoder_hidden_states is not None:
            k = self._split(self.key(encoder_hidden_states))
            v = self._split(self.value(encoder_hidden_states))
            attention_mask = encoder_attention_mask
        else:
            k = self._split(self.key(hidden_states))
            v = self._split(self.value(hidden_states))
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.attention_head_size)
        if attention_mask is not None:
            scores = scores + attention_mask
        probs = self.dropout(F.softmax(scores, dim=-1))
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
        ctx = ctx.view(ctx.size()[:-2] + (self.all_head_size,))
        return (ctx, probs)


_INSTALLED = False


def install():
    """Install all stubs/patches into sys.modules. Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError("reference checkout not present at %s (expected on the GPU box)" % REF_ROOT)

    if not hasattr(np, "float"):
        np.float = float  # util/pos_embed.py:56
    six = types.ModuleType("torch._six")
    six.inf = math.inf
    sys.modules.setdefault("torch._six", six)  # util/misc.py:21
    sys.modules.setdefault("ipdb", types.ModuleType("ipdb"))  # model_ecamp.py:25

    # import transformers BEFORE the torchvision stub exists (its availability probe chokes on a spec-less module)
    import transformers  # noqa: F401
    import transformers.models.bert.modeling_bert  # noqa: F401

    # torchvision.transforms.Resize + InterpolationMode (model_ecamp.py:15,18,318)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvf = types.ModuleType("torchvision.transforms.functional")

        class InterpolationMode:
            BICUBIC = "bicubic"
            BILINEAR = "bilinear"

        class Resize:
            def __init__(self, size, interpolation="bilinear"):
                self.size, self.mode = list(size), interpolation

            def __call__(self, x):
                return F.interpolate(x, size=self.size, mode=self.mode, align_corners=False, antialias=False)

        tvf.InterpolationMode = InterpolationMode
        tvt.Resize = Resize
        tvt.functional = tvf
        tvt.InterpolationMode = InterpolationMode
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
        sys.modules["torchvision.transforms.functional"] = tvf

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm.__version__ = "0.4.12"
        models = types.ModuleType("timm.models")
        vt = types.ModuleType("timm.models.vision_transformer")
        vt.PatchEmbed, vt.Block = _PatchEmbed, _Block
        optim = types.ModuleType("timm.optim")
        of = types.ModuleType("timm.optim.optim_factory")
        of.add_weight_decay = _add_weight_decay
        optim.optim_factory = of
        models.vision_transformer = vt
        timm.models, timm.optim = models, optim
        for k, m in [("timm", timm), ("timm.models", models), ("timm.models.vision_transformer", vt),
                     ("timm.optim", optim), ("timm.optim.optim_factory", of)]:
            sys.modules[k] = m

    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    import transformers.models.bert.modeling_bert as mb

    if not hasattr(mu, "apply_chunking_to_forward"):
        mu.apply_chunking_to_forward = pu.apply_chunking_to_forward  # context_fusion.py:4
    mb.BertSelfAttention = _LegacyBertSelfAttention  # used by context_fusion.py:3,15 and BertAttention

    def _ext_mask(self, attention_mask, input_shape=None, device=None, dtype=None):
        # transformers 4.42.4 modeling_utils.get_extended_attention_mask for a 2-D mask, fp32 model
        m = attention_mask[:, None, None, :].to(torch.float32)
        return (1.0 - m) * torch.finfo(torch.float32).min

    mu.PreTrainedModel.get_extended_attention_mask = _ext_mask
    mu.PreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n

    sys.path.insert(0, REF_ROOT)
    import module.bert_config as bc

    for k, v in dict(is_decoder=False, add_cross_attention=False, chunk_size_feed_forward=0,
                     tie_word_embeddings=True, output_attentions=False, output_hidden_states=False,
                     use_return_dict=True, return_dict=True).items():
        if not hasattr(bc.BertConfig, k):
            setattr(bc.BertConfig, k, v)

    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # model_ecamp.py:211-212,312-317
    _INSTALLED = True


def build_reference_model(tiny=False, num_bert_layers=None, **kw):
    """Construct the reference's ECAMP. `tiny` = ViT-Tiny/16 (D=192, 3 heads) + 2-layer BERT
    (BASELINE.json configs[0]); the reference has no factory for it so it is built from `ECAMP(...)`."""
    install()
    from functools import partial
    import module.bert_config as bc
    import module.bert_encoder as be
    import module.model_ecamp as me

    nl = num_bert_layers if num_bert_layers is not None else (2 if tiny else None)
    orig = be.BertConfig
    if nl is not None:
        be.BertConfig = lambda: bc.BertConfig(num_hidden_layers=nl)
    try:
        if tiny:
            model = me.ECAMP(patch_size=16, in_chans=3, embed_dim=192, depth=12, num_heads=3,
                             decoder_embed_dim=512, decoder_depth=4, decoder_num_heads=16, mlp_ratio=4,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)
        else:
            model = me.ecamp(**kw)
    finally:
        be.BertConfig = orig
    # transformers 4.42.4 BertLMPredictionHead ties decoder.bias to predictions.bias
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
