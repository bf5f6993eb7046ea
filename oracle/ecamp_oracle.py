"""TEST INFRASTRUCTURE ONLY -- the parity oracle.  Never imported by the product (`ecamp_amd`).

A CPU restatement, in plain fp32 PyTorch, of the arithmetic on ECAMP's pre-training hot path
(SURVEY.md section 8a rows a1-a22).  It is written *functionally* over a flat `{state_dict key ->
tensor}` dict so that it shares no structure with the reference's nn.Module tree; every function
cites the reference lines (relative to /root/reference/ECAMP/Pre-training = `PT/`) it restates.

Parity status: PINNED.  `oracle/make_golden.py` runs the reference's own `ECAMP.forward`/backward
(through `oracle/ref_shim.py`) in this container and checks this file against it to <=2e-6
relative on losses / activations / gradients; the resulting vectors are committed under
`tests/golden/*.npz` and re-checked by `tests/test_oracle_golden.py` on every run (CPU) and used
by the `-m gpu` parity tests as the expected outputs of the HIP path.

Third-party arithmetic not vendored in the reference is restated from the pinned versions:
timm==0.4.12 (PatchEmbed, Block, add_weight_decay), transformers==4.42.4 (Bert*),
torchvision==0.14.1 (Resize), torch==1.13.1 (AdamW, GradScaler) -- environment.yml:128-138.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# configuration
# ---------------------------------------------------------------------------------------------
@dataclass
class BertCfg:  # PT/module/bert_config.py:63-94
    vocab_size: int = 30000
    hidden_size: int = 768
    num_hidden_layers: int = 6
    num_attention_heads: int = 6
    intermediate_size: int = 1536
    max_position_embeddings: int = 256
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1


@dataclass
class Cfg:  # PT/module/model_ecamp.py:52-55 defaults overridden by ecamp():328-333
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 4
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    ln_eps: float = 1e-6
    sr_window: int = 12  # model_ecamp.py:208
    bert: BertCfg = field(default_factory=BertCfg)

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_patches(self):
        return self.grid * self.grid


def cfg_base():
    return Cfg()


def cfg_tiny():
    """BASELINE.json configs[0]: ViT-Tiny/16 encoder + 2-layer BERT (decoder unchanged)."""
    return Cfg(embed_dim=192, num_heads=3, bert=BertCfg(num_hidden_layers=2))


def cfg_large448():
    """BASELINE.json configs[3]: ViT-L/16 at 448^2 encoder input."""
    return Cfg(img_size=448, embed_dim=1024, depth=24, num_heads=16, sr_window=24)


# ---------------------------------------------------------------------------------------------
# parameter inventory (SURVEY.md 8b state-dict keys)
# ---------------------------------------------------------------------------------------------
def param_shapes(cfg):
    """name -> (shape, trainable) in the reference's registration order."""
    D, Dd, Hb = cfg.embed_dim, cfg.decoder_embed_dim, cfg.bert.hidden_size
    L = cfg.num_patches
    p, c = cfg.patch_size, cfg.in_chans
    s = OrderedDict()

    def lin(prefix, out_f, in_f):
        s[prefix + ".weight"] = ((out_f, in_f), True)
        s[prefix + ".bias"] = ((out_f,), True)

    def ln(prefix, d):
        s[prefix + ".weight"] = ((d,), True)
        s[prefix + ".bias"] = ((d,), True)

    def vit_block(prefix, d):
        hid = int(d * cfg.mlp_ratio)
        ln(prefix + ".norm1", d)
        lin(prefix + ".attn.qkv", 3 * d, d)
        lin(prefix + ".attn.proj", d, d)
        ln(prefix + ".norm2", d)
        lin(prefix + ".mlp.fc1", hid, d)
        lin(prefix + ".mlp.fc2", d, hid)

    s["cls_token"] = ((1, 1, D), True)
    s["pos_embed"] = ((1, L + 1, D), False)
    s["mask_token"] = ((1, 1, Dd), True)
    s["decoder_pos_embed"] = ((1, L + 1, Dd), False)
    s["patch_embed.proj.weight"] = ((D, c, p, p), True)
    s["patch_embed.proj.bias"] = ((D,), True)
    for i in range(cfg.depth):
        vit_block("blocks.%d" % i, D)
    ln("norm", D)
    lin("decoder_embed", Dd, D)
    for i in range(cfg.decoder_depth):
        vit_block("decoder_blocks.%d" % i, Dd)
    ln("decoder_norm", Dd)
    lin("decoder_pred", p * p * c, Dd)
    for n in ("conv1", "conv2"):
        s["super_res.%s.weight" % n] = ((3, 3, 3, 3), True)
        s["super_res.%s.bias" % n] = ((3,), True)

    b = cfg.bert
    pre = "bert_encoder.model.bert."
    s[pre + "embeddings.word_embeddings.weight"] = ((b.vocab_size, Hb), True)
    s[pre + "embeddings.position_embeddings.weight"] = ((b.max_position_embeddings, Hb), True)
    s[pre + "embeddings.token_type_embeddings.weight"] = ((b.type_vocab_size, Hb), True)
    ln(pre + "embeddings.LayerNorm", Hb)

    def bert_attn(prefix):
        for n in ("query", "key", "value"):
            lin(prefix + ".self." + n, Hb, Hb)
        lin(prefix + ".output.dense", Hb, Hb)
        ln(prefix + ".output.LayerNorm", Hb)

    def bert_ffn(prefix):
        lin(prefix + ".intermediate.dense", b.intermediate_size, Hb)
        lin(prefix + ".output.dense", Hb, b.intermediate_size)
        ln(prefix + ".output.LayerNorm", Hb)

    for i in range(b.num_hidden_layers):
        bert_attn(pre + "encoder.layer.%d.attention" % i)
        bert_ffn(pre + "encoder.layer.%d" % i)
    lin(pre + "pooler.dense", Hb, Hb)
    f = pre + "context_fusion_layer"
    bert_attn(f + ".attention")
    for n in ("query", "key", "value"):
        lin(f + ".cross_self_attention." + n, Hb, Hb)
    bert_ffn(f)
    lin(f + ".gap_mlp", Hb, Hb)
    lin(f + ".out_layer.dense", Hb, Hb)
    ln(f + ".out_layer.LayerNorm", Hb)
    cls = "bert_encoder.model.cls.predictions."
    s[cls + "bias"] = ((b.vocab_size,), True)
    lin(cls + "transform.dense", Hb, Hb)
    ln(cls + "transform.LayerNorm", Hb)
    s[cls + "decoder.weight"] = ((b.vocab_size, Hb), True)
    # transformers 4.42.4: `decoder.bias` IS `predictions.bias` (same Parameter, serialised twice)
    s[cls + "decoder.bias"] = ((b.vocab_size,), True)
    lin("bert_mlp", Hb, D)
    return s


TIED = {"bert_encoder.model.cls.predictions.decoder.bias": "bert_encoder.model.cls.predictions.bias"}
UNUSED = ("bert_encoder.model.bert.pooler.dense.weight", "bert_encoder.model.bert.pooler.dense.bias")


def trainable_names(cfg):
    """Names as `named_parameters()` of the reference would yield them (tied alias removed)."""
    return [k for k, (_, t) in param_shapes(cfg).items() if t and k not in TIED]


# ---------------------------------------------------------------------------------------------
# a22  fixed 2-D sin-cos table   PT/util/pos_embed.py:20-67
# ---------------------------------------------------------------------------------------------
def sincos_2d(embed_dim, grid_size, cls_token=True):
    def one_d(dim, pos):  # pos_embed.py:49-67
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)  # "w first" :30
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)  # :43-46
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return torch.from_numpy(emb).float().unsqueeze(0)


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def _ln(P, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def bicubic_resize(big, size):
    """a7: torchvision 0.14.1 Resize(BICUBIC) on a float tensor == aten upsample_bicubic2d,
    A=-0.75, align_corners=False, no antialias (model_ecamp.py:318)."""
    return F.interpolate(big, size=[size, size], mode="bicubic", align_corners=False, antialias=False)


def random_masking(x, mask_ratio, noise):
    """a9: model_ecamp.py:168-193.  `noise` stands in for torch.rand(N, L) at :177."""
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, :len_keep]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    mask = torch.ones(N, L, dtype=x.dtype)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return x_masked, mask, ids_restore, ids_keep


def vit_block(P, pre, x, heads, eps):
    """a11: timm 0.4.12 Block/Attention/Mlp (pre-LN, fused qkv, exact-erf GELU, no dropout)."""
    B, T, D = x.shape
    hd = D // heads
    h = _ln(P, pre + ".norm1", x, eps)
    qkv = _lin(P, pre + ".attn.qkv", h).reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    a = (att @ v).transpose(1, 2).reshape(B, T, D)
    x = x + _lin(P, pre + ".attn.proj", a)
    h = _ln(P, pre + ".norm2", x, eps)
    x = x + _lin(P, pre + ".mlp.fc2", F.gelu(_lin(P, pre + ".mlp.fc1", h)))
    return x


def image_encoder(P, cfg, imgs, mask_ratio, noise):
    """a8-a12: model_ecamp.py:218-237."""
    w = P["patch_embed.proj.weight"]
    x = F.conv2d(imgs, w, P["patch_embed.proj.bias"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    x = x + P["pos_embed"][:, 1:, :]
    x, mask, ids_restore, ids_keep = random_masking(x, mask_ratio, noise)
    cls = (P["cls_token"] + P["pos_embed"][:, :1, :]).expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    for i in range(cfg.depth):
        x = vit_block(P, "blocks.%d" % i, x, cfg.num_heads, cfg.ln_eps)
    return _ln(P, "norm", x, cfg.ln_eps), mask, ids_restore, ids_keep


def image_decoder(P, cfg, latent, ids_restore):
    """a13: model_ecamp.py:240-264."""
    x = _lin(P, "decoder_embed", latent)
    B, T, Dd = x.shape
    n_mask = ids_restore.shape[1] + 1 - T
    x_ = torch.cat([x[:, 1:, :], P["mask_token"].expand(B, n_mask, Dd)], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    x = torch.cat([x[:, :1, :], x_], dim=1) + P["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = vit_block(P, "decoder_blocks.%d" % i, x, cfg.decoder_num_heads, cfg.ln_eps)
    x = _lin(P, "decoder_pred", _ln(P, "decoder_norm", x, cfg.ln_eps))
    return x[:, 1:, :]


def unpatchify(cfg, x):
    """a14: model_ecamp.py:153-165 -- (N, L, p*p*3) with inner order (p, q, c) -> (N, 3, H, W)."""
    p = cfg.patch_size
    h = w = int(x.shape[1] ** 0.5)
    x = x.reshape(x.shape[0], h, w, p, p, 3)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], 3, h * p, w * p)


def super_res(P, x):
    """a15: InterpolateConvSuperResolution.forward, model_ecamp.py:37-46."""
    u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.relu(F.conv2d(u, P["super_res.conv1.weight"], P["super_res.conv1.bias"], padding=1))
    y = F.conv2d(y, P["super_res.conv2.weight"], P["super_res.conv2.bias"], padding=1)
    return F.relu(y + u)


def mask_2_pixel(cfg, mask, column, row):
    """a16: model_ecamp.py:196-215.  `column` indexes the H axis of the patch grid, `row` the W axis."""
    p = cfg.patch_size
    g = int(mask.shape[1] ** 0.5)
    m = mask.reshape(mask.shape[0], g, g)
    sm = torch.zeros_like(m)
    for i in range(m.shape[0]):
        c, r = int(column[i]), int(row[i])
        sm[i, c:c + cfg.sr_window, r:r + cfg.sr_window] = 1
    pm = torch.kron(m, torch.ones(p, p))
    spm = torch.kron(sm, torch.ones(2 * p, 2 * p))
    return pm.unsqueeze(1).expand(-1, 3, -1, -1), spm.unsqueeze(1).expand(-1, 3, -1, -1)


def forward_loss(P, cfg, imgs, big_imgs, pred, mask, column, row):
    """a17: model_ecamp.py:276-300 -- both losses are means over ALL pixels of the masked products."""
    pm, spm = mask_2_pixel(cfg, mask, column, row)
    pred_img = unpatchify(cfg, pred)
    sr = super_res(P, pred_img)
    mim = F.mse_loss(pred_img * pm, imgs * pm, reduction="mean")
    res = F.mse_loss(sr * spm, big_imgs * spm, reduction="mean")
    return mim, res, pred_img, sr


# ---- BERT side (transformers 4.42.4 arithmetic) ----------------------------------------------
def _drop(x, p, train):
    """nn.Dropout(p) at the reference's dropout sites (BertEmbeddings bert_modeling.py:113; attention probabilities and the
    BertSelfOutput / BertOutput dense outputs of the fusion layer, context_fusion.py:28-57, and of the six BertLayers,
    bert_modeling.py:131).  `train`: False = eval, True = torch's generator (what the reference draws from), or a callable
    (shape, p) -> 0/1 keep-mask: the masks of ANOTHER implementation replayed in the reference's call order (embeddings; fusion:
    self-attention probabilities, attention output, cross-attention probabilities, out_layer, FFN output; per layer: probabilities,
    attention output, FFN output), so that a train-mode step can be compared to tolerance instead of statistically."""
    if callable(train):
        return x if p == 0.0 else x * train(tuple(x.shape), p) / (1.0 - p)
    return F.dropout(x, p, training=train)


def bert_self_attention(P, pre, hidden, ext_mask, heads, drop_p, train, kv=None, return_probs=False):
    """a19/a20: BertSelfAttention 4.42.4 (self mode, or cross mode when `kv` is given)."""
    B, S, H = hidden.shape
    hd = H // heads
    src = hidden if kv is None else kv

    def split(t):
        return t.view(t.shape[0], t.shape[1], heads, hd).permute(0, 2, 1, 3)

    q = split(_lin(P, pre + ".query", hidden))
    k = split(_lin(P, pre + ".key", src))
    v = split(_lin(P, pre + ".value", src))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    if ext_mask is not None:
        scores = scores + ext_mask
    probs = _drop(F.softmax(scores, dim=-1), drop_p, train)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, S, H)
    return (ctx, probs) if return_probs else ctx


def bert_self_output(P, pre, hidden, residual, eps, drop_p, train):
    """BertSelfOutput / BertOutput: LN(dropout(dense(h)) + residual)."""
    return _ln(P, pre + ".LayerNorm", _drop(_lin(P, pre + ".dense", hidden), drop_p, train) + residual, eps)


def bert_ffn(P, pre, x, eps, drop_p, train):
    inter = F.gelu(_lin(P, pre + ".intermediate.dense", x))
    return bert_self_output(P, pre + ".output", inter, x, eps, drop_p, train)


def fusion_layer(P, pre, b, hidden, img, gap, text_mask, img_mask, train):
    """a20: ECAMPFusionLayer.forward, PT/module/context_fusion.py:21-72."""
    eps, dp, ap = b.layer_norm_eps, b.hidden_dropout_prob, b.attention_probs_dropout_prob
    a = bert_self_attention(P, pre + ".attention.self", hidden, text_mask, b.num_attention_heads, ap, train)
    a = bert_self_output(P, pre + ".attention.output", a, hidden, eps, dp, train)  # :32-39
    c = bert_self_attention(P, pre + ".cross_self_attention", a, img_mask, b.num_attention_heads, ap, train, kv=img)  # :45-53
    c = c + _lin(P, pre + ".gap_mlp", gap)  # :54-55 (broadcast over S)
    a2 = bert_self_output(P, pre + ".out_layer", c, a, eps, dp, train)  # :56
    return bert_ffn(P, pre, a2, eps, dp, train)  # :62-72


def bert_layer(P, pre, b, hidden, text_mask, train):
    eps, dp, ap = b.layer_norm_eps, b.hidden_dropout_prob, b.attention_probs_dropout_prob
    a = bert_self_attention(P, pre + ".attention.self", hidden, text_mask, b.num_attention_heads, ap, train)
    a = bert_self_output(P, pre + ".attention.output", a, hidden, eps, dp, train)
    return bert_ffn(P, pre, a, eps, dp, train)


def report_decoder(P, cfg, latent, ids, labels, attention_mask, type_ids, weights, train=False):
    """a18, a19, a21: model_ecamp.py:267-273 -> bert_modeling.py:15-156,165-227."""
    b = cfg.bert
    lat = _lin(P, "bert_mlp", latent)
    gap = lat[:, 1:, :].mean(dim=1, keepdim=True)
    img = lat[:, 1:, :]
    B, S = ids.shape
    pre = "bert_encoder.model.bert."
    fmin = torch.finfo(torch.float32).min
    text_mask = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * fmin  # bert_modeling.py:92
    img_mask = torch.zeros(B, 1, 1, img.shape[1])  # all-ones image mask :79,93
    pos = torch.arange(S)
    # nn.Embedding(vocab, hidden, padding_idx=pad_token_id=0): the PAD row receives NO gradient
    e = (F.embedding(ids, P[pre + "embeddings.word_embeddings.weight"], padding_idx=0)
         + P[pre + "embeddings.token_type_embeddings.weight"][type_ids]
         + P[pre + "embeddings.position_embeddings.weight"][pos][None])
    e = _drop(_ln(P, pre + "embeddings.LayerNorm", e, b.layer_norm_eps), b.hidden_dropout_prob, train)  # :113
    h = fusion_layer(P, pre + "context_fusion_layer", b, e, img, gap, text_mask, img_mask, train)  # :121
    fused = h
    for i in range(b.num_hidden_layers):
        h = bert_layer(P, pre + "encoder.layer.%d" % i, b, h, text_mask, train)  # :131
    cls = "bert_encoder.model.cls.predictions."
    t = _ln(P, cls + "transform.LayerNorm", F.gelu(_lin(P, cls + "transform.dense", h)), b.layer_norm_eps)
    logits = F.linear(t, P[cls + "decoder.weight"], P[cls + "bias"])  # :209 (decoder.bias is predictions.bias)
    ce = F.cross_entropy(logits.view(-1, b.vocab_size), labels.view(-1), reduction="none")  # :213-214
    loss = (ce * weights.view(-1)).mean()  # :215-217 mean over ALL B*S positions
    return loss, dict(fused=fused, seq_out=h, logits=logits, embed=e)


def forward(P, cfg, batch, mask_ratio=0.75, noise=None, train=False, return_aux=False):
    """a6: ECAMP.forward, model_ecamp.py:303-325.  Returns (mim_loss, res_loss, mlm_loss)."""
    big = batch["image"]
    if noise is None:
        noise = torch.rand(big.shape[0], cfg.num_patches)
    imgs = bicubic_resize(big, cfg.img_size)
    latent, mask, ids_restore, ids_keep = image_encoder(P, cfg, imgs, mask_ratio, noise)
    pred = image_decoder(P, cfg, latent, ids_restore)
    mim, res, pred_img, sr = forward_loss(P, cfg, imgs, big, pred, mask, batch["column"], batch["row"])
    mlm, baux = report_decoder(P, cfg, latent, batch["ids"], batch["labels"], batch["attention_mask"],
                               batch["type_ids"], batch["weights"], train)
    if not return_aux:
        return mim, res, mlm
    aux = dict(imgs=imgs, latent=latent, mask=mask, ids_restore=ids_restore, ids_keep=ids_keep, pred=pred,
               pred_img=pred_img, sr=sr, **baux)
    return (mim, res, mlm), aux


def forward_visualization(P, cfg, imgs, ids, attention_mask, type_ids, mask_ratio=0.0, noise=None):
    """f4: the Visualization variant of ECAMP.forward (Visualization/module/model_ecamp.py:308-319): eval-mode encoder on the
    224^2 image with `mask_ratio` (0 there: all 196 patches kept, but still SHUFFLED by argsort(noise) -- the reference hands
    the fusion layer the tokens in `ids_keep` order), bert_mlp, embeddings, the fusion layer's text self-attention block, and
    the cross-attention PROBABILITIES of the text onto the image tokens, which is what the Visualization fusion layer returns
    (Visualization/module/context_fusion.py:45-57 `cross_self_outputs[1]`; bert_modeling.py:113-129).
    -> [B, heads, S, L_keep] f32, plus ids_keep."""
    b = cfg.bert
    if noise is None:
        noise = torch.rand(imgs.shape[0], cfg.num_patches)
    latent, mask, ids_restore, ids_keep = image_encoder(P, cfg, imgs, mask_ratio, noise)
    lat = _lin(P, "bert_mlp", latent)
    img = lat[:, 1:, :]
    B, S = ids.shape
    pre = "bert_encoder.model.bert."
    fmin = torch.finfo(torch.float32).min
    text_mask = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * fmin
    img_mask = torch.zeros(B, 1, 1, img.shape[1])
    pos = torch.arange(S)
    e = (F.embedding(ids, P[pre + "embeddings.word_embeddings.weight"], padding_idx=0)
         + P[pre + "embeddings.token_type_embeddings.weight"][type_ids]
         + P[pre + "embeddings.position_embeddings.weight"][pos][None])
    e = _ln(P, pre + "embeddings.LayerNorm", e, b.layer_norm_eps)
    fp = pre + "context_fusion_layer"
    a = bert_self_attention(P, fp + ".attention.self", e, text_mask, b.num_attention_heads, 0.0, False)
    a = bert_self_output(P, fp + ".attention.output", a, e, b.layer_norm_eps, 0.0, False)
    _, probs = bert_self_attention(P, fp + ".cross_self_attention", a, img_mask, b.num_attention_heads, 0.0, False, kv=img,
                                   return_probs=True)
    return probs, ids_keep


# ---------------------------------------------------------------------------------------------
# engine-side host arithmetic (a1-a5)
# ---------------------------------------------------------------------------------------------
def adjust_learning_rate(epoch, lr, min_lr, warmup_epochs, max_epoch):
    """a2: PT/util/lr_sched.py:9-21 (note max_epoch, not epochs)."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epoch - warmup_epochs)))


def weight_decay_groups(cfg):
    """a4: timm 0.4.12 optim_factory.add_weight_decay -> (no_decay names, decay names)."""
    shapes = param_shapes(cfg)
    no_decay, decay = [], []
    for name in trainable_names(cfg):
        shp = shapes[name][0]
        (no_decay if (len(shp) == 1 or name.endswith(".bias")) else decay).append(name)
    return no_decay, decay


def grad_norm(grads):
    """a3: PT/util/misc.py:280-292 -- L2 norm of the per-tensor L2 norms."""
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.95, eps=1e-8):
    """a4: torch 1.13.1 AdamW (single-tensor path), betas from main_pretrain.py:254. In place."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def new_params(cfg, requires_grad=True):
    """Zero-filled parameter dict with the frozen sin-cos tables filled in (model_ecamp.py:105-112)."""
    P = OrderedDict()
    for k, (shp, tr) in param_shapes(cfg).items():
        if k in TIED:
            continue
        P[k] = torch.zeros(shp)
    P["pos_embed"] = sincos_2d(cfg.embed_dim, cfg.grid)
    P["decoder_pos_embed"] = sincos_2d(cfg.decoder_embed_dim, cfg.grid)
    return P


def load_state(P, state):
    """Copy a reference-style state dict (may contain the tied alias) into P; returns P with grad flags."""
    for k in P:
        P[k] = state[k].detach().clone().float()
    return P


def set_requires_grad(P, cfg):
    shapes = param_shapes(cfg)
    for k in P:
        P[k].requires_grad_(shapes[k][1])
    return P
