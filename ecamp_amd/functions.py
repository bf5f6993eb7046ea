"""Stage-level autograd Functions: each one is a hand-written forward AND backward made of libecamp_hip.so
kernel launches (ecamp_amd.hip_ops).  torch.autograd only chains the ~35 stages; it never differentiates an
op, never accumulates a parameter gradient (the kernels add straight into the gradient arena) and never
launches an ATen compute kernel for them.

Reference restated per stage (paths relative to ECAMP/Pre-training/):
  StemFn        model_ecamp.py:318 (bicubic), :218-230 (patch-embed, masking, cls)      K1-K4
  VitBlockFn    timm 0.4.12 Block (call sites model_ecamp.py:66-68,80-82,233-234,254-255) K5-K8
  NormFn        model_ecamp.py:69,235                                                    K5
  DecStemFn     model_ecamp.py:242-251                                                   K9,K10
  ImgLossFn     model_ecamp.py:256-262 (norm, pred), :153-165,:28-46,:196-215,:276-300   K11-K13
  ReportStemFn  model_ecamp.py:268-271                                                   K15
  BertEmbedFn   HF BertEmbeddings (bert_modeling.py:113)                                 K14
  FusionFn      context_fusion.py:21-72                                                  K16-K19
  BertLayerFn   HF BertLayer (bert_modeling.py:131)                                      K16,K17,K19
  MlmHeadFn     bert_modeling.py:209-217                                                 K20,K21
"""
import math

import torch

from . import hip_ops as ops


def _none(n):
    return (None,) * n


_BLOCK = None   # [arena, [(dy, x, gw, gb, accumulate)], [parameters reported ready], first group issued] while a block's backward collects its weight gradients


class _wgrad_block:
    """`with _wgrad_block(arena):` around the backward of one transformer block: its weight gradients (2-4 linear layers over the
    same rows) are collected and issued as ONE grouped launch on the side stream when the block's backward has been queued
    (ops.wgrad_group_async; per-layer launches when the group does not qualify).  `arena.ready(...)` calls made inside are held
    back until then: the data-parallel reducer may only see a parameter once its gradient GEMM is in a stream."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        global _BLOCK
        self.prev, _BLOCK = _BLOCK, [self.arena, [], [], False]
        self._ready = self.arena.ready
        self.arena.ready = lambda *ps, _b=_BLOCK: _b[2].extend(ps)
        return self

    def __exit__(self, et, ev, tb):
        global _BLOCK
        blk, _BLOCK = _BLOCK, self.prev
        del self.arena.ready          # back to the class method
        if et is None:
            _issue(blk[1])
            if blk[2]:
                self.arena.ready(*blk[2])
        return False


def _issue(items, workgroups=0):
    if len(items) >= 2 and ops.wgrad_group_supported(items):
        ops.wgrad_group_async(items, workgroups=workgroups)
    else:
        for dy, x, gw, gb, acc in items:
            ops.linear_wgrad_async(dy, x, gw, gb=gb, accumulate=acc)


_MLM_CHUNK_ROWS = int(__import__("os").environ.get("ECAMP_MLM_CHUNK_ROWS", "0"))   # > 0: MlmHeadFn._chunked (the decoder + CE + its backward, `rows` at a time)
_FIRST_GROUP = int(__import__("os").environ.get("ECAMP_WGRAD_GROUP_SIZE", "4"))   # layers in a block's first launch, issued as soon as they
# are collected (1: the first layer alone, 2: the MLP pair, 4: nothing early -- the whole block as ONE launch when its backward has been queued;
# with the segment-major item table of round 3 that measured 38.3 against 38.65 ms per step for the pair launches, same box)


def _wgrad(A, dy, x, w, gb=None, alpha_dev=None, shape=None):
    """Weight gradient of one nn.Linear (or of adjacent ones run as a single GEMM) on the side stream: dW (+)= dy^T x straight into
    the gradient arena -- overwriting on the first backward after zero_grad(), accumulating afterwards (ParamArena.gradw) -- and
    db += column sums of dy inside the same GEMM.  Inside a `_wgrad_block` it joins the block's grouped launch."""
    gw, acc = A.gradw(w, shape)
    if _BLOCK is not None and _BLOCK[0] is A and alpha_dev is None and len(_BLOCK[1]) < 4:
        _BLOCK[1].append((dy, x, gw, gb, acc))
        if not _BLOCK[3] and len(_BLOCK[1]) == _FIRST_GROUP:
            _BLOCK[3] = True
            _issue(_BLOCK[1])
            del _BLOCK[1][:]
        return
    ops.linear_wgrad_async(dy, x, gw, alpha_dev=alpha_dev, gb=gb, accumulate=acc)


def _f8_site(m, x, weights):
    """fp8 forward (configs[4]), delayed scaling: (scale, amax slots) to hand to `ops.layernorm_fwd(q8_site=...)` when the LayerNorm's
    output `x`-to-be feeds the e4m3 GEMM of `weights` and that site has been calibrated (its first use runs the two-pass scaling)."""
    if not (m.fp8_forward and x.dtype == torch.bfloat16):
        return None
    _, sc, am, cal = m.arena.f8_site(weights)
    return (sc, am) if cal else None


def _dense(m, x, weights, bias, shape=None, x8=None, next_w=None, **kw):
    """Forward of a dense layer y = act(x W^T + b) (+ residual).  `weights`: one parameter or a list of arena-adjacent ones run as a
    single [N, K] = `shape` GEMM (BERT's fused query/key/value).  bf16 GEMM, or -- model.fp8_forward, BASELINE configs[4]: the ViT
    qkv / proj / fc1 / fc2 layers and the BERT / fusion dense layers -- the e4m3 GEMM on per-tensor-scaled copies: the weight's is
    re-quantised once per optimizer step (ParamArena.w8), the activation's comes from its producer (`x8`: a LayerNorm that quantised
    its own output, a GELU epilogue that did the same: `next_w` asks this layer for it) or from ONE pass over x with the site's
    delayed scale.  The backward is the bf16 one either way."""
    A = m.arena
    plist = list(weights) if isinstance(weights, (list, tuple)) else [weights]
    if m.fp8_forward and x.dtype == torch.bfloat16:
        i, sc, am, cal = A.f8_site(plist)
        if x8 is None:
            x8 = ops.quantize_fp8_site(x, sc, am, cal)
            A.f8_cal.add(i)
        w8, ws = A.w8(plist, shape)
        if next_w is not None:   # GELU layer whose output feeds the dense layer of `next_w`: -> (y, pre, y8 or None)
            _, nsc, nam, ncal = A.f8_site(next_w)
            if ncal:
                return ops.gemm_fp8(x8, sc, w8, ws, bias, q8_site=(nsc, nam), **kw)
            return ops.gemm_fp8(x8, sc, w8, ws, bias, **kw) + (None,)
        return ops.gemm_fp8(x8, sc, w8, ws, bias, **kw)
    w = A.fused_w(plist, shape) if len(plist) > 1 else A.w(plist[0])
    r = ops.linear_fwd(x, w, bias, **kw)
    return r + (None,) if next_w is not None else r


def _vit_linear(m, x, lin, x8=None, next_w=None, **kw):
    """Forward of a timm Attention.qkv / proj or Mlp.fc1 / fc2 layer (see _dense)."""
    return _dense(m, x, lin.weight, lin.bias.data, x8=x8, next_w=next_w, **kw)


def _ln_q8(m, x, ln, weights, eps=None, **kw):
    """LayerNorm whose output feeds the dense layer of `weights`: -> (y, z, mean, rstd, y8 or None)."""
    site = _f8_site(m, x, weights)
    r = ops.layernorm_fwd(x, ln.weight.data, ln.bias.data, ln.eps if eps is None else eps, q8_site=site, **kw)
    return r if site is not None else r + (None,)


# =============================================================================================
class StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, big, noise, m, mask_ratio, anchor):
        A = m.arena
        cd = m.compute_dtype
        B = big.shape[0]
        R, p, D = m.img_size, m.patch_size, m.embed_dim
        L = m.num_patches
        Lk = int(L * (1 - mask_ratio))
        imgs = ops.bicubic_resize(big, R, R) if (big.shape[-1] != R or ops.is_u8_image(big)) else big
        if noise is None:
            seed, off = m.next_rng()
            noise = ops.uniform((B, L), big.device, seed, off)
        ids_restore, ids_keep, mask = ops.mask_indices(noise, Lk)
        cols = ops.im2col_gather(imgs, ids_keep, p, cd)
        pw = m.patch_embed.proj.weight
        x = ops.linear_fwd(cols, A.w(pw).view(D, -1), m.patch_embed.proj.bias.data)
        ops.assemble_tokens_(x, m.cls_token.data, m.pos_embed.data, ids_keep, B, Lk, D)
        ctx.m, ctx.cols, ctx.Lk = m, cols, Lk
        ctx.mark_non_differentiable(imgs, mask, ids_restore, ids_keep)
        return x, imgs, mask, ids_restore, ids_keep

    @staticmethod
    def backward(ctx, dx, *_):
        m, A, Tt = ctx.m, ctx.m.arena, ctx.Lk + 1
        dx = dx.contiguous()
        pe = m.patch_embed.proj
        _wgrad(A, dx, ctx.cols, pe.weight, shape=(m.embed_dim, -1))
        ops.colsum(dx, A.grad(pe.bias), period=Tt, lo=1, hi=Tt)  # cls rows carry no patch: excluded from the bias gradient
        ops.colsum(dx, A.grad(m.cls_token).view(-1), period=Tt, lo=0, hi=1)
        A.ready(pe.weight, pe.bias, m.cls_token)
        return _none(5)


# =============================================================================================
class VitBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, m, B, T, heads):
        D = x.shape[1]
        hd = D // heads
        eps = blk.norm1.eps
        h, _, mean1, rstd1, h8 = _ln_q8(m, x, blk.norm1, blk.attn.qkv.weight)
        qkv = _vit_linear(m, h, blk.attn.qkv, x8=h8)
        st = (T * 3 * D, 3 * D, hd)
        flat = qkv.view(-1)
        a, lse = ops.attn_fwd(flat, flat[D:], flat[2 * D:], B, heads, T, T, hd, st, st, st, hd ** -0.5)
        a = a.view(B * T, D)
        x1 = _vit_linear(m, a, blk.attn.proj, residual=x)
        h2, _, mean2, rstd2, h28 = _ln_q8(m, x1, blk.norm2, blk.mlp.fc1.weight)
        u, pre, u8 = _vit_linear(m, h2, blk.mlp.fc1, x8=h28, next_w=blk.mlp.fc2.weight, act=m.gelu_act, save_pre=True)   # (act 2: `pre` holds gelu')
        x2 = _vit_linear(m, u, blk.mlp.fc2, x8=u8, residual=x1)
        ctx.s = (x, mean1, rstd1, h, qkv, a, lse, x1, mean2, rstd2, h2, pre, u)
        ctx.cfg = (blk, m, B, T, heads)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x, mean1, rstd1, h, qkv, a, lse, x1, mean2, rstd2, h2, pre, u = ctx.s
        blk, m, B, T, heads = ctx.cfg
        A = m.arena
        G = A.grad
        D = x.shape[1]
        hd = D // heads
        dx2 = dx2.contiguous()
        fc1, fc2, proj, qk = blk.mlp.fc1, blk.mlp.fc2, blk.attn.proj, blk.attn.qkv
        with _wgrad_block(A):
            dx = VitBlockFn._backward_body(ctx, dx2, A, G, D, hd, fc1, fc2, proj, qk)
        ctx.s = None
        return (dx,) + _none(5)

    @staticmethod
    def _backward_body(ctx, dx2, A, G, D, hd, fc1, fc2, proj, qk):
        x, mean1, rstd1, h, qkv, a, lse, x1, mean2, rstd2, h2, pre, u = ctx.s
        blk, m, B, T, heads = ctx.cfg
        _wgrad(A, dx2, u, fc2.weight, gb=G(fc2.bias))
        dpre = ops.linear_dgrad(dx2, A.w(fc2.weight), gmul=pre, gmul_is_grad=m.gelu_act == 2)
        _wgrad(A, dpre, h2, fc1.weight, gb=G(fc1.bias))
        dh2 = ops.linear_dgrad(dpre, A.w(fc1.weight))
        dx1 = ops.layernorm_bwd(dh2, x1, mean2, rstd2, blk.norm2.weight.data, G(blk.norm2.weight), G(blk.norm2.bias), dres=dx2)
        A.ready(fc2.weight, fc2.bias, fc1.weight, fc1.bias, blk.norm2.weight, blk.norm2.bias)
        _wgrad(A, dx1, a, proj.weight, gb=G(proj.bias))
        da = ops.linear_dgrad(dx1, A.w(proj.weight))
        dqkv = torch.empty_like(qkv)
        st = (T * 3 * D, 3 * D, hd)
        f, df = qkv.view(-1), dqkv.view(-1)
        ops.attn_bwd(f, f[D:], f[2 * D:], a.view(B, T, D), da.view(B, T, D), lse, df, df[D:], df[2 * D:], B, heads, T, T, hd,
                     st, st, st, st, st, st, hd ** -0.5)
        _wgrad(A, dqkv, h, qk.weight, gb=G(qk.bias))
        dh = ops.linear_dgrad(dqkv, A.w(qk.weight))
        dx = ops.layernorm_bwd(dh, x, mean1, rstd1, blk.norm1.weight.data, G(blk.norm1.weight), G(blk.norm1.bias), dres=dx1)
        A.ready(proj.weight, proj.bias, qk.weight, qk.bias, blk.norm1.weight, blk.norm1.bias)
        return dx


# =============================================================================================
class NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln, m):
        y, _, mean, rstd = ops.layernorm_fwd(x, ln.weight.data, ln.bias.data, ln.eps)
        ctx.s = (x, mean, rstd, ln, m)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, ln, m = ctx.s
        G = m.arena.grad
        dx = ops.layernorm_bwd(dy.contiguous(), x, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias))
        m.arena.ready(ln.weight, ln.bias)
        ctx.s = None
        return dx, None, None


# =============================================================================================
class DecStemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latent, ids_restore, ids_keep, m, B):
        A = m.arena
        L, Dd = m.num_patches, m.decoder_embed_dim
        Lk = ids_keep.shape[1]
        y = ops.linear_fwd(latent, A.w(m.decoder_embed.weight), m.decoder_embed.bias.data)
        xd = ops.unshuffle_fwd(y, ids_restore, m.mask_token.data, m.decoder_pos_embed.data, B, L, Lk, Dd)
        ctx.s = (latent, ids_restore, ids_keep, m, B, Lk)
        return xd.view(B * (L + 1), Dd)

    @staticmethod
    def backward(ctx, dxd):
        latent, ids_restore, ids_keep, m, B, Lk = ctx.s
        A = m.arena
        G = A.grad
        L, Dd = m.num_patches, m.decoder_embed_dim
        dy = ops.unshuffle_bwd(dxd.contiguous(), ids_restore, ids_keep, G(m.mask_token).view(-1), B, L, Lk, Dd).view(-1, Dd)
        de = m.decoder_embed
        _wgrad(A, dy, latent, de.weight, gb=G(de.bias))
        dlat = ops.linear_dgrad(dy, A.w(de.weight))
        A.ready(de.weight, de.bias, m.mask_token)
        ctx.s = None
        return (dlat,) + _none(4)


# =============================================================================================
def _dec_head_fwd(m, xd):
    """decoder_norm -> decoder_pred on [B*(L+1), Dd] (model_ecamp.py:256-259); the cls row is still in the output."""
    ln = m.decoder_norm
    h, _, mean, rstd = ops.layernorm_fwd(xd, ln.weight.data, ln.bias.data, ln.eps)
    pred = ops.linear_fwd(h, m.arena.w(m.decoder_pred.weight), m.decoder_pred.bias.data)
    return pred, (xd, mean, rstd, h)


def _dec_head_bwd(m, rec, dpred):
    xd, mean, rstd, h = rec
    A = m.arena
    G = A.grad
    dp, ln = m.decoder_pred, m.decoder_norm
    _wgrad(A, dpred, h, dp.weight, gb=G(dp.bias))
    dh = ops.linear_dgrad(dpred, A.w(dp.weight))
    dxd = ops.layernorm_bwd(dh, xd, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias))
    A.ready(dp.weight, dp.bias, ln.weight, ln.bias)
    return dxd


def _pixel_loss_fwd(m, pred, imgs, big, mask, column, row, B):
    """unpatchify -> masked MSE  +  SR head -> windowed MSE (model_ecamp.py:276-300) on pred [B*(L+1), p*p*3] (cls row ignored).
    -> ([mim_loss, res_loss] f32, record for the backward)."""
    cd = m.compute_dtype
    R, p = m.img_size, m.patch_size
    sums = ops.zeros((2,), pred.device)
    pred_img = ops.unpatchify_mim(pred, imgs, mask, sums[0:], B, R, p)
    sr = m.super_res
    # bf16 mode: the SR stencils run on the matrix cores (u / c1 / ds / dc1 rounded to bf16); f32 mode keeps the exact f32 stencils
    sr_mode = 0 if cd == torch.float32 else 1
    ops.sr_fwd(pred_img, big, column, row, sr.conv1.weight.data, sr.conv1.bias.data, sr.conv2.weight.data, sr.conv2.bias.data,
               sums[1:], 2 * p, m.sr_window, sr_mode)
    n1, n2 = B * 3 * R * R, B * 3 * 4 * R * R
    m._aux = dict(pred=pred, pred_img=pred_img) if m.keep_aux else None
    return sums * m._loss_norm(n1, n2, pred.device), (pred_img, imgs, mask, big, column, row, B, n1, n2, pred.dtype)


def _pixel_loss_bwd(m, rec, g):
    pred_img, imgs, mask, big, column, row, B, n1, n2, pdt = rec
    G = m.arena.grad
    cd = m.compute_dtype
    R, p = m.img_size, m.patch_size
    gm_gs = (g * (2.0 * m._loss_norm(n1, n2, g.device))).contiguous()  # [g_mim*2/N1, g_res*2/N2] (2 floats, on device)
    sr = m.super_res
    ws = ops.zeros((168,), pred_img.device)
    dsr = ops.sr_bwd(pred_img, big, column, row, sr.conv1.weight.data, sr.conv1.bias.data, sr.conv2.weight.data,
                     sr.conv2.bias.data, ws, 2 * p, m.sr_window, 0 if cd == torch.float32 else 1)
    ops.scaled_accum(ws[0:81], G(sr.conv1.weight), gm_gs, 1)
    ops.scaled_accum(ws[81:84], G(sr.conv1.bias), gm_gs, 1)
    ops.scaled_accum(ws[84:165], G(sr.conv2.weight), gm_gs, 1)
    ops.scaled_accum(ws[165:168], G(sr.conv2.bias), gm_gs, 1)
    m.arena.ready(sr.conv1.weight, sr.conv1.bias, sr.conv2.weight, sr.conv2.bias)
    return ops.img_loss_bwd(pred_img, imgs, mask, dsr, gm_gs, B, R, p, pdt)


class ImgLossFn(torch.autograd.Function):
    """decoder_norm -> decoder_pred -> unpatchify -> masked MSE  +  SR head -> windowed MSE.
    Returns one 2-element f32 tensor [mim_loss, res_loss]."""

    @staticmethod
    def forward(ctx, xd, imgs, big, mask, column, row, m, B):
        pred, head = _dec_head_fwd(m, xd)
        out, tail = _pixel_loss_fwd(m, pred, imgs, big, mask, column, row, B)
        ctx.s = (m, head, tail)
        return out

    @staticmethod
    def backward(ctx, g):
        m, head, tail = ctx.s
        dxd = _dec_head_bwd(m, head, _pixel_loss_bwd(m, tail, g))
        ctx.s = None
        return (dxd,) + _none(7)


class DecHeadFn(torch.autograd.Function):
    """The tail of `image_decoder` alone (model_ecamp.py:256-259) for the stage-wise public API."""

    @staticmethod
    def forward(ctx, xd, m):
        pred, head = _dec_head_fwd(m, xd)
        ctx.s = (m, head)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        m, head = ctx.s
        ctx.s = None
        return _dec_head_bwd(m, head, dpred.contiguous()), None


class PixelLossFn(torch.autograd.Function):
    """`forward_loss` alone (model_ecamp.py:276-300) on a prediction that still carries its cls row."""

    @staticmethod
    def forward(ctx, pred, imgs, big, mask, column, row, m, B):
        out, tail = _pixel_loss_fwd(m, pred, imgs, big, mask, column, row, B)
        ctx.s = (m, tail)
        return out

    @staticmethod
    def backward(ctx, g):
        m, tail = ctx.s
        ctx.s = None
        return (_pixel_loss_bwd(m, tail, g),) + _none(7)


# =============================================================================================
class ReportStemFn(torch.autograd.Function):
    """bert_mlp on all T tokens; gap = mean over the T-1 patch tokens.  Returns (lat [B*T,H], gap [B,H])."""

    @staticmethod
    def forward(ctx, latent, m, B, T):
        A = m.arena
        lat = ops.linear_fwd(latent, A.w(m.bert_mlp.weight), m.bert_mlp.bias.data)
        H = lat.shape[1]
        gap = ops.seq_sum(lat.view(B, T, H), 1, T, 1.0 / (T - 1))
        ctx.s = (latent, m, B, T, H)
        return lat, gap

    @staticmethod
    def backward(ctx, dlat, dgap):
        latent, m, B, T, H = ctx.s
        A = m.arena
        G = A.grad
        dlat = dlat.clone(memory_format=torch.contiguous_format)   # autograd may still own the incoming buffer (a second consumer of
        ops.seq_bcast(dgap.contiguous(), dlat.view(B, T, H), 1, T, 1.0 / (T - 1), 1)   # `lat`, retain_graph): add into a private copy
        _wgrad(A, dlat, latent, m.bert_mlp.weight, gb=G(m.bert_mlp.bias))
        dl = ops.linear_dgrad(dlat, A.w(m.bert_mlp.weight))
        A.ready(m.bert_mlp.weight, m.bert_mlp.bias)
        ctx.s = None
        return (dl,) + _none(3)


# =============================================================================================
class BertEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, type_ids, emb, m, p, anchor):
        B, S = ids.shape
        seed, off = m.next_rng()
        ln = emb.LayerNorm
        e, z, mean, rstd = ops.bert_embed_fwd(ids, type_ids, emb.word_embeddings.weight.data, emb.position_embeddings.weight.data,
                                              emb.token_type_embeddings.weight.data, ln.weight.data, ln.bias.data, ln.eps,
                                              m.compute_dtype, p, seed, off)
        ctx.s = (ids, type_ids, emb, m, p, seed, off, z, mean, rstd)
        return e

    @staticmethod
    def backward(ctx, de):
        ids, type_ids, emb, m, p, seed, off, z, mean, rstd = ctx.s
        G = m.arena.grad
        B, S = ids.shape
        ln = emb.LayerNorm
        ops.bert_embed_bwd(de.contiguous(), z, mean, rstd, ln.weight.data, ids, type_ids, G(emb.word_embeddings.weight),
                           G(emb.position_embeddings.weight), G(emb.token_type_embeddings.weight), G(ln.weight), G(ln.bias), B, S,
                           z.shape[1], p, seed, off, pad_id=m.bert_config.pad_token_id)
        m.arena.ready(emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight, ln.weight, ln.bias)
        ctx.s = None
        return _none(6)


# =============================================================================================
# BERT sub-blocks shared by FusionFn and BertLayerFn (plain helpers; they append what backward needs to `tape`)
def _qkv_params(att):
    return [att.query, att.key, att.value]


def _self_attn_fwd(m, att, out, h, B, S, key_mask, pa, ph, tape, next_w=None):
    """BertAttention: LN(dropout(dense(SelfAttn(h))) + h).  q/k/v projections run as ONE GEMM over the
    arena-adjacent [3H,H] weight block."""
    A = m.arena
    H = h.shape[1]
    heads = m.bert_config.num_attention_heads
    hd = H // heads
    qkvp = _qkv_params(att)
    b3 = A.fused_f32([l.bias for l in qkvp], (3 * H,))
    qkv = _dense(m, h, [l.weight for l in qkvp], b3, shape=(3 * H, H))
    st = (S * 3 * H, 3 * H, hd)
    f = qkv.view(-1)
    s1, o1 = m.next_rng()
    a, lse, bits = ops.attn_fwd(f, f[H:], f[2 * H:], B, heads, S, S, hd, st, st, st, 1.0 / math.sqrt(hd), key_mask, pa, s1, o1, want_mask=True)
    a = a.view(B * S, H)
    y = _dense(m, a, out.dense.weight, out.dense.bias.data)
    s2, o2 = m.next_rng()
    ln = out.LayerNorm
    # (`next_w`: the dense layer this LayerNorm's output feeds -- its e4m3 copy is made here when that site is calibrated)
    o, z, mean, rstd, o8 = _ln_q8(m, y, ln, next_w, residual=h, drop_p=ph, seed=s2, offset=o2) if next_w is not None else \
        ops.layernorm_fwd(y, ln.weight.data, ln.bias.data, ln.eps, residual=h, drop_p=ph, seed=s2, offset=o2) + (None,)
    tape.append((att, out, h, qkv, a, lse, z, mean, rstd, (s1, o1), (s2, o2), bits))
    return (o, o8) if next_w is not None else o


def _self_attn_bwd(m, rec, dout, B, S, key_mask, pa, ph):
    att, out, h, qkv, a, lse, z, mean, rstd, (s1, o1), (s2, o2), bits = rec
    A = m.arena
    G = A.grad
    H = h.shape[1]
    heads = m.bert_config.num_attention_heads
    hd = H // heads
    ln = out.LayerNorm
    if ph > 0:
        dz, dy = ops.layernorm_bwd(dout, z, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias), drop_p=ph, seed=s2, offset=o2, want_drop=True)
    else:
        dz = dy = ops.layernorm_bwd(dout, z, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias))
    _wgrad(A, dy, a, out.dense.weight, gb=G(out.dense.bias))
    da = ops.linear_dgrad(dy, A.w(out.dense.weight))
    dqkv = torch.empty_like(qkv)
    st = (S * 3 * H, 3 * H, hd)
    f, df = qkv.view(-1), dqkv.view(-1)
    ops.attn_bwd(f, f[H:], f[2 * H:], a.view(B, S, H), da.view(B, S, H), lse, df, df[H:], df[2 * H:], B, heads, S, S, hd, st, st, st,
                 st, st, st, 1.0 / math.sqrt(hd), key_mask, pa, s1, o1, drop_bits=bits)
    qkvp = _qkv_params(att)
    _wgrad(A, dqkv, h, [l.weight for l in qkvp], shape=(3 * H, H), gb=A.fused_grad([l.bias for l in qkvp], (3 * H,)))
    dh = ops.linear_dgrad(dqkv, A.fused_w([l.weight for l in qkvp], (3 * H, H)), residual=dz)
    A.ready(ln.weight, ln.bias, out.dense.weight, out.dense.bias, *[l.weight for l in qkvp], *[l.bias for l in qkvp])
    return dh


def _ffn_fwd(m, inter, out, x, ph, tape, x8=None):
    """BertIntermediate + BertOutput: LN(dropout(W2 gelu(W1 x)) + x)."""
    A = m.arena
    u, pre, u8 = _dense(m, x, inter.dense.weight, inter.dense.bias.data, x8=x8, next_w=out.dense.weight, act=m.gelu_act, save_pre=True)
    y = _dense(m, u, out.dense.weight, out.dense.bias.data, x8=u8)
    s, o = m.next_rng()
    ln = out.LayerNorm
    r, z, mean, rstd = ops.layernorm_fwd(y, ln.weight.data, ln.bias.data, ln.eps, residual=x, drop_p=ph, seed=s, offset=o)
    tape.append((inter, out, x, u, pre, z, mean, rstd, (s, o)))
    return r


def _ffn_bwd(m, rec, dout, ph):
    inter, out, x, u, pre, z, mean, rstd, (s, o) = rec
    A = m.arena
    G = A.grad
    ln = out.LayerNorm
    if ph > 0:
        dz, dy = ops.layernorm_bwd(dout, z, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias), drop_p=ph, seed=s, offset=o, want_drop=True)
    else:
        dz = dy = ops.layernorm_bwd(dout, z, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias))
    _wgrad(A, dy, u, out.dense.weight, gb=G(out.dense.bias))
    dpre = ops.linear_dgrad(dy, A.w(out.dense.weight), gmul=pre, gmul_is_grad=m.gelu_act == 2)
    _wgrad(A, dpre, x, inter.dense.weight, gb=G(inter.dense.bias))
    dx = ops.linear_dgrad(dpre, A.w(inter.dense.weight), residual=dz)
    A.ready(ln.weight, ln.bias, out.dense.weight, out.dense.bias, inter.dense.weight, inter.dense.bias)
    return dx


class BertLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, layer, m, B, S, key_mask, pa, ph):
        tape = []
        a, a8 = _self_attn_fwd(m, layer.attention.self, layer.attention.output, h, B, S, key_mask, pa, ph, tape, next_w=layer.intermediate.dense.weight)
        o = _ffn_fwd(m, layer.intermediate, layer.output, a, ph, tape, x8=a8)
        ctx.s = (tape, m, B, S, key_mask, pa, ph)
        return o

    @staticmethod
    def backward(ctx, do):
        tape, m, B, S, key_mask, pa, ph = ctx.s
        with _wgrad_block(m.arena):
            da = _ffn_bwd(m, tape[1], do.contiguous(), ph)
            dh = _self_attn_bwd(m, tape[0], da, B, S, key_mask, pa, ph)
        ctx.s = None
        return (dh,) + _none(7)


class FusionFn(torch.autograd.Function):
    """ECAMPFusionLayer (context_fusion.py:21-72): text self-attention block, cross-attention of the text onto the
    49 visible image tokens (+ broadcast gap_mlp(gap token)), out_layer, FFN."""

    @staticmethod
    def forward(ctx, e, lat, gap, fl, m, B, S, T, key_mask, pa, ph):
        A = m.arena
        H = e.shape[1]
        heads = m.bert_config.num_attention_heads
        hd = H // heads
        tape = []
        a1 = _self_attn_fwd(m, fl.attention.self, fl.attention.output, e, B, S, key_mask, pa, ph, tape)
        ca = fl.cross_self_attention
        q = _dense(m, a1, ca.query.weight, ca.query.bias.data)
        bkv = A.fused_f32([ca.key.bias, ca.value.bias], (2 * H,))
        kv = _dense(m, lat, [ca.key.weight, ca.value.weight], bkv, shape=(2 * H, H))  # [B*T, 2H]; token 0 (cls) is skipped by the attention via a pointer offset
        f = kv.view(-1)
        qs, ks = (S * H, H, hd), (T * 2 * H, 2 * H, hd)
        s1, o1 = m.next_rng()
        c, lse, cbits = ops.attn_fwd(q, f[2 * H:], f[3 * H:], B, heads, S, T - 1, hd, qs, ks, ks, 1.0 / math.sqrt(hd), None, pa, s1, o1, want_mask=True)
        gp = ops.linear_fwd(gap, A.w(fl.gap_mlp.weight), fl.gap_mlp.bias.data)
        c2 = ops.bcast_add(c, gp).view(B * S, H)
        ol = fl.out_layer
        y = _dense(m, c2, ol.dense.weight, ol.dense.bias.data)
        s2, o2 = m.next_rng()
        a2, z, mean, rstd, a28 = _ln_q8(m, y, ol.LayerNorm, fl.intermediate.dense.weight, residual=a1, drop_p=ph, seed=s2, offset=o2)
        out = _ffn_fwd(m, fl.intermediate, fl.output, a2, ph, tape, x8=a28)
        ctx.s = (tape, e, lat, gap, fl, m, B, S, T, key_mask, pa, ph, a1, q, kv, c, lse, c2, z, mean, rstd, (s1, o1), (s2, o2), cbits)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tape, e, lat, gap, fl, m, B, S, T, key_mask, pa, ph, a1, q, kv, c, lse, c2, z, mean, rstd, (s1, o1), (s2, o2), cbits) = ctx.s
        A = m.arena
        G = A.grad
        H = e.shape[1]
        heads = m.bert_config.num_attention_heads
        hd = H // heads
        da2 = _ffn_bwd(m, tape[1], dout.contiguous(), ph)
        ol = fl.out_layer
        ln = ol.LayerNorm
        if ph > 0:
            dz, dy = ops.layernorm_bwd(da2, z, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias), drop_p=ph, seed=s2, offset=o2, want_drop=True)
        else:
            dz = dy = ops.layernorm_bwd(da2, z, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias))
        _wgrad(A, dy, c2, ol.dense.weight, gb=G(ol.dense.bias))
        dc2 = ops.linear_dgrad(dy, A.w(ol.dense.weight))                       # [B*S, H] == d c (broadcast add passes through)
        dgp = ops.seq_sum(dc2.view(B, S, H), 0, S, 1.0)                          # [B, H]
        gm = fl.gap_mlp
        _wgrad(A, dgp, gap, gm.weight, gb=G(gm.bias))
        dgap = ops.linear_dgrad(dgp, A.w(gm.weight))
        ca = fl.cross_self_attention
        dq = torch.empty_like(q)
        dkv = ops.zeros(kv.shape, kv.device, kv.dtype)                           # cls rows stay zero
        f, df = kv.view(-1), dkv.view(-1)
        qs, ks = (S * H, H, hd), (T * 2 * H, 2 * H, hd)
        ops.attn_bwd(q, f[2 * H:], f[3 * H:], c, dc2.view(B, S, H), lse, dq, df[2 * H:], df[3 * H:], B, heads, S, T - 1, hd, qs, ks, ks,
                     qs, ks, ks, 1.0 / math.sqrt(hd), None, pa, s1, o1, drop_bits=cbits)
        _wgrad(A, dkv, lat, [ca.key.weight, ca.value.weight], shape=(2 * H, H), gb=A.fused_grad([ca.key.bias, ca.value.bias], (2 * H,)))
        dlat = ops.linear_dgrad(dkv, A.fused_w([ca.key.weight, ca.value.weight], (2 * H, H)))
        _wgrad(A, dq, a1, ca.query.weight, gb=G(ca.query.bias))
        da1 = ops.linear_dgrad(dq, A.w(ca.query.weight), residual=dz)
        A.ready(ln.weight, ln.bias, ol.dense.weight, ol.dense.bias, gm.weight, gm.bias, ca.query.weight, ca.query.bias,
                ca.key.weight, ca.key.bias, ca.value.weight, ca.value.bias)
        de = _self_attn_bwd(m, tape[0], da1, B, S, key_mask, pa, ph)
        ctx.s = None
        return (de, dlat, dgap) + _none(8)


# =============================================================================================
class MlmHeadFn(torch.autograd.Function):
    """transform(dense+GELU+LN) -> 30000-way decoder -> weighted CE, mean over ALL B*S rows (bert_modeling.py:209-217).
    The CE kernel overwrites the logits with d loss / d logits, so the largest activation of the model exists once."""

    @staticmethod
    def forward(ctx, h, labels, weights, cls, m):
        A = m.arena
        pr = cls.predictions
        # (fp8 forward, model.fp8_head: the transform dense layer and the 30000-way decoder on the e4m3 kernel as well; the decoder's input
        # is quantised inside the transform LayerNorm)
        if m.fp8_forward and m.fp8_head:
            t1, pre = _dense(m, h, pr.transform.dense.weight, pr.transform.dense.bias.data, act=1, save_pre=True)
            ln = pr.transform.LayerNorm
            t, _, mean, rstd, t8 = _ln_q8(m, t1, ln, pr.decoder.weight)
            logits = _dense(m, t, pr.decoder.weight, pr.bias.data, x8=t8)
        else:
            t1, pre = ops.linear_fwd(h, A.w(pr.transform.dense.weight), pr.transform.dense.bias.data, act=1, save_pre=True)
            ln = pr.transform.LayerNorm
            t, _, mean, rstd = ops.layernorm_fwd(t1, ln.weight.data, ln.bias.data, ln.eps)
            if _MLM_CHUNK_ROWS > 0 and t.shape[0] > _MLM_CHUNK_ROWS and ctx.needs_input_grad[0] and not m.keep_aux:
                return MlmHeadFn._chunked(ctx, h, labels, weights, cls, m, t1, pre, mean, rstd, t, _MLM_CHUNK_ROWS)
            logits = ops.linear_fwd(t, A.w(pr.decoder.weight), pr.bias.data)
        if m.keep_aux:
            m._aux_logits = logits.clone()
        s = ops.zeros((1,), h.device)
        # IEEE-half mode: the kernel leaves 256 M x the gradient (|w (p - onehot)| x 256 <= 65504 for token weights up to 255; a probability
        # keeps its 11 bits down to 2.4e-7) and backward divides the upstream gradient -- the loss scale -- by the same power of two
        ctx.gain = 256.0 * logits.shape[0] if logits.dtype == torch.float16 else 1.0
        ops.ce_fwd_bwd_(logits, labels.view(-1), weights.view(-1), s, gain=ctx.gain)
        ctx.s = (h, pre, t1, mean, rstd, t, logits, cls, m)
        return s * (1.0 / logits.shape[0])

    @staticmethod
    def _chunked(ctx, h, labels, weights, cls, m, t1, pre, mean, rstd, t, chunk):
        """SURVEY K20, the form that fits this machine ("never materialise the logits" without recomputing them): the 30000-way decoder,
        the cross-entropy and the decoder's OWN backward run chunk by chunk over the rows inside forward -- logits of `chunk` rows are
        written, turned into their gradient in place and consumed by the data- and weight-gradient GEMMs while they are still in the
        last-level cache; what survives the forward pass is d loss / d t [M, 768] and the decoder's weight / bias gradient for a UNIT upstream
        gradient (f32, 92 MB), which backward scales by the upstream gradient that arrives then.  The [M, 30000] tensor (1.97 GB at
        configs[1]) never exists."""
        A, pr = m.arena, cls.predictions
        M, V = t.shape[0], pr.decoder.weight.shape[0]
        W = A.w(pr.decoder.weight)
        lab, wts = labels.view(-1), weights.view(-1)
        s = ops.zeros((1,), h.device)
        gain = 256.0 * M if t.dtype == torch.float16 else 1.0
        gw = torch.empty((V, W.shape[1]), device=h.device, dtype=torch.float32)
        gb = ops.zeros((V,), h.device)
        dts = []
        for r0 in range(0, M, chunk):
            r1 = min(M, r0 + chunk)
            lg = ops.linear_fwd(t[r0:r1], W, pr.bias.data)
            ops.ce_fwd_bwd_(lg, lab[r0:r1], wts[r0:r1], s, gain=gain * (r1 - r0) / M)     # (the kernel divides by its own row count)
            dts.append(ops.linear_dgrad(lg, W))
            ops.linear_wgrad(lg, t[r0:r1], gw, gb=gb, accumulate=r0 > 0)
        ctx.gain = gain
        ctx.s = (h, pre, t1, mean, rstd, t, None, cls, m)
        ctx.unit = (torch.cat(dts, 0), gw, gb)
        return s * (1.0 / M)

    @staticmethod
    def backward(ctx, g):
        h, pre, t1, mean, rstd, t, dlog, cls, m = ctx.s
        A = m.arena
        G = A.grad
        pr = cls.predictions
        g = g.contiguous() if ctx.gain == 1.0 else g * (1.0 / ctx.gain)
        if dlog is None:     # chunked head: the decoder's gradients exist for a unit upstream gradient
            dt_u, gw_u, gb_u = ctx.unit
            ctx.unit = None
            gw, acc = A.gradw(pr.decoder.weight)
            if not acc:
                ops.zero_(gw)
            gs = g.reshape(1)
            ops.scaled_accum(gw_u.view(-1), gw.view(-1), gs, 0)
            ops.scaled_accum(gb_u, G(pr.bias), gs, 0)
            dt = ops.scale_(dt_u, alpha_dev=gs)     # (in f32 arithmetic: the upstream gradient is not a power of two in general)
        else:
            _wgrad(A, dlog, t, pr.decoder.weight, alpha_dev=g, gb=G(pr.bias))
            dt = ops.linear_dgrad(dlog, A.w(pr.decoder.weight), alpha_dev=g)
        ln = pr.transform.LayerNorm
        dt1 = ops.layernorm_bwd(dt, t1, mean, rstd, ln.weight.data, G(ln.weight), G(ln.bias))
        # t1 = gelu(pre): chain through GELU' elementwise via the dgrad-style epilogue of an identity is not available;
        # fold it into the transform.dense backward: d pre = dt1 * gelu'(pre)
        dpre = ops.mul_gelu_grad(dt1, pre)
        td = pr.transform.dense
        _wgrad(A, dpre, h, td.weight, gb=G(td.bias))
        dh = ops.linear_dgrad(dpre, A.w(td.weight))
        A.ready(pr.decoder.weight, pr.bias, ln.weight, ln.bias, td.weight, td.bias)
        ctx.s = None
        return (dh,) + _none(4)
