"""-m gpu: every HIP kernel, called through the C ABI (ecamp_amd.hip_ops -> libecamp_hip.so), against a plain
fp32 PyTorch CPU restatement of the same op on identical seeded inputs.

Tolerances (relative to max|ref|): f32 mode 2e-5 (exact-f32 MFMA, different summation order only);
bf16 mode 2e-2 on activations (8-bit mantissa storage; inputs are pre-rounded to bf16 so only the
kernel's own rounding is measured) and 1e-2 on f32-accumulated parameter gradients.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import h16

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}   # half keeps 11 significant bits where bfloat16 keeps 8


def ops():
    from ecamp_amd import hip_ops
    return hip_ops


def rnd(t, dtype):
    return t.to(dtype).to(torch.float32)


def check(name, got, ref, tol):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), name + ": non-finite output"
    err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-20)
    print("  %-40s rel-err %.3e (tol %.1e)" % (name, err, tol))
    assert err <= tol, "%s: rel err %.3e > %.1e" % (name, err, tol)


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(394, 768, 192), (100, 2304, 768), (512, 128, 3072), (256, 30000, 768), (37 * 4, 512, 64)])
def test_gemm_fwd_epilogues(dev, dtype, M, N, K):
    o = ops()
    x, w, b, r = rnd(gen(M, K, seed=1), dtype), rnd(gen(N, K, seed=2, scale=K ** -0.5), dtype), gen(N, seed=3), rnd(gen(M, N, seed=4), dtype)
    xd, wd, bd, rd = x.to(dev, dtype), w.to(dev, dtype), b.to(dev), r.to(dev, dtype)
    tol = TOL[dtype]
    y = o.linear_fwd(xd, wd, bd)
    check("linear", y, x @ w.T + b, tol)
    y = o.linear_fwd(xd, wd, None, residual=rd)
    check("linear+residual", y, x @ w.T + r, tol)
    y, pre = o.linear_fwd(xd, wd, bd, act=1, save_pre=True)
    ref_pre = x @ w.T + b
    check("linear pre", pre, ref_pre, tol)
    check("linear gelu", y, F.gelu(rnd(ref_pre, dtype) if dtype != torch.float32 else ref_pre), tol)
    if dtype != torch.float32:
        y32 = o.linear_fwd(xd, wd, bd, out_dtype=torch.float32)
        assert y32.dtype == torch.float32
        check("linear f32-out", y32, x @ w.T + b, 2e-3)


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("M,N,K,q8", [(394, 768, 192, 0), (1000, 520, 200, 2), (512, 3072, 768, 2), (512, 3072, 768, 0)])
def test_gemm_gelu_saved_derivative(dev, M, N, K, q8):
    """ecamp_gemm act = 2 (bf16): the GELU epilogue leaves gelu'(pre-activation) in `pre_out` instead of the pre-activation, and the
    data-gradient form multiplies by what it is handed -- torch's GeluBackward (nn.GELU at timm Mlp.act / HF BertIntermediate) split
    over the two epilogues.  Both kernel families (128^2 and the persistent one); y is bit-identical to the act = 1 result, the saved
    derivative is gelu' of the ROUNDED pre-activation to one bf16 rounding, and the two-step gradient equals the one-step (act = 1)
    gradient to bf16 accuracy."""
    o = ops()
    dt = h16()
    x, w, b = rnd(gen(M, K, seed=1), dt), rnd(gen(N, K, seed=2, scale=K ** -0.5), dt), gen(N, seed=3)
    dy, w2 = rnd(gen(M, 256, seed=5), dt), rnd(gen(256, N, seed=6, scale=256 ** -0.5), dt)   # the next layer: [M,256] = gelu(..)[M,N] @ w2^T
    xd, wd, bd, dyd, w2d = x.to(dev, dt), w.to(dev, dt), b.to(dev), dy.to(dev, dt), w2.to(dev, dt)
    try:
        o.set_option("q8_mode", q8)
        y1, pre = o.linear_fwd(xd, wd, bd, act=1, save_pre=True)
        y2, der = o.linear_fwd(xd, wd, bd, act=2, save_pre=True)
        assert torch.equal(y1, y2), "the activation must not depend on what is saved beside it"
        pr = pre.float().cpu().requires_grad_(True)
        F.gelu(pr).sum().backward()
        check("saved gelu'", der, pr.grad, 2.0 ** -8)          # one bf16 rounding of a value in [-0.13, 1.13]
        d1 = o.linear_dgrad(dyd, w2d, gmul=pre)
        d2 = o.linear_dgrad(dyd, w2d, gmul=der, gmul_is_grad=True)
        check("dgrad * saved gelu' vs dgrad * gelu'(pre)", d2, d1.float(), 2.0 ** -7)
        check("dgrad * saved gelu' vs autograd", d2, (dy.float() @ w2.float()) * pr.grad, TOL[dt])
        with pytest.raises(RuntimeError):
            o.linear_fwd(xd.float(), wd.float(), bd, act=2, save_pre=True)   # bf16 only
    finally:
        o.set_option("q8_mode", -1)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(394, 768, 192), (100, 3072, 768), (256, 30000, 768)])
def test_gemm_dgrad_wgrad(dev, dtype, M, N, K):
    o = ops()
    x = rnd(gen(M, K, seed=1), dtype)
    w = rnd(gen(N, K, seed=2, scale=N ** -0.5), dtype)
    dy = rnd(gen(M, N, seed=5), dtype)
    pre = rnd(gen(M, K, seed=6), dtype)
    xd, wd, dyd, pred = x.to(dev, dtype), w.to(dev, dtype), dy.to(dev, dtype), pre.to(dev, dtype)
    tol = TOL[dtype]
    check("dgrad", o.linear_dgrad(dyd, wd), dy @ w, tol)
    pr = pre.clone().requires_grad_(True)
    F.gelu(pr).backward(dy @ w)
    check("dgrad*gelu'", o.linear_dgrad(dyd, wd, gmul=pred), pr.grad, tol)
    check("dgrad alpha", o.linear_dgrad(dyd, wd, alpha=0.25), 0.25 * (dy @ w), tol)
    check("dgrad + residual", o.linear_dgrad(dyd, wd, residual=pred), dy @ w + pre, tol)
    check("dgrad*gelu' + residual", o.linear_dgrad(dyd, wd, gmul=pred, residual=xd), pr.grad + x, tol)
    g0 = gen(N, K, seed=7)
    gw = g0.to(dev).contiguous()
    o.linear_wgrad(dyd, xd, gw, alpha=0.5)
    check("wgrad (accumulate, split-K)", gw, g0 + 0.5 * (dy.T @ x), 2e-5 if dtype == torch.float32 else 1e-2)
    gb = torch.zeros(N, device=dev)
    o.colsum(dyd, gb, alpha=2.0)
    check("bias grad", gb, 2.0 * dy.sum(0), 2e-5 if dtype == torch.float32 else 1e-2)
    gw2, gb2 = torch.zeros(N, K, device=dev), gen(N, seed=9).to(dev)
    o.linear_wgrad(dyd, xd, gw2, alpha=0.5, gb=gb2)  # bias gradient fused into the wgrad GEMM (ones-fragment MFMA)
    check("wgrad + fused bias grad (w)", gw2, 0.5 * (dy.T @ x), 2e-5 if dtype == torch.float32 else 1e-2)
    check("wgrad + fused bias grad (b)", gb2, gen(N, seed=9) + 0.5 * dy.sum(0), 2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.usefixtures("both_halves")
def test_gemm_rejects_bad_alignment(dev):
    o = ops()
    from ecamp_amd._lib import EcampHipError
    x = torch.zeros(8, 30, device=dev, dtype=h16())
    w = torch.zeros(16, 30, device=dev, dtype=h16())
    with pytest.raises(EcampHipError):
        o.linear_fwd(x, w)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols,eps", [(100, 768, 1e-6), (394, 512, 1e-6), (7, 192, 1e-6), (256, 768, 1e-12), (33, 1024, 1e-6),
                                           (40, 2048, 1e-6)])   # 2048 = the documented maximum (64 KB of reduction slices in the backward)
def test_layernorm(dev, dtype, rows, cols, eps):
    o = ops()
    x = rnd(gen(rows, cols, seed=1) * 2 + 0.3, dtype)
    res = rnd(gen(rows, cols, seed=2), dtype)
    g, b = 1 + 0.1 * gen(cols, seed=3), 0.1 * gen(cols, seed=4)
    dy = rnd(gen(rows, cols, seed=5), dtype)
    dres = rnd(gen(rows, cols, seed=6), dtype)
    tol = TOL[dtype]
    gtol = 2e-5 if dtype == torch.float32 else 1e-2
    # plain
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (cols,), gr, br, eps)
    yr.backward(dy)
    y, z, mean, rstd = o.layernorm_fwd(x.to(dev, dtype), g.to(dev), b.to(dev), eps)
    check("ln y", y, yr, tol)
    check("ln mean", mean, x.mean(1), 1e-5)
    gg, gb = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dz = o.layernorm_bwd(dy.to(dev, dtype), z, mean, rstd, g.to(dev), gg, gb, dres=dres.to(dev, dtype))
    check("ln dx(+dres)", dz, xr.grad + dres, tol)
    check("ln dgamma", gg, gr.grad, gtol)
    check("ln dbeta", gb, br.grad, gtol)
    # fused residual
    y2, z2, m2, r2 = o.layernorm_fwd(x.to(dev, dtype), g.to(dev), b.to(dev), eps, residual=res.to(dev, dtype))
    zz = rnd(x + res, dtype)
    check("ln(x+res) z", z2, zz, tol)
    check("ln(x+res) y", y2, F.layer_norm(zz, (cols,), g, b, eps), tol)


@pytest.mark.parametrize("dtype", DT)
def test_layernorm_dropout_consistency(dev, dtype):
    """fwd and bwd must regenerate the SAME Philox mask; keep-rate ~ 1-p; scale 1/(1-p)."""
    o = ops()
    rows, cols, p = 512, 768, 0.1
    x = torch.ones(rows, cols, device=dev, dtype=dtype)
    res = torch.zeros(rows, cols, device=dev, dtype=dtype)
    g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
    y, z, mean, rstd = o.layernorm_fwd(x, g, b, 1e-6, residual=res, drop_p=p, seed=1234, offset=77)
    zf = z.float().cpu()
    keep = (zf != 0)
    rate = keep.float().mean().item()
    assert abs(rate - 0.9) < 0.01, rate
    assert torch.allclose(zf[keep], torch.full_like(zf[keep], 1 / 0.9), rtol=1e-2)
    dy = torch.randn(rows, cols, device=dev).to(dtype)
    gg, gb = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dz, dxd = o.layernorm_bwd(dy, z, mean, rstd, g, gg, gb, drop_p=p, seed=1234, offset=77, want_drop=True)
    dzf, dxf = dz.float().cpu(), dxd.float().cpu()
    assert ((dxf != 0) <= keep).all(), "bwd mask is not a subset of the fwd mask"
    check("dropout bwd scale", dxf, dzf * keep.float() / 0.9, 2e-2 if dtype != torch.float32 else 1e-6)
    # a different offset must give a different mask
    _, z3, _, _ = o.layernorm_fwd(x, g, b, 1e-6, residual=res, drop_p=p, seed=1234, offset=78)
    assert (z3.float().cpu() != zf).float().mean().item() > 0.05


# ------------------------------------------------------------------------------------------------ attention
def ref_attn(q, k, v, scale, key_mask=None):
    # q [B,H,Tq,hd], k/v [B,H,Tk,hd]
    s = (q @ k.transpose(-1, -2)) * scale
    if key_mask is not None:
        s = s + (1.0 - key_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    return s.softmax(-1) @ v


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,T,hd", [(2, 3, 50, 64), (2, 16, 197, 32), (3, 12, 50, 64), (1, 2, 130, 128)])
def test_attention_packed_qkv(dev, dtype, B, H, T, hd):
    """timm layout: one [B,T,3,H,hd] buffer straight out of the qkv GEMM."""
    o = ops()
    D = H * hd
    qkv = rnd(gen(B, T, 3, H, hd, seed=1), dtype)
    do = rnd(gen(B, T, D, seed=2), dtype)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (qr[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    out = ref_attn(q, k, v, hd ** -0.5).transpose(1, 2).reshape(B, T, D)
    out.backward(do)
    qd = qkv.to(dev, dtype)
    st = (T * 3 * D, 3 * D, hd)
    qp, kp, vp = qd.view(-1)[0:], qd.view(-1)[D:], qd.view(-1)[2 * D:]
    og, lse = o.attn_fwd(qp, kp, vp, B, H, T, T, hd, st, st, st, hd ** -0.5)
    tol = TOL[dtype]
    check("attn out", og, out, tol)
    dqkv = torch.empty_like(qd)
    dv_ = dqkv.view(-1)
    o.attn_bwd(qp, kp, vp, og, do.to(dev, dtype), lse, dv_[0:], dv_[D:], dv_[2 * D:], B, H, T, T, hd, st, st, st, st, st, st, hd ** -0.5)
    check("attn dqkv", dqkv, qr.grad, tol * 2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("S,Tk,cross", [(128, 128, False), (256, 256, False), (128, 49, True), (40, 49, True)])
def test_attention_bert_masked_and_cross(dev, dtype, S, Tk, cross):
    """HF layout: separate [B,S,H*hd] projections; key-padding mask; cross-attention reads tokens 1..49 of a 50-token buffer."""
    o = ops()
    B, H, hd = 2, 6, 128
    D = H * hd
    q = rnd(gen(B, S, D, seed=1), dtype)
    kvlen = Tk + 1 if cross else Tk
    k = rnd(gen(B, kvlen, D, seed=2), dtype)
    v = rnd(gen(B, kvlen, D, seed=3), dtype)
    do = rnd(gen(B, S, D, seed=4), dtype)
    lens = torch.tensor([Tk // 2 + 3, Tk])
    km = None if cross else (torch.arange(Tk)[None, :] < lens[:, None]).int()
    qr, kr, vr = q.clone().requires_grad_(True), k.clone().requires_grad_(True), v.clone().requires_grad_(True)
    off = 1 if cross else 0
    sp = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)
    out = ref_attn(sp(qr, S), sp(kr[:, off:], Tk), sp(vr[:, off:], Tk), 1 / math.sqrt(hd), km).permute(0, 2, 1, 3).reshape(B, S, D)
    out.backward(do)
    qd, kd, vd = q.to(dev, dtype), k.to(dev, dtype), v.to(dev, dtype)
    kmd = km.to(dev) if km is not None else None
    qs, ks = (S * D, D, hd), (kvlen * D, D, hd)
    kp, vp = kd.view(-1)[off * D:], vd.view(-1)[off * D:]
    og, lse = o.attn_fwd(qd, kp, vp, B, H, S, Tk, hd, qs, ks, ks, 1 / math.sqrt(hd), key_mask=kmd)
    tol = TOL[dtype]
    check("bert attn out", og, out, tol)
    dq = torch.empty_like(qd)
    dk = torch.zeros_like(kd)
    dv = torch.zeros_like(vd)
    o.attn_bwd(qd, kp, vp, og, do.to(dev, dtype), lse, dq, dk.view(-1)[off * D:], dv.view(-1)[off * D:], B, H, S, Tk, hd, qs, ks, ks,
               qs, ks, ks, 1 / math.sqrt(hd), key_mask=kmd)
    check("bert attn dq", dq, qr.grad, tol * 2)
    check("bert attn dk", dk, kr.grad, tol * 2)
    check("bert attn dv", dv, vr.grad, tol * 2)


def test_attention_dropout_statistics(dev):
    """P-dropout: E[out] is preserved and fwd/bwd use the same mask (finite-difference-free check via linearity in v)."""
    o = ops()
    B, H, T, hd = 4, 6, 128, 128
    D = H * hd
    q = gen(B, T, D, seed=1).to(dev)
    k = gen(B, T, D, seed=2).to(dev)
    v = torch.ones(B, T, D, device=dev)
    st = (T * D, D, hd)
    o0, _ = o.attn_fwd(q, k, v, B, H, T, T, hd, st, st, st, hd ** -0.5)
    o1, lse = o.attn_fwd(q, k, v, B, H, T, T, hd, st, st, st, hd ** -0.5, drop_p=0.1, seed=5, offset=9)
    assert torch.allclose(o0, torch.ones_like(o0), atol=1e-4)  # softmax rows sum to 1
    m = o1.mean().item()
    assert abs(m - 1.0) < 0.02, m
    assert o1.std().item() > 1e-3
    # out is linear in v for a fixed mask: dv from bwd with do=1 equals column sums of the dropped P, whose total is B*H*T*mean(o1)*hd
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    o.attn_bwd(q, k, v, o1, torch.ones_like(o1), lse, dq, dk, dv, B, H, T, T, hd, st, st, st, st, st, st, hd ** -0.5, drop_p=0.1, seed=5, offset=9)
    assert abs(dv.sum().item() / o1.sum().item() - 1.0) < 1e-3


@pytest.mark.parametrize("B,H,Tq,Tk,hd,masked,p", [(2, 6, 128, 128, 128, True, 0.1), (2, 6, 128, 49, 128, False, 0.1), (3, 16, 197, 197, 32, False, 0.0),
                                                   (2, 12, 50, 50, 64, False, 0.2), (1, 4, 250, 256, 64, True, 0.1), (2, 3, 33, 17, 32, True, 0.3),
                                                   # the same kernel instantiation first below, then above the 48 KB LDS opt-in threshold
                                                   (1, 2, 70, 70, 64, False, 0.0), (1, 2, 200, 200, 64, False, 0.0)])
@pytest.mark.usefixtures("both_halves")
def test_attention_head_kernels_match_streaming_kernels(dev, B, H, Tq, Tk, hd, masked, p):
    """The head-resident kernels (one workgroup per (batch, head), P kept in registers, one fused backward kernel) and the 64-row
    streaming kernels implement the same function with the same Philox dropout mask: same seed/offset -> same dropped entries, outputs
    and gradients equal to bf16 rounding.  Covers key masks, cross attention with an odd key count (per-element mask path) and
    sequences that do not fill the last 32-key pair."""
    o = ops()
    D = H * hd
    dt = h16()
    q = rnd(gen(B, Tq, D, seed=11), dt).to(dev, dt)
    k = rnd(gen(B, Tk, D, seed=12), dt).to(dev, dt)
    v = rnd(gen(B, Tk, D, seed=13), dt).to(dev, dt)
    do = rnd(gen(B, Tq, D, seed=14), dt).to(dev, dt)
    km = None
    if masked:
        lens = torch.randint(max(1, Tk // 3), Tk + 1, (B,), generator=torch.Generator().manual_seed(5))
        km = (torch.arange(Tk)[None, :] < lens[:, None]).int().to(dev)
    qs, ks = (Tq * D, D, hd), (Tk * D, D, hd)
    res = []
    try:
        for mode in (1, 0):
            o.set_option("attn_head", mode)
            out, lse = o.attn_fwd(q, k, v, B, H, Tq, Tk, hd, qs, ks, ks, hd ** -0.5, km, p, 77, 5)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            o.attn_bwd(q, k, v, out, do, lse, dq, dk, dv, B, H, Tq, Tk, hd, qs, ks, ks, qs, ks, ks, hd ** -0.5, km, p, 77, 5)
            res.append((out.float().cpu(), lse.float().cpu(), dq.float().cpu(), dk.float().cpu(), dv.float().cpu()))
    finally:
        o.set_option("attn_head", -1)
    for name, a, b in zip(("out", "lse", "dq", "dk", "dv"), res[0], res[1]):
        assert torch.isfinite(a).all(), name
        tol = 1e-5 if name == "lse" else 2e-2   # two bf16 roundings of P / dS taken in different summation orders
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
        assert err <= tol, "%s: head vs streaming rel err %.3e" % (name, err)
    if p > 0:   # dropout really happened and hit the same entries: dv = P_dropped^T dO differs from the undropped run
        o.set_option("attn_head", 1)
        try:
            out0, _ = o.attn_fwd(q, k, v, B, H, Tq, Tk, hd, qs, ks, ks, hd ** -0.5, km, 0.0, 77, 5)
        finally:
            o.set_option("attn_head", -1)
        assert (out0.float().cpu() - res[0][0]).abs().max().item() > 1e-3


@pytest.mark.parametrize("B,H,Tq,Tk,hd,masked,p", [(2, 6, 128, 128, 128, True, 0.1), (2, 6, 128, 49, 128, False, 0.1), (1, 4, 250, 256, 64, True, 0.1),
                                                   (2, 3, 33, 17, 32, True, 0.3), (2, 12, 50, 50, 64, False, 0.2)])
@pytest.mark.usefixtures("both_halves")
def test_attention_saved_dropout_bits_equal_regenerated_mask(dev, B, H, Tq, Tk, hd, masked, p):
    """The forward pass can leave the dropout keep-mask as bits (ecamp_attn_mask_bytes / drop_mask) so that the backward pass does not
    evaluate Philox again: the gradients must be BIT-IDENTICAL to the backward pass that regenerates the mask from (seed, offset), for
    key masks, an odd key count (per-element Philox path), partial last tile pairs and the longest head-resident sequence."""
    o = ops()
    D = H * hd
    dt = h16()
    q = rnd(gen(B, Tq, D, seed=21), dt).to(dev, dt)
    k = rnd(gen(B, Tk, D, seed=22), dt).to(dev, dt)
    v = rnd(gen(B, Tk, D, seed=23), dt).to(dev, dt)
    do = rnd(gen(B, Tq, D, seed=24), dt).to(dev, dt)
    km = None
    if masked:
        lens = torch.randint(max(1, Tk // 3), Tk + 1, (B,), generator=torch.Generator().manual_seed(6))
        km = (torch.arange(Tk)[None, :] < lens[:, None]).int().to(dev)
    qs, ks = (Tq * D, D, hd), (Tk * D, D, hd)
    out, lse, bits = o.attn_fwd(q, k, v, B, H, Tq, Tk, hd, qs, ks, ks, hd ** -0.5, km, p, 91, 3, want_mask=True)
    out2, lse2 = o.attn_fwd(q, k, v, B, H, Tq, Tk, hd, qs, ks, ks, hd ** -0.5, km, p, 91, 3)
    assert bits is not None and bits.dtype == torch.uint8 and bits.numel() == B * H * Tq * 32
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    grads = []
    for b_ in (bits, None):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        o.attn_bwd(q, k, v, out, do, lse, dq, dk, dv, B, H, Tq, Tk, hd, qs, ks, ks, qs, ks, ks, hd ** -0.5, km, p, 91, 3, drop_bits=b_)
        grads.append((dq.clone(), dk.clone(), dv.clone()))
    for name, a, b in zip(("dq", "dk", "dv"), grads[0], grads[1]):
        assert torch.isfinite(a.float()).all(), name
        assert torch.equal(a, b), "%s: saved bits differ from the regenerated mask (max diff %.3e)" % (name, (a.float() - b.float()).abs().max().item())
    # the share of kept probabilities recorded in the bits is 1 - p
    nkp = ((Tk + 15) // 16 + 1) // 2
    kept = sum(bin(int(x)).count("1") for x in bits.view(B * H * Tq, 4, 8)[:, :, :nkp].flatten().cpu().tolist()[:20000])
    total = min(B * H * Tq * 4 * nkp, 20000) * 8
    assert abs(kept / total - (1 - p)) < 0.02


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Tq,Tk,hd,masked,cross,p", [(2, 6, 128, 128, 128, True, False, 0.1),    # report side: key mask, hd 128 (every timed BERT layer)
                                                         (2, 6, 128, 49, 128, False, True, 0.1),     # fusion cross-attention onto tokens 1..49, odd key count
                                                         (2, 3, 33, 17, 32, True, False, 0.3),       # odd Tk: the per-element Philox path
                                                         (2, 12, 50, 50, 64, False, False, 0.2),
                                                         (1, 4, 250, 256, 64, True, False, 0.1)])    # the longest head-resident sequence
def test_attention_dropout_matches_pytorch_under_the_same_mask(dev, dtype, B, H, Tq, Tk, hd, masked, cross, p):
    _attn_dropout_case(dev, dtype, B, H, Tq, Tk, hd, masked, cross, p)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,Tq,Tk,hd,p", [(2, 6, 128, 128, 128, 0.1), (2, 6, 128, 128, 128, 0.0), (1, 4, 250, 256, 64, 0.1), (2, 3, 33, 17, 32, 0.3)])
def test_attention_padded_keys_with_large_scores_stay_out_of_the_gradients(dev, dtype, B, H, Tq, Tk, hd, p):
    """A padded report token is a key like any other to the QK^T product: its raw score is not part of the row's log-sum-exp, so
    2^(s - lse) can be astronomically large (seen: 233 steps into a run in IEEE half, where it is inf and inf x 0 = NaN turned dV of the
    padded key, then every gradient upstream of the value projection, into NaN -- the dynamic loss scaler halved itself to zero).  Keys
    behind the mask carry 40 x the scale of the real ones here: forward and all three gradients must equal the reference, and the padded
    keys' dK / dV rows must be exactly zero."""
    _attn_dropout_case(dev, dtype, B, H, Tq, Tk, hd, True, False, p, pad_scale=40.0)


def _attn_dropout_case(dev, dtype, B, H, Tq, Tk, hd, masked, cross, p, pad_scale=1.0):
    """HF BertSelfAttention in TRAIN mode (context_fusion.py:28-57 / bert_modeling.py:131: `attention_probs = self.dropout(probs)`):
    out = (softmax(s) * keep / (1 - p)) @ v and its gradients, against plain fp32 PyTorch under the SAME keep-mask -- the Philox mask of
    (seed, offset) materialised by the development ABI `ecamp_dropout_mask` (element index ((b*H + h)*Tq + i)*Tk + j).  Both the
    backward that reads the forward's saved mask bits and the one that regenerates the mask are held to the reference."""
    o = ops()
    D = H * hd
    seed, offset = 0x1234ABCD5678, 41
    q = rnd(gen(B, Tq, D, seed=31), dtype)
    kvlen = Tk + 1 if cross else Tk
    k = rnd(gen(B, kvlen, D, seed=32), dtype)
    v = rnd(gen(B, kvlen, D, seed=33), dtype)
    do = rnd(gen(B, Tq, D, seed=34), dtype)
    km = None
    if masked:
        lens = torch.randint(max(1, Tk // 3), Tk + 1, (B,), generator=torch.Generator().manual_seed(7))
        km = (torch.arange(Tk)[None, :] < lens[:, None]).int()
        if pad_scale != 1.0:
            assert not cross
            k = torch.where(km[:, :, None].bool(), k, rnd(k * pad_scale, dtype))
    keep = o.dropout_mask((B, H, Tq, Tk), dev, p, seed, offset).float().cpu() if p > 0 else torch.ones(B, H, Tq, Tk)
    assert p == 0 or abs(keep.mean().item() - (1 - p)) < 0.02
    off = 1 if cross else 0
    qr, kr, vr = q.clone().requires_grad_(True), k.clone().requires_grad_(True), v.clone().requires_grad_(True)
    sp = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)
    sc = (sp(qr, Tq) @ sp(kr[:, off:], Tk).transpose(-1, -2)) / math.sqrt(hd)
    if km is not None:
        sc = sc + (1.0 - km[:, None, None, :].float()) * torch.finfo(torch.float32).min
    out = ((sc.softmax(-1) * keep / (1 - p)) @ sp(vr[:, off:], Tk)).permute(0, 2, 1, 3).reshape(B, Tq, D)
    out.backward(do)
    qd, kd, vd = q.to(dev, dtype), k.to(dev, dtype), v.to(dev, dtype)
    kmd = km.to(dev) if km is not None else None
    qs, ks = (Tq * D, D, hd), (kvlen * D, D, hd)
    kp, vp = kd.view(-1)[off * D:], vd.view(-1)[off * D:]
    og, lse, bits = o.attn_fwd(qd, kp, vp, B, H, Tq, Tk, hd, qs, ks, ks, 1 / math.sqrt(hd), kmd, p, seed, offset, want_mask=True)
    tol = TOL[dtype]
    check("dropout attn out", og, out, tol)
    for b_ in ([bits, None] if bits is not None else [None]):
        dq, dk, dv = torch.empty_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
        o.attn_bwd(qd, kp, vp, og, do.to(dev, dtype), lse, dq, dk.view(-1)[off * D:], dv.view(-1)[off * D:], B, H, Tq, Tk, hd, qs, ks, ks,
                   qs, ks, ks, 1 / math.sqrt(hd), kmd, p, seed, offset, drop_bits=b_)
        tag = "saved bits" if b_ is not None else "regenerated"
        check("dropout attn dq (%s)" % tag, dq, qr.grad, tol * 2)
        check("dropout attn dk (%s)" % tag, dk, kr.grad, tol * 2)
        check("dropout attn dv (%s)" % tag, dv, vr.grad, tol * 2)
        if pad_scale != 1.0:
            pad = (km == 0)
            assert float(dk.float().cpu()[pad].abs().max()) == 0.0 and float(dv.float().cpu()[pad].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols,p", [(512, 768, 0.1), (100, 768, 0.1), (37, 192, 0.3), (64, 1024, 0.1)])
def test_layernorm_dropout_residual_matches_pytorch_under_the_same_mask(dev, dtype, rows, cols, p):
    """HF BertSelfOutput / BertOutput in TRAIN mode (LN(dropout(dense_out) + residual), context_fusion.py:37-39,56,70-72 and the six
    BertLayers of bert_modeling.py:131): forward, the gradient reaching the dense output through the mask, the residual gradient and
    the LayerNorm parameter gradients against fp32 PyTorch under the SAME keep-mask (ecamp_dropout_mask, element index row*cols + col)."""
    o = ops()
    seed, offset, eps = 987654321, 17, 1e-12
    x = rnd(gen(rows, cols, seed=1), dtype)
    res = rnd(gen(rows, cols, seed=2), dtype)
    g = 1 + 0.1 * gen(cols, seed=3)
    b = 0.1 * gen(cols, seed=4)
    dy = rnd(gen(rows, cols, seed=5), dtype)
    keep = o.dropout_mask((rows, cols), dev, p, seed, offset).float().cpu()
    assert abs(keep.mean().item() - (1 - p)) < 0.03
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z_ref = xr * keep / (1 - p) + rr
    y_ref = F.layer_norm(z_ref, (cols,), gr, br, eps)
    y_ref.backward(dy)
    y, z, mean, rstd = o.layernorm_fwd(x.to(dev, dtype), g.to(dev), b.to(dev), eps, residual=res.to(dev, dtype), drop_p=p, seed=seed, offset=offset)
    tol = TOL[dtype]
    check("ln dropout z", z, z_ref, tol)
    check("ln dropout y", y, y_ref, tol * 2)
    gg, gb = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dz, dxd = o.layernorm_bwd(dy.to(dev, dtype), z, mean, rstd, g.to(dev), gg, gb, drop_p=p, seed=seed, offset=offset, want_drop=True)
    check("ln dropout d residual", dz, rr.grad, tol * 2)
    check("ln dropout d dense-out", dxd, xr.grad, tol * 2)
    ptol = 1e-2 if dtype != torch.float32 else 2e-5
    check("ln dropout dgamma", gg, gr.grad, ptol)
    check("ln dropout dbeta", gb, br.grad, ptol)


# ------------------------------------------------------------------------------------------------ image side
@pytest.mark.parametrize("Hs,Hd", [(448, 224), (64, 32), (48, 20), (40, 40)])
def test_bicubic_from_uint8_crops_equals_the_f32_schema(dev, Hs, Hd):
    """Compact image schema: ecamp_bicubic_resize_u8 on uint8 [B,Hs,Hs] grayscale crops gives, bit for bit, what ecamp_bicubic_resize
    gives on the reference's f32 [B,3,Hs,Hs] image (Grayscale(3) + ToTensor + Normalize, pretrain_datasets.py:50-52): the exact-2x path,
    an arbitrary ratio and no resize at all."""
    from ecamp_amd.data import normalise_u8
    o = ops()
    u8 = torch.randint(0, 256, (3, Hs, Hs), generator=torch.Generator().manual_seed(Hs), dtype=torch.uint8)
    u8[0, 0, :8] = torch.tensor([0, 255, 1, 254, 127, 128, 0, 255], dtype=torch.uint8)
    f32 = normalise_u8(u8)
    a = o.bicubic_resize(u8.to(dev), Hd, Hd)
    b = o.bicubic_resize(f32.to(dev), Hd, Hd) if Hs != Hd else f32.to(dev)
    assert a.shape == (3, 3, Hd, Hd) and a.dtype == torch.float32
    if Hs == 2 * Hd or Hs == Hd:      # the hot path's ratio (and no resize): identical bits
        assert torch.equal(a, b), float((a - b).abs().max())
    else:                             # any other ratio: the same taps and weights; the compiler may contract the two kernels' sums differently
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    assert torch.equal(a[:, 0], a[:, 1]) and torch.equal(a[:, 0], a[:, 2])


def test_bicubic(dev):
    o = ops()
    x = gen(3, 3, 64, 64, seed=1)
    check("bicubic 2x down", o.bicubic_resize(x.to(dev), 32, 32), F.interpolate(x, size=[32, 32], mode="bicubic", align_corners=False), 2e-6)
    x = gen(2, 3, 448, 448, seed=2)
    check("bicubic 448->224", o.bicubic_resize(x.to(dev), 224, 224), F.interpolate(x, size=[224, 224], mode="bicubic", align_corners=False), 2e-6)
    x = gen(1, 2, 40, 52, seed=3)
    check("bicubic generic", o.bicubic_resize(x.to(dev), 17, 33), F.interpolate(x, size=[17, 33], mode="bicubic", align_corners=False), 1e-5)


@pytest.mark.parametrize("L,ratio", [(196, 0.75), (784, 0.75), (196, 0.0), (16, 0.5)])
def test_mask_indices(dev, L, ratio):
    o = ops()
    B = 5
    noise = torch.rand(B, L, generator=torch.Generator().manual_seed(3))
    noise[0, 3] = noise[0, 7]  # a tie: stable order must hold
    len_keep = int(L * (1 - ratio))
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    mask = torch.ones(B, L)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    r, kk, m = o.mask_indices(noise.to(dev), len_keep)
    assert (r.cpu().long() == ids_restore).all()
    assert (kk.cpu().long() == ids_shuffle[:, :len_keep]).all()
    assert (m.cpu() == mask).all()


@pytest.mark.parametrize("dtype", DT)
def test_patch_embed_path(dev, dtype):
    """im2col(visible) + GEMM + assemble == conv2d patch-embed + pos + gather + cls concat (model_ecamp.py:218-230)."""
    o = ops()
    B, R, p, D = 3, 64, 16, 192
    G = R // p
    L, Lk = G * G, 4
    imgs = rnd(gen(B, 3, R, R, seed=1), dtype)
    w = rnd(gen(D, 3, p, p, seed=2, scale=0.05), dtype)
    b, cls, pos = gen(D, seed=3), gen(1, 1, D, seed=4), gen(1, L + 1, D, seed=5)
    ids_keep = torch.stack([torch.randperm(L, generator=torch.Generator().manual_seed(i))[:Lk] for i in range(B)])
    x = F.conv2d(imgs, w, b, stride=p).flatten(2).transpose(1, 2) + pos[:, 1:]
    x = torch.gather(x, 1, ids_keep[:, :, None].expand(-1, -1, D))
    ref = torch.cat([(cls + pos[:, :1]).expand(B, -1, -1), x], 1)
    ik = ids_keep.int().to(dev)
    cols = o.im2col_gather(imgs.to(dev), ik, p, dtype)
    assert (cols.view(B, Lk + 1, -1)[:, 0] == 0).all()
    y = o.linear_fwd(cols, w.view(D, -1).to(dev, dtype), b.to(dev))
    o.assemble_tokens_(y, cls.to(dev), pos.to(dev), ik, B, Lk, D)
    check("patch-embed path", y.view(B, Lk + 1, D), ref, TOL[dtype])


@pytest.mark.parametrize("dtype", DT)
def test_unshuffle(dev, dtype):
    o = ops()
    B, L, Lk, D = 3, 16, 4, 64
    y = rnd(gen(B, Lk + 1, D, seed=1), dtype)
    mtok, dpos = gen(1, 1, D, seed=2), gen(1, L + 1, D, seed=3)
    noise = torch.rand(B, L, generator=torch.Generator().manual_seed(4))
    ids_shuffle = torch.argsort(noise, 1)
    ids_restore = torch.argsort(ids_shuffle, 1)
    ids_keep = ids_shuffle[:, :Lk]
    yr, mr = y.clone().requires_grad_(True), mtok.clone().requires_grad_(True)
    x_ = torch.cat([yr[:, 1:], mr.expand(B, L - Lk, D)], 1)
    x_ = torch.gather(x_, 1, ids_restore[:, :, None].expand(-1, -1, D))
    ref = torch.cat([yr[:, :1], x_], 1) + dpos
    dxd = rnd(gen(B, L + 1, D, seed=5), dtype)
    ref.backward(dxd)
    xd = o.unshuffle_fwd(y.to(dev, dtype), ids_restore.int().to(dev), mtok.to(dev), dpos.to(dev), B, L, Lk, D)
    check("unshuffle fwd", xd, ref, TOL[dtype])
    gm = torch.zeros(D, device=dev)
    dy = o.unshuffle_bwd(dxd.to(dev, dtype), ids_restore.int().to(dev), ids_keep.int().to(dev), gm, B, L, Lk, D)
    check("unshuffle dy", dy, yr.grad, 1e-6)
    check("unshuffle dmask_token", gm, mr.grad.view(-1), 1e-5)


@pytest.mark.parametrize("R,win", [(64, 3), (32, 2), (96, 4)])
def test_sr_head_matrix_core_mode(dev, R, win):
    """mode 1 of ecamp_sr_fwd/bwd (bf16 4x4x4 MFMA stencils, used when compute_dtype is bf16) against torch autograd of the reference
    SR head (model_ecamp.py:28-46,291-299).  u, c1, ds, dc1 are rounded to bf16: loss within 2e-3, gradients within a few 1e-2 of
    their scale on average; single pixels whose ReLU gate sits within rounding of zero may flip (that is what bf16 activations do)."""
    o = ops()
    B = 3
    pimg, big = gen(B, 3, R, R, seed=2), gen(B, 3, 2 * R, 2 * R, seed=3)
    column, row = torch.tensor([0, 1, 1]), torch.tensor([1, 0, 1])
    ws = [gen(3, 3, 3, 3, seed=5, scale=0.3), gen(3, seed=6, scale=0.1), gen(3, 3, 3, 3, seed=7, scale=0.3), gen(3, seed=8, scale=0.1)]
    pr = pimg.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in ws]
    u = F.interpolate(pr, scale_factor=2, mode="bilinear", align_corners=False)
    sr = F.relu(F.conv2d(F.relu(F.conv2d(u, wr[0], wr[1], padding=1)), wr[2], wr[3], padding=1) + u)
    G = 2 * R // 32
    sm = torch.zeros(B, G, G)
    for i in range(B):
        sm[i, column[i]:column[i] + win, row[i]:row[i] + win] = 1
    spm = torch.kron(sm, torch.ones(32, 32))[:, None].expand(-1, 3, -1, -1)
    loss = 0.5 * ((sr * spm - big * spm) ** 2).sum()
    loss.backward()
    wd = [t.to(dev).contiguous() for t in ws]
    s = torch.zeros(1, device=dev)
    o.sr_fwd(pimg.to(dev), big.to(dev), column.to(dev), row.to(dev), *wd, s, 32, win, 1)
    assert abs(s.item() - 2 * loss.item()) / (2 * loss.item()) < 2e-3
    gw = torch.zeros(168, device=dev)
    dsr = o.sr_bwd(pimg.to(dev), big.to(dev), column.to(dev), row.to(dev), *wd, gw, 32, win, 1).cpu()
    err = (dsr - pr.grad).abs()
    print("sr mfma R=%d: mean |err| / mean |grad| = %.3e, grad-norm rel %.3e" % (R, err.mean() / pr.grad.abs().mean(), abs(dsr.norm() - pr.grad.norm()) / pr.grad.norm()))
    assert err.mean() / pr.grad.abs().mean() < 3e-2
    assert abs(dsr.norm() - pr.grad.norm()) / pr.grad.norm() < 2e-2
    assert (err > 0.25 * pr.grad.abs().max()).float().mean() < 1e-3          # gate flips are rare
    check("d conv1.weight (mfma)", gw[0:81], wr[0].grad.view(-1), 2e-2)
    check("d conv1.bias (mfma)", gw[81:84], wr[1].grad, 2e-2)
    check("d conv2.weight (mfma)", gw[84:165], wr[2].grad.view(-1), 2e-2)
    check("d conv2.bias (mfma)", gw[165:168], wr[3].grad, 2e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("R,win", [(64, 3), (32, 2)])
def test_image_losses_and_sr_head(dev, dtype, R, win):
    """unpatchify + masked MSE + SR head + windowed MSE, forward and backward, vs the reference formulas
    (model_ecamp.py:28-46,153-165,196-215,276-300) differentiated by torch autograd."""
    o = ops()
    B, p = 3, 16
    G = R // p
    L = G * G
    pred = rnd(gen(B, L + 1, p * p * 3, seed=1), dtype)
    imgs, big = gen(B, 3, R, R, seed=2), gen(B, 3, 2 * R, 2 * R, seed=3)
    mask = (torch.rand(B, L, generator=torch.Generator().manual_seed(4)) < 0.75).float()
    column, row = torch.tensor([0, 1, 1][:B]), torch.tensor([1, 0, 1][:B])
    w1, b1, w2, b2 = gen(3, 3, 3, 3, seed=5, scale=0.3), gen(3, seed=6, scale=0.1), gen(3, 3, 3, 3, seed=7, scale=0.3), gen(3, seed=8, scale=0.1)
    pr = pred.clone().requires_grad_(True)
    ws = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    x = pr[:, 1:].reshape(B, G, G, p, p, 3)
    pimg = torch.einsum("nhwpqc->nchpwq", x).reshape(B, 3, R, R)
    u = F.interpolate(pimg, scale_factor=2, mode="bilinear", align_corners=False)
    sr = F.relu(F.conv2d(F.relu(F.conv2d(u, ws[0], ws[1], padding=1)), ws[2], ws[3], padding=1) + u)
    pm = torch.kron(mask.view(B, G, G), torch.ones(p, p))[:, None].expand(-1, 3, -1, -1)
    sm = torch.zeros(B, G, G)
    for i in range(B):
        sm[i, column[i]:column[i] + win, row[i]:row[i] + win] = 1
    spm = torch.kron(sm, torch.ones(2 * p, 2 * p))[:, None].expand(-1, 3, -1, -1)
    mim = F.mse_loss(pimg * pm, imgs * pm)
    res = F.mse_loss(sr * spm, big * spm)
    g_mim, g_res = 0.7, 1.3
    (g_mim * mim + g_res * res).backward()

    sums = torch.zeros(2, device=dev)
    pd = pred.to(dev, dtype)
    pimg_d = o.unpatchify_mim(pd, imgs.to(dev), mask.to(dev), sums[0:], B, R, p)
    check("unpatchify", pimg_d, pimg, 1e-6)
    wd = [t.to(dev).contiguous() for t in (w1, b1, w2, b2)]
    o.sr_fwd(pimg_d, big.to(dev), column.to(dev), row.to(dev), *wd, sums[1:], 2 * p, win)
    n1, n2 = B * 3 * R * R, B * 3 * 4 * R * R
    tol = TOL[dtype]
    check("mim loss", sums[0:1] / n1, mim.view(1), 1e-5)
    check("res loss", sums[1:2] / n2, res.view(1), 1e-5)
    gw = torch.zeros(168, device=dev)
    dsr = o.sr_bwd(pimg_d, big.to(dev), column.to(dev), row.to(dev), *wd, gw, 2 * p, win)
    gmgs = torch.tensor([g_mim * 2 / n1, g_res * 2 / n2], device=dev)
    dpred = o.img_loss_bwd(pimg_d, imgs.to(dev), mask.to(dev), dsr, gmgs, B, R, p, dtype)
    check("d pred (mim + SR branch)", dpred.view(B, L + 1, -1), pr.grad, 1e-2 if dtype != torch.float32 else 1e-4)
    s = g_res * 2 / n2
    gtol = 1e-4  # the fused SR head is f32 in LDS whatever the activation dtype
    check("d conv1.weight", gw[0:81] * s, ws[0].grad.view(-1), gtol)
    check("d conv1.bias", gw[81:84] * s, ws[1].grad, gtol)
    check("d conv2.weight", gw[84:165] * s, ws[2].grad.view(-1), gtol)
    check("d conv2.bias", gw[165:168] * s, ws[3].grad, gtol)


# ------------------------------------------------------------------------------------------------ report side
@pytest.mark.parametrize("dtype", DT)
def test_bert_embeddings(dev, dtype):
    o = ops()
    B, S, H, V = 6, 40, 768, 500
    g0 = torch.Generator().manual_seed(1)
    ids = torch.randint(0, V, (B, S), generator=g0)
    ids[:, 0] = 2
    ids[torch.rand(B, S, generator=g0) < 0.3] = 3
    ids[:, -5:] = 0  # PAD tail
    ty = (torch.rand(B, S, generator=g0) < 0.2).long()
    word, pos, typ = gen(V, H, seed=2, scale=0.5), gen(64, H, seed=3, scale=0.5), gen(2, H, seed=4, scale=0.5)
    g, b = 1 + 0.1 * gen(H, seed=5), 0.1 * gen(H, seed=6)
    de = rnd(gen(B * S, H, seed=7), dtype)
    wr, pr, tr, gr, br = (t.clone().requires_grad_(True) for t in (word, pos, typ, g, b))
    e = F.embedding(ids, wr, padding_idx=0) + F.embedding(ty, tr) + pr[:S][None]
    ref = F.layer_norm(e, (H,), gr, br, 1e-12)
    ref.backward(de.view(B, S, H))
    ed, z, mean, rstd = o.bert_embed_fwd(ids.to(dev), ty.to(dev), word.to(dev), pos.to(dev), typ.to(dev), g.to(dev), b.to(dev), 1e-12, dtype)
    check("bert embed", ed.view(B, S, H), ref, TOL[dtype])
    gw, gp, gt = torch.zeros(V, H, device=dev), torch.zeros(64, H, device=dev), torch.zeros(2, H, device=dev)
    gg, gb = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    o.bert_embed_bwd(de.to(dev, dtype), z, mean, rstd, g.to(dev), ids.to(dev), ty.to(dev), gw, gp, gt, gg, gb, B, S, H)
    gtol = 1e-4 if dtype == torch.float32 else 2e-2
    assert (gw[0] == 0).all(), "PAD row must get no gradient (padding_idx=0)"
    check("d word_embeddings", gw, wr.grad, gtol)
    check("d position_embeddings", gp, pr.grad, gtol)
    check("d token_type_embeddings", gt, tr.grad, gtol)
    check("d LN gamma", gg, gr.grad, gtol)
    check("d LN beta", gb, br.grad, gtol)


@pytest.mark.parametrize("dtype", DT)
def test_bert_embeddings_dropout_matches_pytorch_under_the_same_mask(dev, dtype):
    """HF BertEmbeddings in TRAIN mode (bert_modeling.py:113): dropout(LayerNorm(word + position + type)) and the embedding / LayerNorm
    gradients behind it, against fp32 PyTorch under the SAME keep-mask (ecamp_dropout_mask over [B*S, H])."""
    o = ops()
    B, S, H, V, p = 4, 48, 768, 300, 0.1
    seed, offset = 55555, 3
    g0 = torch.Generator().manual_seed(2)
    ids = torch.randint(1, V, (B, S), generator=g0)
    ids[:, 0] = 2
    ids[:, -7:] = 0
    ty = torch.zeros(B, S, dtype=torch.long)
    word, pos, typ = gen(V, H, seed=2, scale=0.5), gen(64, H, seed=3, scale=0.5), gen(2, H, seed=4, scale=0.5)
    g, b = 1 + 0.1 * gen(H, seed=5), 0.1 * gen(H, seed=6)
    de = rnd(gen(B * S, H, seed=7), dtype)
    keep = o.dropout_mask((B * S, H), dev, p, seed, offset).float().cpu().view(B, S, H)
    wr, pr, tr, gr, br = (t.clone().requires_grad_(True) for t in (word, pos, typ, g, b))
    e = F.embedding(ids, wr, padding_idx=0) + F.embedding(ty, tr) + pr[:S][None]
    ref = F.layer_norm(e, (H,), gr, br, 1e-12) * keep / (1 - p)
    ref.backward(de.view(B, S, H))
    ed, z, mean, rstd = o.bert_embed_fwd(ids.to(dev), ty.to(dev), word.to(dev), pos.to(dev), typ.to(dev), g.to(dev), b.to(dev), 1e-12, dtype,
                                         drop_p=p, seed=seed, offset=offset)
    check("bert embed (dropout)", ed.view(B, S, H), ref, TOL[dtype])
    gw, gp, gt = torch.zeros(V, H, device=dev), torch.zeros(64, H, device=dev), torch.zeros(2, H, device=dev)
    gg, gb = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    o.bert_embed_bwd(de.to(dev, dtype), z, mean, rstd, g.to(dev), ids.to(dev), ty.to(dev), gw, gp, gt, gg, gb, B, S, H, drop_p=p, seed=seed, offset=offset)
    gtol = 1e-4 if dtype == torch.float32 else 2e-2
    check("d word_embeddings (dropout)", gw, wr.grad, gtol)
    check("d position_embeddings (dropout)", gp, pr.grad, gtol)
    check("d LN gamma (dropout)", gg, gr.grad, gtol)
    check("d LN beta (dropout)", gb, br.grad, gtol)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,V", [(64, 30000), (33, 1000), (7, 256), (5, 64), (3, 20000)])
def test_weighted_cross_entropy(dev, dtype, M, V):
    o = ops()
    logits = rnd(gen(M, V, seed=1) * 3, dtype)
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2))
    w = torch.rand(M, generator=torch.Generator().manual_seed(3)) * 2
    lr = logits.clone().requires_grad_(True)
    loss = (F.cross_entropy(lr, labels, reduction="none") * w).mean()
    loss.backward()
    ld = logits.to(dev, dtype)
    s = torch.zeros(1, device=dev)
    o.ce_fwd_bwd_(ld, labels.to(dev), w.to(dev), s)
    check("mlm loss", s / M, loss.view(1), 1e-5)
    check("d logits", ld, lr.grad, TOL[dtype] if dtype != torch.float32 else 1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_cross_entropy_gradient_gain_at_the_benchmark_row_count(dev, dtype):
    """The MLM head's rows at configs[1] (M = 256 x 128 = 32768): d(mean CE)/d logit = w (p - onehot) / M is ~1e-9 off the label column --
    below IEEE half's smallest subnormal (6e-8): written as it is, the softmax part of the gradient is ZERO in the f16 build (the run
    learns from the one-hot column alone; `tools/dtype_trajectory.py` saw the MLM loss fall 40 % slower).  MlmHeadFn therefore asks for
    gain = 256 M and divides it out of the upstream gradient; bfloat16's eight exponent bits need none (gain 1 == the round-5 call)."""
    o = ops()
    M, V = 32768, 512
    logits = rnd(gen(M, V, seed=1) * 3, dtype)
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2))
    w = torch.rand(M, generator=torch.Generator().manual_seed(3)) * 2
    lr = logits.clone().requires_grad_(True)
    (F.cross_entropy(lr, labels, reduction="none") * w).mean().backward()
    gain = 256.0 * M if dtype == torch.float16 else 1.0
    ld, s = logits.to(dev, dtype), torch.zeros(1, device=dev)
    o.ce_fwd_bwd_(ld, labels.to(dev), w.to(dev), s, gain=gain)
    check("d logits x gain", ld.float() / gain, lr.grad, TOL[dtype])
    off = torch.ones(M, V, dtype=torch.bool)
    off[torch.arange(M), labels] = False
    check("d logits off the label column", (ld.float().cpu() / gain)[off], lr.grad[off], TOL[dtype])
    if dtype == torch.float16:
        l1, s1 = logits.to(dev, dtype), torch.zeros(1, device=dev)
        o.ce_fwd_bwd_(l1, labels.to(dev), w.to(dev), s1)
        lost = float((l1.float().cpu()[off] == 0).float().mean())
        print("  without the gain %.1f %% of the off-label gradient entries are zero in IEEE half" % (100 * lost))
        assert lost > 0.5


@pytest.mark.parametrize("dtype", DT)
def test_weighted_cross_entropy_ignore_index(dev, dtype):
    """CrossEntropyLoss's ignore_index (-100, honoured by the reference's loss at bert_modeling.py:212): zero loss and a zero
    gradient row, still counted in the mean's denominator; any label outside [0, V) is treated the same and never used as an index."""
    o = ops()
    M, V = 40, 30000
    logits = rnd(gen(M, V, seed=1) * 3, dtype)
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2))
    labels[::3] = -100
    w = torch.rand(M, generator=torch.Generator().manual_seed(3)) * 2
    lr = logits.clone().requires_grad_(True)
    loss = (F.cross_entropy(lr, labels, reduction="none") * w).mean()
    loss.backward()
    ld = logits.to(dev, dtype)
    s = torch.zeros(1, device=dev)
    o.ce_fwd_bwd_(ld, labels.to(dev), w.to(dev), s)
    check("mlm loss (ignored labels)", s / M, loss.view(1), 1e-5)
    check("d logits (ignored labels)", ld, lr.grad, TOL[dtype] if dtype != torch.float32 else 1e-5)
    assert float(ld[::3].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ small ops / optimizer
@pytest.mark.parametrize("dtype", DT)
def test_small_ops(dev, dtype):
    o = ops()
    B, S, H = 3, 50, 768
    x, g = rnd(gen(B, S, H, seed=1), dtype), rnd(gen(B, H, seed=2), dtype)
    xd, gd = x.to(dev, dtype), g.to(dev, dtype)
    tol = TOL[dtype]
    check("add", o.add(xd, xd), 2 * x, tol)
    check("bcast_add", o.bcast_add(xd, gd), x + g[:, None], tol)
    check("seq_sum (gap mean)", o.seq_sum(xd, 1, S, 1.0 / (S - 1)), x[:, 1:].mean(1), tol)
    a_dev = torch.tensor([0.3], device=dev)
    check("scale_ (x *= alpha * alpha_dev, in place)", o.scale_(xd.clone(), alpha=2.0, alpha_dev=a_dev), 0.6 * x, tol)
    y = torch.full((B, S, H), 7.0, device=dev, dtype=dtype)
    o.seq_bcast(gd, y, 1, S, 0.5, 0)
    ref = torch.zeros(B, S, H)
    ref[:, 1:] = 0.5 * g[:, None]
    check("seq_bcast set", y, ref, tol)
    o.seq_bcast(gd, y, 0, S, 1.0, 1)
    check("seq_bcast add", y, ref + g[:, None], tol)
    out = torch.zeros(H, device=dev)
    o.colsum(xd.view(B * S, H), out, 1.0, S, 0, 1)
    check("colsum cls rows", out, x[:, 0].sum(0), 1e-2 if dtype != torch.float32 else 1e-5)
    out = torch.zeros(H, device=dev)
    o.colsum(xd.view(B * S, H), out, 1.0, S, 1, S)
    check("colsum non-cls rows", out, x[:, 1:].sum((0, 1)), 1e-2 if dtype != torch.float32 else 1e-5)
    u = o.uniform((4, 196), dev, 42, 0)
    assert 0.0 <= u.min().item() and u.max().item() < 1.0 and abs(u.mean().item() - 0.5) < 0.05
    assert (u != o.uniform((4, 196), dev, 42, 1)).any() and (u == o.uniform((4, 196), dev, 42, 0)).all()
    f = gen(1000, seed=3).to(dev)
    h = torch.empty(1000, device=dev, dtype=h16())
    o.cast(f, h)
    assert (h.cpu() == f.cpu().to(h16())).all(), "f32->bf16 must be round-to-nearest-even"


@pytest.mark.usefixtures("both_halves")
def test_adamw_and_gradnorm(dev):
    o = ops()
    n = 4096 * 3
    p0, g0 = gen(n, seed=1), gen(n, seed=2) * 0.01
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p16 = torch.empty(n, device=dev, dtype=h16())
    for step in range(1, 4):
        g = g0 * step
        pr.grad = g.clone()
        opt.step()
        o.adamw(p, g.to(dev), m, v, p16, 1.5e-4, 0.9, 0.95, 1e-8, 0.05, step)
    check("adamw 3 steps (params)", p, pr.detach(), 1e-6)
    check("adamw 3 steps (update; f32 cancellation-limited)", p - p0.to(dev), pr.detach() - p0, 3e-3)
    assert (p16.cpu() == p.cpu().to(h16())).all()
    s = torch.zeros(1, device=dev)
    o.sumsq(g0.to(dev), s)
    check("grad norm", s.sqrt(), g0.norm().view(1), 1e-5)


@pytest.mark.parametrize("B,H,Tq,Tk,hd,masked", [(1, 16, 785, 785, 32, False), (2, 4, 300, 300, 64, True), (1, 3, 70, 520, 128, False)])
def test_attention_long_sequence_f32(dev, B, H, Tq, Tk, hd, masked):
    """Tk > 256 in the exact-f32 parity path (attn_long_* kernels) against plain PyTorch fp32, forward and backward, with a key mask."""
    o = ops()
    D = H * hd
    q, k, v, do = gen(B, Tq, D, seed=1), gen(B, Tk, D, seed=2), gen(B, Tk, D, seed=3), gen(B, Tq, D, seed=4)
    km = (torch.arange(Tk)[None, :] < torch.tensor([Tk - 37, Tk][:B])[:, None]).int() if masked else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    sp = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)
    out = ref_attn(sp(qr, Tq), sp(kr, Tk), sp(vr, Tk), hd ** -0.5, km).permute(0, 2, 1, 3).reshape(B, Tq, D)
    out.backward(do)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    kmd = km.to(dev) if km is not None else None
    qs, ks = (Tq * D, D, hd), (Tk * D, D, hd)
    og, lse = o.attn_fwd(qd, kd, vd, B, H, Tq, Tk, hd, qs, ks, ks, hd ** -0.5, key_mask=kmd)
    check("long f32 attn out", og, out, 2e-5)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    o.attn_bwd(qd, kd, vd, og, do.to(dev), lse, dq, dk, dv, B, H, Tq, Tk, hd, qs, ks, ks, qs, ks, ks, hd ** -0.5, key_mask=kmd)
    check("long f32 attn dq", dq, qr.grad, 5e-5)
    check("long f32 attn dk", dk, kr.grad, 5e-5)
    check("long f32 attn dv", dv, vr.grad, 5e-5)


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("B,H,Tq,Tk,hd,masked", [(1, 16, 785, 785, 32, False), (2, 4, 300, 300, 64, True), (1, 3, 70, 520, 128, False)])
def test_attention_long_sequence_bf16(dev, B, H, Tq, Tk, hd, masked):
    """Tk > 256 (ViT-L/16 at 448^2: decoder sequence 785) runs the online-softmax forward and the chunk-streaming backward."""
    o = ops()
    dtype = h16()
    D = H * hd
    q, k, v = rnd(gen(B, Tq, D, seed=1), dtype), rnd(gen(B, Tk, D, seed=2), dtype), rnd(gen(B, Tk, D, seed=3), dtype)
    do = rnd(gen(B, Tq, D, seed=4), dtype)
    km = (torch.arange(Tk)[None, :] < torch.tensor([Tk - 37, Tk][:B])[:, None]).int() if masked else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    sp = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)
    out = ref_attn(sp(qr, Tq), sp(kr, Tk), sp(vr, Tk), hd ** -0.5, km).permute(0, 2, 1, 3).reshape(B, Tq, D)
    out.backward(do)
    qd, kd, vd = q.to(dev, dtype), k.to(dev, dtype), v.to(dev, dtype)
    kmd = km.to(dev) if km is not None else None
    qs, ks = (Tq * D, D, hd), (Tk * D, D, hd)
    og, lse = o.attn_fwd(qd, kd, vd, B, H, Tq, Tk, hd, qs, ks, ks, hd ** -0.5, key_mask=kmd)
    check("long attn out", og, out, 2e-2)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    o.attn_bwd(qd, kd, vd, og, do.to(dev, dtype), lse, dq, dk, dv, B, H, Tq, Tk, hd, qs, ks, ks, qs, ks, ks, hd ** -0.5, key_mask=kmd)
    check("long attn dq", dq, qr.grad, 4e-2)
    check("long attn dk", dk, kr.grad, 4e-2)
    check("long attn dv", dv, vr.grad, 4e-2)


# ------------------------------------------------------------------------------------------------ persistent 256^2 GEMM kernel
@pytest.fixture
def q8_always():
    """Route every bf16 GEMM that meets its alignment / size conditions to the round-2 persistent kernel (gemm_q8.h); the automatic
    rule only picks it from 128 tiles of 256x256 up (and hands the 768-wide outputs to the four-wave kernel: switched off here)."""
    o = ops()
    o.set_option("q8_mode", 2)
    o.set_option("q16_mode", 0)
    yield o
    o.set_option("q8_mode", -1)
    o.set_option("q16_mode", -1)


def _q8_count():
    from ecamp_amd import _lib
    return int(_lib.load().ecamp_gemm_q8_launches())


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("M,N,K", [(394, 768, 192), (100, 2304, 768), (512, 128, 3072), (256, 30000, 768), (1000, 1000, 200), (300, 520, 136)])
def test_gemm_q8_fwd_epilogues_ragged(dev, q8_always, M, N, K):
    """test_gemm_fwd_epilogues on the Q8 kernel: ragged M and N (edge tiles, groups of 8 past N), K with a partial last K tile,
    two to 48 K tiles, one or many output tiles per workgroup (N = 30000: 118 tiles)."""
    n0 = _q8_count()
    test_gemm_fwd_epilogues(dev, h16(), M, N, K)
    assert _q8_count() >= n0 + 3, "the Q8 kernel did not run"


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("M,N,K", [(394, 768, 192), (100, 3072, 768), (256, 30000, 768), (1000, 520, 264)])
def test_gemm_q8_dgrad_wgrad_ragged(dev, q8_always, M, N, K):
    """Data gradient (strided weight operand, transpose reads) with gelu' / residual epilogues, weight gradient (both operands
    strided, split-K slabs) and the bias gradient summed inside it, on ragged shapes."""
    n0 = _q8_count()
    test_gemm_dgrad_wgrad(dev, h16(), M, N, K)
    assert _q8_count() >= n0 + 5, "the Q8 kernel did not run"


@pytest.fixture
def q16_always():
    from ecamp_amd import hip_ops
    hip_ops.set_option("q16_mode", 3)
    hip_ops.set_option("q8_mode", 2)
    yield
    hip_ops.set_option("q16_mode", -1)
    hip_ops.set_option("q8_mode", -1)


def _q16_count():
    from ecamp_amd import _lib
    return int(_lib.load().ecamp_gemm_q16_launches())


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("M,N,K", [(394, 768, 192), (1000, 520, 200), (512, 1536, 136), (300, 2304, 768), (2048, 192, 1088), (777 * 8, 768, 264)])
def test_gemm_q16_forward_and_data_gradient_ragged(dev, q16_always, M, N, K):
    """The four-wave v_mfma_f32_16x16x32_bf16 kernel (csrc/gemm_q16.h) forced on ragged shapes: forward form y = x w^T + b (+ residual) and
    data-gradient form dx = dy w (+ residual) against fp32 PyTorch -- edge tiles in M and N (both the 256- and the 192-column tile are
    picked by the shapes: N = 768 / 192 / 1536 take 192, N = 520 / 2304 take 256), partial last K tiles, one tile per workgroup and several;
    the launch counter proves the kernel ran."""
    o = ops()
    dt = h16()
    x = rnd(gen(M, K, seed=1), dt)
    w = rnd(gen(N, K, seed=2) * K ** -0.5, dt)
    b = gen(N, seed=3)
    res = rnd(gen(M, N, seed=4), dt)
    dy = rnd(gen(M, N, seed=5), dt)
    resx = rnd(gen(M, K, seed=6), dt)
    xd, wd, bd, rd, dyd, rxd = x.to(dev, dt), w.to(dev, dt), b.to(dev), res.to(dev, dt), dy.to(dev, dt), resx.to(dev, dt)
    n0 = _q16_count()
    y0 = o.linear_fwd(xd, wd, bd)
    y1 = o.linear_fwd(xd, wd, bd, residual=rd)
    y2 = o.linear_fwd(xd, wd)
    check("q16 fwd + bias", y0, x @ w.T + b, 2e-2)
    check("q16 fwd + bias + residual", y1, x @ w.T + b + res, 2e-2)
    check("q16 fwd plain", y2, x @ w.T, 2e-2)
    if K % 8 == 0 and N % 8 == 0:
        d0 = o.linear_dgrad(dyd, wd)
        d1 = o.linear_dgrad(dyd, wd, residual=rxd)
        check("q16 dgrad", d0, dy @ w, 2e-2)
        check("q16 dgrad + residual", d1, dy @ w + resx, 2e-2)
        assert _q16_count() - n0 == 5
    else:
        assert _q16_count() - n0 == 3


@pytest.mark.usefixtures("both_halves")
def test_gemm_q16_is_deterministic_and_selected_for_the_768_wide_outputs(dev):
    """Default kernel selection ("q16_mode" 1): the model's 768-wide outputs at B = 256 (150 / 384 tiles of 256 x 256: 41 % / 25 % of the last
    round idle) run on the four-wave kernel's 256 x 192 tile, the other shapes stay on the eight-wave kernel; repeated launches are
    bit-identical (no atomics, a race screen of the DMA / barrier protocol)."""
    o = ops()
    dt = h16()
    for M, N, K, want in ((12800, 768, 3072, 1), (32768, 768, 768, 1), (12800, 3072, 768, 0), (32768, 1536, 768, 0)):
        x = rnd(gen(M, K, seed=1), dt).to(dev, dt)
        w = rnd(gen(N, K, seed=2) * K ** -0.5, dt).to(dev, dt)
        dy = rnd(gen(M, N, seed=3), dt).to(dev, dt)
        n0 = _q16_count()
        y = [o.linear_fwd(x, w) for _ in range(3)]
        d = [o.linear_dgrad(dy, w) for _ in range(3)]
        torch.cuda.synchronize()
        # forward: output width N; data gradient: output width K
        assert (_q16_count() - n0) == 3 * want + 3 * (1 if K == 768 else 0), (M, N, K, _q16_count() - n0)
        assert torch.equal(y[0], y[1]) and torch.equal(y[0], y[2]) and torch.equal(d[0], d[1]) and torch.equal(d[0], d[2])
        ref = (x[:64].float() @ w.float().T)
        assert (y[0][:64].float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("M,N,K", [(12800, 768, 3072), (1000, 520, 264), (50432, 512, 2048)])
def test_gemm_q8_data_gradient_one_tile_per_workgroup_is_bit_identical(dev, q8_always, M, N, K):
    """`q8_bwd_grid` (what the data-parallel wrapper sets): the data-gradient form on one workgroup per output tile, on half as many
    workgroups as tiles and on the persistent grid computes every tile with the same instruction sequence -- the three outputs are
    the same bits, with the gelu' / residual epilogues as well."""
    o = q8_always
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(M, K, generator=g).to(dev, h16())          # dX[M, N] = dY[M, K] W[K, N]  (W stored [K, N]: strided operand)
    w = (torch.randn(K, N, generator=g) * K ** -0.5).to(dev, h16())
    pre = torch.randn(M, N, generator=g).to(dev, h16())
    res = torch.randn(M, N, generator=g).to(dev, h16())
    outs = []
    tiles = -(-M // 256) * -(-N // 256)
    try:
        for grid in (0, 1 << 20, max(1, tiles // 2)):
            o.set_option("q8_bwd_grid", grid)
            n0 = _q8_count()
            a = o.linear_dgrad(dy, w)
            b = o.linear_dgrad(dy, w, gmul=pre)
            c = o.linear_dgrad(dy, w, residual=res)
            assert _q8_count() >= n0 + 3, "the Q8 kernel did not run"
            outs.append((a, b, c))
    finally:
        o.set_option("q8_bwd_grid", 0)
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)


@pytest.mark.usefixtures("both_halves")
@pytest.mark.parametrize("M,N,K", [(12800, 768, 3072), (1000, 520, 264), (32768, 768, 1536)])
def test_gemm_q16_data_gradient_one_tile_per_workgroup_is_bit_identical(dev, q16_always, M, N, K):
    """with the default kernel selection the 768-wide data gradients run on the FOUR-wave kernel, so `q8_bwd_grid` must reach
    that launch too: persistent grid, one workgroup per tile and half as many workgroups as tiles give the same bits (plain and residual
    epilogues -- the forms the four-wave kernel has), and the four-wave kernel is what ran."""
    from ecamp_amd import hip_ops as o
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(M, K, generator=g).to(dev, h16())
    w = (torch.randn(K, N, generator=g) * K ** -0.5).to(dev, h16())
    res = torch.randn(M, N, generator=g).to(dev, h16())
    outs = []
    tiles = -(-M // 256) * -(-N // 192)
    try:
        for grid in (0, 1 << 20, max(1, tiles // 2)):
            o.set_option("q8_bwd_grid", grid)
            n0 = _q16_count()
            a = o.linear_dgrad(dy, w)
            c = o.linear_dgrad(dy, w, residual=res)
            assert _q16_count() == n0 + 2, "the four-wave kernel did not run"
            outs.append((a, c))
    finally:
        o.set_option("q8_bwd_grid", 0)
    ref = dy[:64].float() @ w.float()
    assert (outs[0][0][:64].float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)


def test_gemm_full_size_kernels_agree(dev):
    """BASELINE configs[1] sizes (timm Mlp.fc1 of the encoder at B=256: 12800 x 3072 x 768): forward with bias + GELU + saved
    pre-activation, data gradient through GELU', weight + bias gradient -- the persistent kernel against the 128^2 kernel on the
    same inputs (both accumulate in f32; only the summation order differs), and both against a float64 sample of rows."""
    o = ops()
    M, N, K = 12800, 3072, 768
    x = (torch.randn(M, K, generator=torch.Generator().manual_seed(1))).to(dev, h16())
    w = (torch.randn(N, K, generator=torch.Generator().manual_seed(2)) * K ** -0.5).to(dev, h16())
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(dev)
    dy = torch.randn(M, N, generator=torch.Generator().manual_seed(4)).to(dev, h16())
    res = {}
    for mode in (0, 8):   # 128^2 kernel, persistent 256^2 kernel (Q8)
        o.set_option("q8_mode", 2 if mode == 8 else 0)
        y, pre = o.linear_fwd(x, w, b, act=1, save_pre=True)
        y2, der = o.linear_fwd(x, w, b, act=2, save_pre=True)   # the production form: gelu' saved in place of the pre-activation
        assert torch.equal(y, y2)
        pr = pre[::97].float().requires_grad_(True)
        F.gelu(pr).sum().backward()
        assert float((der[::97].float() - pr.grad).abs().max()) <= 2.0 ** -8 * 1.13 + 1e-6, "saved gelu' (mode %d)" % mode
        dx = o.linear_dgrad(dy, w)
        gw, gb = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        o.linear_wgrad(dy, x, gw, gb=gb)
        res[mode] = (y.float(), pre.float(), dx.float(), gw, gb)
    o.set_option("q8_mode", -1)
    for other, label in ((8, "Q8"),):
        for name, a, c, tol in zip(("y", "pre", "dx", "gw", "gb"), res[0], res[other], (1e-2, 1e-2, 1e-2, 2e-3, 2e-3)):
            err = float((a - c).abs().max() / c.abs().max())
            print("  full-size %-3s 128^2 vs %s rel %.3e" % (name, label, err))
            assert err < tol, (label, name)
    rows = torch.arange(0, M, 997, device=dev)
    ref_pre = x[rows].double() @ w.double().t() + b.double()
    assert float((res[8][1][rows].double() - ref_pre).abs().max() / ref_pre.abs().max()) < 1e-2
    cols = torch.arange(0, N, 211, device=dev)
    ref_gw = dy[:, cols].double().t() @ x.double()
    assert float((res[8][3][cols].double() - ref_gw).abs().max() / ref_gw.abs().max()) < 2e-3
    assert float((res[8][4].double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max()) < 2e-3


# ------------------------------------------------------------------------------------------------ fp8 forward (BASELINE configs[4])
def _e4m3_ref(x):
    """Host restatement of per-tensor e4m3 quantisation: (dequantised values, scale) via torch.float8_e4m3fn (OCP, RNE)."""
    amax = x.abs().max().clamp_min(1e-20)
    scale = amax.float() * (1.0 / 448.0)
    q = (x.float() / scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("dtype", DT)
def test_quantize_fp8_is_bit_exact(dev, dtype):
    o = ops()
    x = rnd(gen(300, 512, seed=11, scale=3.0), dtype)
    x[0, 0] = 0.0
    q, s = o.quantize_fp8(x.to(dev, dtype).contiguous())
    qr, sr = _e4m3_ref(x)
    assert abs(s.item() - sr.item()) <= 1e-6 * sr.item()
    same = (q.cpu().view(torch.uint8) == qr.view(torch.uint8))
    # the only admissible differences: x/scale on the device is x * (1/scale) -- a tie can round the other way by one code
    assert same.float().mean().item() > 0.999
    d = (q.cpu().view(torch.float8_e4m3fn).float() - qr.float()).abs() / qr.float().abs().clamp_min(2 ** -6)
    assert d.max().item() <= 0.126


@pytest.mark.parametrize("M,N,K", [(394, 768, 256), (100, 2304, 768), (512, 132, 3072), (130, 512, 48)])
def test_gemm_fp8_forward_epilogues(dev, M, N, K):
    """fp8 forward GEMM == f32 matmul of the DEQUANTISED operands (the only rounding left is the bf16 output / f32 summation
    order), for every forward epilogue; and within the e4m3 quantisation noise of the unquantised product."""
    o = ops()
    dtype = h16()
    x, w, b, r = rnd(gen(M, K, seed=1), dtype), rnd(gen(N, K, seed=2, scale=K ** -0.5), dtype), gen(N, seed=3), rnd(gen(M, N, seed=4), dtype)
    xd, wd, bd, rd = x.to(dev, dtype), w.to(dev, dtype), b.to(dev), r.to(dev, dtype)
    w8, ws = o.quantize_fp8(wd)
    xq, xs = o.quantize_fp8(xd)
    xdq = xq.cpu().view(torch.float8_e4m3fn).float() * xs.item()
    wdq = w8.cpu().view(torch.float8_e4m3fn).float() * ws.item()
    ref = xdq @ wdq.T
    tol = TOL[dtype]
    check("fp8 linear", o.linear_fwd_fp8(xd, w8, ws, bd), ref + b, tol)
    check("fp8 linear+residual", o.linear_fwd_fp8(xd, w8, ws, None, residual=rd), ref + r, tol)
    y, pre = o.linear_fwd_fp8(xd, w8, ws, bd, act=1, save_pre=True)
    check("fp8 linear pre", pre, ref + b, tol)
    check("fp8 linear gelu", y, F.gelu(rnd(ref + b, dtype)), tol)
    # (the 128^2 kernel has no third output: ecamp_gemm_fp8 quantises C in a pass of its own -- same bytes as the persistent kernel's epilogue)
    sc, sl, sl0 = torch.tensor([0.02], device=dev), torch.zeros(512, device=dev), torch.zeros(512, device=dev)
    y2, pre2, y8 = o.gemm_fp8(xq, xs, w8, ws, bd, act=1, save_pre=True, q8_site=(sc, sl))
    assert torch.equal(y2, y) and torch.equal(y8, o.quantize_fp8_site(y, sc, sl0, True)) and sl.max().item() == y.float().abs().max().item()
    exact = x @ w.T + b
    e = ((o.linear_fwd_fp8(xd, w8, ws, bd).float().cpu() - exact).norm() / exact.norm()).item()
    print("  fp8 vs unquantised product: relative Frobenius error %.3e" % e)
    assert e < 6e-2


@pytest.mark.parametrize("M,N,K", [(394, 768, 256), (1000, 520, 400), (700, 2304, 768), (513, 136, 3072)])
def test_gemm_fp8_persistent_kernel_forced_on_ragged_shapes(dev, M, N, K):
    """The persistent 256 x 256 x 128 e4m3 kernel (gemm_q8.h, F8: the kernel configs[4] runs at its full size) forced on small ragged
    shapes: edge tiles in M and N, a partial last K tile (K = 400 = 3 x 128 + 16), every forward epilogue, against the f32 product of the
    DEQUANTISED operands; a launch counter proves which kernel ran."""
    o = ops()
    from ecamp_amd import _lib
    lib = _lib.load()
    dtype = h16()
    x, w, b, r = rnd(gen(M, K, seed=1), dtype), rnd(gen(N, K, seed=2, scale=K ** -0.5), dtype), gen(N, seed=3), rnd(gen(M, N, seed=4), dtype)
    xd, wd, bd, rd = x.to(dev, dtype), w.to(dev, dtype), b.to(dev), r.to(dev, dtype)
    w8, ws = o.quantize_fp8(wd)
    xq, xs = o.quantize_fp8(xd)
    ref = (xq.cpu().view(torch.float8_e4m3fn).float() * xs.item()) @ (w8.cpu().view(torch.float8_e4m3fn).float() * ws.item()).T
    tol = TOL[dtype]
    o.set_option("q8_mode", 2)
    try:
        n0 = lib.ecamp_gemm_f8_q8_launches()
        check("fp8 q8 linear", o.linear_fwd_fp8(xd, w8, ws, bd), ref + b, tol)
        check("fp8 q8 linear (no bias)", o.linear_fwd_fp8(xd, w8, ws, None), ref, tol)
        check("fp8 q8 linear+residual", o.linear_fwd_fp8(xd, w8, ws, None, residual=rd), ref + r, tol)
        y, pre = o.linear_fwd_fp8(xd, w8, ws, bd, act=1, save_pre=True)
        check("fp8 q8 linear pre", pre, ref + b, tol)
        check("fp8 q8 linear gelu", y, F.gelu(rnd(ref + b, dtype)), tol)
        assert lib.ecamp_gemm_f8_q8_launches() - n0 == 4
        # the GELU epilogue's third output: the e4m3 copy of y for the next dense layer, quantised in the epilogue with that layer's scale
        sc, sl, sl0 = torch.tensor([0.02], device=dev), torch.zeros(512, device=dev), torch.zeros(512, device=dev)
        y2, pre2, y8 = o.gemm_fp8(xq, xs, w8, ws, bd, act=1, save_pre=True, q8_site=(sc, sl))
        assert torch.equal(y2, y) and torch.equal(pre2, pre)
        assert torch.equal(y8, o.quantize_fp8_site(y, sc, sl0, True))
        assert sl.view(16, 32)[:, 0].max().item() == y.float().abs().max().item()
        assert lib.ecamp_gemm_f8_q8_launches() - n0 == 5
        # act = 2, the saved derivative: same activation and e4m3 copy, gelu'(rounded pre-activation) where the pre-activation was
        y3, der, y83 = o.gemm_fp8(xq, xs, w8, ws, bd, act=2, save_pre=True, q8_site=(sc, torch.zeros(512, device=dev)))
        assert torch.equal(y3, y) and torch.equal(y83, y8)
        pr = pre.float().cpu().requires_grad_(True)
        F.gelu(pr).sum().backward()
        check("fp8 q8 saved gelu'", der, pr.grad, 2.0 ** -8)
    finally:
        o.set_option("q8_mode", -1)


def test_fp8_delayed_scaling_kernels(dev):
    """Delayed per-tensor scaling (configs[4], round 4): ecamp_quant_fp8_delayed quantises with a GIVEN scale in one pass and leaves
    max|x| in the site's amax slots; ecamp_fp8_roll turns the slots into the next scale and clears them (a site nobody fed keeps its
    scale); ecamp_layernorm_fwd_q8's e4m3 copy is bit-identical to quantising its bf16 output afterwards, with and without the fused
    residual + dropout, for 768- and 512-column rows (16-B lane accesses) and a 100-column row (the 4-wide form)."""
    o = ops()
    x = (gen(300, 512, seed=11, scale=3.0)).to(dev, h16())
    scale = torch.tensor([0.0123], device=dev)
    slots = torch.zeros(2 * 512, device=dev)
    q = o.quantize_fp8_site(x, scale, slots[:512], True)
    ref = (x.float().cpu() / 0.0123).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = (q.cpu().view(torch.uint8) == ref.view(torch.uint8)).float().mean().item()
    assert same > 0.999   # x * (1 / scale) on the device against x / scale here: a tie may round the other way by one code
    assert slots[:512].view(16, 32)[:, 0].max().item() == x.float().abs().max().item() and slots[:512].view(16, 32)[:, 1:].abs().max().item() == 0
    scales = torch.tensor([7.0, 9.0], device=dev)
    o.fp8_roll(slots, scales)
    assert abs(scales[0].item() - x.float().abs().max().item() / 448.0) < 1e-7 and scales[1].item() == 9.0 and slots.abs().max().item() == 0
    # first use of a site = the two-pass current scaling; it seeds scale and slot 0
    sc2, sl2 = torch.ones(1, device=dev), torch.zeros(512, device=dev)
    q2 = o.quantize_fp8_site(x, sc2, sl2, False)
    q3, s3 = o.quantize_fp8(x)
    assert torch.equal(q2, q3) and sc2.item() == s3.item() and sl2[0].item() == x.float().abs().max().item()
    for rows, cols in ((260, 768), (130, 512), (50, 100)):
        xx = gen(rows, cols, seed=3).to(dev, h16())
        rr = gen(rows, cols, seed=4).to(dev, h16())
        g, b = (1.0 + 0.1 * gen(cols, seed=5)).to(dev), (0.1 * gen(cols, seed=6)).to(dev)
        for kw in ({}, dict(residual=rr, drop_p=0.1, seed=7, offset=9)):
            sc, sl = torch.tensor([0.011], device=dev), torch.zeros(512, device=dev)
            y, z, mean, rstd, y8 = o.layernorm_fwd(xx, g, b, 1e-6, q8_site=(sc, sl), **kw)
            y0 = o.layernorm_fwd(xx, g, b, 1e-6, **kw)[0]
            assert torch.equal(y, y0)
            sl0 = torch.zeros(512, device=dev)
            assert torch.equal(y8, o.quantize_fp8_site(y0, sc, sl0, True))
            assert sl.view(16, 32)[:, 0].max().item() == y0.float().abs().max().item()


@pytest.mark.usefixtures("both_halves")
def test_gemm_rows_past_2gb_are_split(dev):
    """An output of more than 2 GB (the vocabulary projection at B = 512) runs as two row halves on the persistent kernel; the rows on
    both sides of the seam match the same rows computed as a small GEMM."""
    o = ops()
    M, N, K = 65536, 16400, 128           # 65536 x 16400 bf16 = 2.15 GB
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(dev, h16())
    w = (torch.randn(N, K, generator=torch.Generator().manual_seed(2)) * K ** -0.5).to(dev, h16())
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(dev)
    n0 = _q8_count()
    y = o.linear_fwd(x, w, b)
    assert _q8_count() >= n0 + 2, "the two halves did not run on the Q8 kernel"
    for r0 in (0, 32768 - 8, 32768, 65536 - 16):
        ref = (x[r0:r0 + 16].float() @ w.float().t() + b).to(h16()).float()
        assert (y[r0:r0 + 16].float() - ref).abs().max() < 3e-2 * ref.abs().max()
    del y


@pytest.mark.usefixtures("both_halves")
def test_gemm_wgrad_contraction_past_2gb_is_split(dev):
    """A weight gradient whose dy operand passes 2 GB (the vocabulary head at B = 512) runs as two calls over halves of the rows, the second
    accumulating: weight and bias gradient against torch on a column sample."""
    o = ops()
    M, N, K = 65536, 16400, 128           # dy [M, N] bf16 = 2.15 GB
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(dev, h16())
    dy = torch.randn(M, N, device=dev, dtype=h16())
    gw = torch.zeros(N, K, device=dev)
    gb = torch.zeros(N, device=dev)
    n0 = _q8_count()
    o.linear_wgrad(dy, x, gw, gb=gb, accumulate=False)
    assert _q8_count() >= n0 + 2, "the two halves did not run on the Q8 kernel"
    cols = torch.tensor([0, 255, 256, 8191, 16399], device=dev)
    ref = dy[:, cols].float().t() @ x.float()
    assert (gw[cols] - ref).abs().max() < 2e-3 * ref.abs().max()
    refb = dy[:, cols].float().sum(0)
    assert (gb[cols] - refb).abs().max() < 2e-3 * refb.abs().max() + 1e-2


@pytest.mark.parametrize("rows,shapes", [(1280, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
                                         (2560, [(1000, 520), (264, 264), (520, 1000)]),
                                         (1000, [(1000, 520), (264, 264), (520, 1000)]),      # rows: no multiple of 64 (partial last K tile)
                                         (12608, [(3072, 1024), (1024, 1024)]),               # ViT-L/448: 197 K tiles (odd)
                                         (12800, [(2304, 768), (768, 768), (3072, 768), (768, 3072)])])
@pytest.mark.usefixtures("both_halves")
def test_wgrad_group_matches_per_layer_weight_gradients(dev, rows, shapes):
    """ecamp_wgrad_group (the weight and bias gradients of a block's linear layers as ONE item-table launch of the Q8 kernel + one
    grouped reduce) against the per-layer GEMMs: K ranges cut across tile boundaries, ragged output shapes, overwrite and accumulate,
    layers with and without a bias, and the production shapes of one encoder block."""
    o = ops()
    gen_ = torch.Generator().manual_seed(rows)
    items, refs = [], []
    for i, (n_out, k_in) in enumerate(shapes):
        dy = (torch.randn(rows, n_out, generator=gen_) * 0.5).to(dev, h16())
        x = torch.randn(rows, k_in, generator=gen_).to(dev, h16())
        acc = i % 2 == 1
        base = torch.randn(n_out, k_in, generator=gen_).to(dev) if acc else torch.zeros(n_out, k_in, device=dev)
        gw, gw_ref = base.clone(), base.clone()
        bias = i != 1
        gb, gb_ref = (torch.ones(n_out, device=dev), torch.ones(n_out, device=dev)) if bias else (None, None)
        o.linear_wgrad(dy, x, gw_ref, gb=gb_ref, accumulate=acc)
        items.append((dy, x, gw, gb, acc))
        refs.append((gw_ref, gb_ref))
    assert o.wgrad_group_supported(items)
    o.wgrad_group(items)
    torch.cuda.synchronize()
    for (dy, x, gw, gb, acc), (gw_ref, gb_ref) in zip(items, refs):
        scale = gw_ref.abs().max()
        assert (gw - gw_ref).abs().max() < 1e-4 * scale + 1e-3, (gw - gw_ref).abs().max()
        if gb is not None:
            assert (gb - gb_ref).abs().max() < 1e-3 * gb_ref.abs().max() + 1e-2
    # a second call accumulates on top of the first where asked to and overwrites elsewhere
    o.wgrad_group(items)
    for (dy, x, gw, gb, acc), (gw_ref, gb_ref) in zip(items, refs):
        o.linear_wgrad(dy, x, gw_ref, gb=gb_ref, accumulate=acc)
        assert (gw - gw_ref).abs().max() < 1e-4 * gw_ref.abs().max() + 1e-3


@pytest.mark.usefixtures("both_halves")
def test_wgrad_group_refuses_a_table_built_for_another_plan(dev):
    """the grouped launch reads counts and offsets from the host-side plan and the items from the caller's device table; a
    process-wide switch flipped through the RAW C ecamp_set_option (which does not clear the Python table cache) changes the plan.
    The call must fail loudly instead of reading the old image at the new offsets."""
    from ecamp_amd import _lib
    o = ops()
    lib = _lib.load()
    rows, shapes = 12800, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    items = [((torch.randn(rows, a) * 0.5).to(dev, h16()), torch.randn(rows, b).to(dev, h16()), torch.zeros(a, b, device=dev),
              torch.zeros(a, device=dev), False) for a, b in shapes]
    o.wgrad_group(items)
    try:
        assert lib.ecamp_set_option(b"p8_wgrad_reserve_cus", 48) == 0     # raw C: same workgroup count (192), fewer pieces allowed (208 < 216): hip_ops._WG_TABLES still holds the old plan's image
        with pytest.raises(_lib.EcampHipError, match="not built for the current plan"):
            o.wgrad_group(items)
    finally:
        o.set_option("p8_wgrad_reserve_cus", 0)                            # (the wrapper clears the cache)
    o.wgrad_group(items)
    torch.cuda.synchronize()


@pytest.mark.usefixtures("both_halves")
def test_profiling_events_are_bounded(dev):
    """csrc/profile.hip: `main_pretrain.py --profile` brackets every GEMM / attention launch of a whole epoch with HIP events; the pool
    holds at most 4096 pairs however many launches are recorded (finished records are folded into running totals), and the totals
    still count every launch."""
    import ctypes
    from ecamp_amd import _lib
    lib = _lib.load()
    o = ops()
    x = torch.randn(256, 128, device=dev).to(h16())
    w = torch.randn(64, 128, device=dev).to(h16())
    lib.ecamp_prof_collect(-1, None, None, None)
    lib.ecamp_prof_enable(1)
    try:
        for _ in range(9000):
            o.linear_fwd(x, w)
    finally:
        lib.ecamp_prof_enable(0)
    assert 0 < int(lib.ecamp_prof_live_events()) <= 4096
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    lib.ecamp_prof_collect(0, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    assert n.value == 9000 and ms.value > 0 and fl.value == pytest.approx(9000 * 2.0 * 256 * 64 * 128)
    lib.ecamp_prof_collect(-1, None, None, None)
    assert int(lib.ecamp_prof_live_events()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_masks_are_statistically_sound(dev, p):
    """The library's mask convention (csrc/common.h: halfword (e & 7) of Philox4x32-7(counter e >> 3) >= round(65536 p)) on 2^22
    elements: keep rate = 1 - round(65536 p) / 65536 within 4 sigma -- overall and for each of the eight halfword positions of a call --
    no serial correlation at lags 1, 2, 8 and 4096 (|r| < 4 / sqrt(n)), and masks of neighbouring streams (offset, seed) are uncorrelated
    with each other.  (Seven rounds is the smallest Philox4x32 that passes BigCrush; this is a regression guard for the counter /
    halfword plumbing, not a substitute for that battery.)"""
    o = ops()
    n = 1 << 22
    q = 1.0 - round(65536 * p) / 65536.0
    m = o.dropout_mask((n,), dev, p, 1234, 77).float()
    sig = (q * (1 - q) / n) ** 0.5
    assert abs(m.mean().item() - q) < 4 * sig
    byp = m.view(-1, 8).mean(0)
    assert (byp - q).abs().max().item() < 4 * sig * 8 ** 0.5, byp.tolist()
    c = m - q
    var = (c * c).mean().item()
    for lag in (1, 2, 8, 4096):
        r = (c[:-lag] * c[lag:]).mean().item() / var
        assert abs(r) < 4 / n ** 0.5, (lag, r)
    for seed, off in ((1234, 78), (1235, 77), (1234, 77 + (1 << 32))):
        m2 = o.dropout_mask((n,), dev, p, seed, off).float() - q
        r = (c * m2).mean().item() / var
        assert abs(r) < 4 / n ** 0.5, (seed, off, r)
    u = o.uniform((n,), dev, 99, 5)
    assert abs(u.mean().item() - 0.5) < 4 * (1 / 12 / n) ** 0.5 and 0.0 <= u.min().item() and u.max().item() < 1.0
    cu = u - 0.5
    assert abs((cu[:-1] * cu[1:]).mean().item() * 12) < 4 / n ** 0.5


def test_randomised_shapes_through_gemm_attention_layernorm_and_cross_entropy(dev):
    """tools/fuzz_kernels.py, 60 random cases per kernel family and 16-bit build: shapes the tables above do not pin, with the kernel-selection
    options forced at random (128^2 / eight-wave / four-wave GEMMs), padded keys carrying 30 x the real keys' scale, dropout under the
    library's own masks.  (Round 6 ran it with 1 400 cases per family: profiles/r06_fuzz.txt.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_kernels.py"), "--cases", "60", "--seed", "11"], cwd=root, capture_output=True, text=True,
                       timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-12:])
    print(tail)
    assert r.returncode == 0 and " 0 failures" in tail, tail + "\n" + r.stderr[-1500:]
