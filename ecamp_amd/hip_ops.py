"""Thin tensor-level wrappers over the C ABI (include/ecamp_hip.h).  torch is used ONLY for device memory
(`torch.empty`) and the current HIP stream handle; every FLOP / byte moved below happens in libecamp_hip.so.
No CPU fallback exists: CPU tensors raise."""
import ctypes

import torch

from . import _lib
from ._lib import BF16, F32, call

_I64x3 = ctypes.c_int64 * 3


def code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16 or dtype == torch.float16:
        want = "bf16" if dtype == torch.bfloat16 else "f16"
        if _lib.half() != want:   # the 16-bit format is a property of the loaded build, not of the call: the wrong one would reinterpret the bits
            raise TypeError("a %s tensor reached the %s build of libecamp_hip -- call ecamp_amd._lib.set_half(%r) first "
                            "(ECAMP(compute_dtype=...) does)" % (dtype, _lib.half(), want))
        return BF16
    raise TypeError("ecamp_amd supports float32, bfloat16 and float16 activations, got %s" % dtype)


def _chk(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EcampHipError("ecamp_amd ops need tensors on an MI355X (HIP) device; got a CPU tensor. "
                                     "There is no CPU fallback in the product path.")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# --------------------------------------------------------------------------------------------- gemm
def gemm(A, B, C, M, N, K, a_kc, lda, b_kc, ldb, ldc, bias=None, residual=None, ldr=0, pre_out=None, ldp=0, gmul=None,
         ldg=0, act=0, alpha=1.0, alpha_dev=None, out_f32=False, accumulate=False, split_k=1, rowsum=None):
    _chk(A, B, C)
    nws = int(_lib.load().ecamp_gemm_workspace_bytes(M, N, K, int(split_k))) // 4
    ws = torch.empty((nws,), device=A.device, dtype=torch.float32) if nws > 0 else None
    call("ecamp_gemm", ptr(A), ptr(B), ptr(C), M, N, K, int(a_kc), lda, int(b_kc), ldb, ldc, ptr(bias), ptr(residual), ldr,
         ptr(pre_out), ldp, ptr(gmul), ldg, int(act), float(alpha), ptr(alpha_dev), code(A.dtype), int(out_f32), int(accumulate), int(split_k),
         ptr(ws), ptr(rowsum), stream())
    return C


def linear_fwd(x, w, bias=None, act=0, residual=None, save_pre=False, out_dtype=None):
    """y = act(x @ w.T + bias) (+ residual);  x [M,K] (K-contiguous rows), w [N,K] in x.dtype."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1 and w.dtype == x.dtype
    out_f32 = out_dtype == torch.float32 and x.dtype != torch.float32
    y = torch.empty((M, N), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    pre = torch.empty((M, N), device=x.device, dtype=x.dtype) if save_pre else None
    gemm(x, w, y, M, N, K, True, x.stride(0), True, w.stride(0), N, bias=bias, residual=residual,
         ldr=residual.stride(0) if residual is not None else 0, pre_out=pre, ldp=N, act=act, out_f32=out_f32)
    return (y, pre) if save_pre else y


# --------------------------------------------------------------------------------------------- fp8 forward (configs[4])
def quantize_fp8(x):
    """Per-tensor e4m3 quantisation of a contiguous f32/bf16 tensor (numel % 4 == 0): -> (q uint8 same shape, scale f32[1]);
    x ~= q * scale, scale = max|x| / 448 taken from THIS tensor (current scaling: two passes, no history)."""
    _chk(x)
    assert x.is_contiguous()
    amax = torch.empty((1,), device=x.device, dtype=torch.float32)
    zero_(amax)
    call("ecamp_amax", ptr(x), ptr(amax), x.numel(), code(x.dtype), stream())
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    scale = torch.empty((1,), device=x.device, dtype=torch.float32)
    call("ecamp_quant_fp8", ptr(x), ptr(amax), ptr(q), ptr(scale), x.numel(), code(x.dtype), stream())
    return q, scale


def fp8_roll(amax_slots, scale, hist=None, hist_pos=0, margin=1.0):
    """Once per optimizer step: scale[i] = margin * max(site i's 16 amax slots, its history) / 448 where a producer fed the site; slots = 0.
    hist: optional f32 [sites, k] ring of the last k maxima (delayed scaling of activation sites); None + margin 1 = exact current scaling."""
    call("ecamp_fp8_roll", ptr(amax_slots), ptr(scale), scale.numel(), ptr(hist), hist.shape[1] if hist is not None else 0, int(hist_pos),
         float(margin), stream())


def fp8_weights(w16, w8, items, amax_slots, scales, pass_):
    """One pass of the whole-arena weight quantisation (ParamArena._quantize_weights): 0 = per-matrix maxima, 1 = quantise."""
    call("ecamp_fp8_weights", ptr(w16), ptr(w8), ptr(items), items.shape[0], ptr(amax_slots), ptr(scales), int(pass_), stream())


def quantize_fp8_site(x, scale, amax_slots, calibrated):
    """e4m3 copy of x for a GEMM-input site (ParamArena.f8_site).  Calibrated: ONE pass -- quantise with the site's scale (the
    previous step's maximum), record this step's maximum.  First use: the two-pass current scaling, which seeds scale and slots."""
    assert x.is_contiguous() and x.numel() % 4 == 0
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    if calibrated:
        call("ecamp_quant_fp8_delayed", ptr(x), ptr(scale), ptr(q), ptr(amax_slots), x.numel(), code(x.dtype), stream())
    else:
        call("ecamp_amax", ptr(x), ptr(amax_slots), x.numel(), code(x.dtype), stream())
        call("ecamp_quant_fp8", ptr(x), ptr(amax_slots), ptr(q), ptr(scale), x.numel(), code(x.dtype), stream())
    return q


def gemm_fp8(x8, x_scale, w8, w_scale, bias=None, act=0, residual=None, save_pre=False, q8_site=None):
    """y (bf16) = act(dequant(x8) @ dequant(w8).T + bias) (+ residual) on the e4m3 GEMM (ecamp_gemm_fp8): x8 [M,K], w8 [N,K] uint8.
    q8_site = (scale, amax slots) of the NEXT dense layer (GELU epilogue only): a third result, the e4m3 copy of y made in the epilogue."""
    M, K = x8.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and w8.dtype == torch.uint8 and x8.dtype == torch.uint8 and x8.is_contiguous() and w8.is_contiguous()
    y = torch.empty((M, N), device=x8.device, dtype=torch.bfloat16)
    pre = torch.empty((M, N), device=x8.device, dtype=torch.bfloat16) if save_pre else None
    y8 = torch.empty((M, N), device=x8.device, dtype=torch.uint8) if q8_site is not None else None
    call("ecamp_gemm_fp8", ptr(x8), ptr(w8), ptr(y), M, N, K, K, K, N, ptr(x_scale), ptr(w_scale), ptr(bias), ptr(residual),
         residual.stride(0) if residual is not None else 0, ptr(pre), N, int(act), ptr(y8), ptr(q8_site[0] if q8_site else None),
         ptr(q8_site[1] if q8_site else None), stream())
    if q8_site is not None:
        return y, pre, y8
    return (y, pre) if save_pre else y


def linear_fwd_fp8(x, w8, w_scale, bias=None, act=0, residual=None, save_pre=False):
    """y (bf16) = act(dequant(q(x)) @ dequant(w8).T + bias) (+ residual): x [M,K] bf16 is quantised here, w8 [N,K] uint8 + scale
    come from `quantize_fp8(weight)` (once per optimizer step).  The backward of the layer keeps using the bf16 x and w."""
    M, K = x.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and w8.dtype == torch.uint8 and x.dtype == torch.bfloat16 and x.is_contiguous() and w8.is_contiguous()
    x8, xs = quantize_fp8(x)
    y = torch.empty((M, N), device=x.device, dtype=torch.bfloat16)
    pre = torch.empty((M, N), device=x.device, dtype=torch.bfloat16) if save_pre else None
    call("ecamp_gemm_fp8", ptr(x8), ptr(w8), ptr(y), M, N, K, K, K, N, ptr(xs), ptr(w_scale), ptr(bias), ptr(residual),
         residual.stride(0) if residual is not None else 0, ptr(pre), N, int(act), ptr(None), ptr(None), ptr(None), stream())
    return (y, pre) if save_pre else y


def linear_dgrad(dy, w, gmul=None, alpha=1.0, alpha_dev=None, residual=None, gmul_is_grad=False):
    """dx = alpha * (dy @ w) [* gelu'(gmul)] [+ residual];  dy [M,N], w [N,K].  gmul_is_grad: `gmul` is what a forward with act=2 saved,
    gelu' itself (ecamp_gemm act = 2), and multiplies the result as it is."""
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), device=dy.device, dtype=dy.dtype)
    gemm(dy, w, dx, M, K, N, True, dy.stride(0), False, w.stride(0), K, gmul=gmul, ldg=gmul.stride(0) if gmul is not None else 0,
         residual=residual, ldr=residual.stride(0) if residual is not None else 0, alpha=alpha, alpha_dev=alpha_dev,
         act=2 if (gmul_is_grad and gmul is not None) else 0)
    return dx


_SPLIT_CACHE = {}
_WG_TABLES = {}   # (device, shapes, has_bias, rows, workgroups) -> device image of a weight-gradient group's item table (uploaded once, kept)


def set_option(name, value):
    """ecamp_set_option + invalidation of the split-count cache (the suggested split depends on the kernel selection)."""
    call("ecamp_set_option", name.encode(), int(value))
    _SPLIT_CACHE.clear()
    _WG_TABLES.clear()     # the grouped weight gradients' item tables depend on the CU reserve


def _split_k(n_out, k_in, m, dtype=torch.bfloat16):
    """Split count of the weight-gradient GEMM (M=n_out, N=k_in, K=m): asked of the library, which knows which kernel runs."""
    key = (n_out, k_in, m, dtype)
    s = _SPLIT_CACHE.get(key)
    if s is None:
        s = _SPLIT_CACHE[key] = max(1, int(_lib.load().ecamp_gemm_suggest_split(n_out, k_in, m, 0, 0, code(dtype))))
    return s


def linear_wgrad(dy, x, gw, alpha=1.0, alpha_dev=None, gb=None, accumulate=True):
    """gw[N,K] (f32) += (accumulate) or = alpha * dy[M,N]^T @ x[M,K];  gb[N] (f32, always accumulated) += alpha * sum_m dy[m,:] if given."""
    M, N = dy.shape
    K = x.shape[1]
    assert gw.dtype == torch.float32 and gw.is_contiguous() and gw.numel() == N * K
    gemm(dy, x, gw, N, K, M, False, dy.stride(0), False, x.stride(0), K, alpha=alpha, alpha_dev=alpha_dev, out_f32=True,
         accumulate=bool(accumulate), split_k=_split_k(N, K, M, dy.dtype), rowsum=gb)


WGRAD_GROUP = __import__("os").environ.get("ECAMP_WGRAD_GROUP", "1") != "0"   # one launch per transformer block for its weight gradients


def wgrad_group_supported(items):
    """items: [(dy, x, gw, gb or None, accumulate)] -- can they run as one grouped launch (same row count, bf16, aligned shapes)?"""
    if not WGRAD_GROUP or not 1 <= len(items) <= 4:
        return False
    rows = items[0][0].shape[0]
    for dy, x, gw, gb, _ in items:
        if dy.dtype not in (torch.bfloat16, torch.float16) or x.dtype != dy.dtype or dy.shape[0] != rows or x.shape[0] != rows:
            return False
        if not (dy.is_contiguous() and x.is_contiguous() and gw.is_contiguous() and gw.dtype == torch.float32):
            return False
    n = len(items)
    no = (ctypes.c_int64 * n)(*[it[0].shape[1] for it in items])
    ki = (ctypes.c_int64 * n)(*[it[1].shape[1] for it in items])
    return bool(_lib.load().ecamp_wgrad_group_supported(n, ctypes.cast(no, ctypes.c_void_p), ctypes.cast(ki, ctypes.c_void_p), rows))


def _wgrad_group_table(dev, n, no, ki, hb, rows, workgroups):
    lib = _lib.load()
    nwg = int(lib.ecamp_wgrad_group_workgroups(int(workgroups)))
    key = (str(dev), tuple(no), tuple(ki), tuple(hb), int(rows), nwg)
    t = _WG_TABLES.get(key)
    if t is None:
        cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
        cap = int(lib.ecamp_wgrad_group_table_bytes(n, cv(no), cv(ki), rows))
        host = torch.empty((cap,), dtype=torch.uint8)
        used = int(lib.ecamp_wgrad_group_table(n, cv(no), cv(ki), cv((ctypes.c_int32 * n)(*hb)), rows, int(workgroups), ctypes.c_void_p(host.data_ptr())))
        if used <= 0:
            raise _lib.EcampHipError("ecamp_wgrad_group_table failed: %s" % lib.ecamp_last_error().decode())
        # one-time upload on the current stream (a blocking copy from pageable memory: the first call for a shape set, never in a captured region)
        t = _WG_TABLES[key] = host[:used].to(dev)
    return t


def wgrad_group(items, alpha=1.0, alpha_dev=None, workgroups=0):
    """gw_p [N_p, K_p] (f32) (+)= alpha * dy_p^T x_p and gb_p += alpha * column sums of dy_p for every (dy, x, gw, gb, accumulate) of
    `items` in ONE persistent launch + one reduce (ecamp_wgrad_group): the weight gradients of one transformer block."""
    n = len(items)
    rows = items[0][0].shape[0]
    no = (ctypes.c_int64 * n)(*[it[0].shape[1] for it in items])
    ki = (ctypes.c_int64 * n)(*[it[1].shape[1] for it in items])
    vp = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])
    dyp, xp, gwp, gbp = vp([it[0] for it in items]), vp([it[1] for it in items]), vp([it[2] for it in items]), vp([it[3] for it in items])
    acc = (ctypes.c_int32 * n)(*[1 if it[4] else 0 for it in items])
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    dev = items[0][0].device
    table = _wgrad_group_table(dev, n, no, ki, [1 if it[3] is not None else 0 for it in items], rows, workgroups)
    nws = int(_lib.load().ecamp_wgrad_group_workspace_bytes(n, cv(no), cv(ki), rows)) // 4
    ws = torch.empty((nws,), device=dev, dtype=torch.float32)
    call("ecamp_wgrad_group", n, cv(dyp), cv(xp), cv(gwp), cv(gbp), cv(no), cv(ki), rows, float(alpha), ptr(alpha_dev), cv(acc), ptr(ws), ptr(table),
         table.numel(), int(workgroups), stream())


def colsum(x, out, alpha=1.0, period=0, lo=0, hi=0, alpha_dev=None):
    """out[N] (f32) += alpha * sum_m x[m, :]  (optionally only rows with lo <= m % period < hi)."""
    _chk(x, out)
    M, N = x.shape
    call("ecamp_colsum", ptr(x), x.stride(0), M, N, float(alpha), ptr(alpha_dev), period, lo, hi, ptr(out), code(x.dtype), stream())


# --------------------------------------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, eps, residual=None, drop_p=0.0, seed=0, offset=0, q8_site=None):
    """-> (y, z, mean, rstd);  z is x itself unless residual/dropout are fused (then the materialised LN input).
    q8_site = (scale f32[1], amax slots f32[512]) of the GEMM that consumes y (fp8 forward, calibrated site): a fifth result, the e4m3
    copy of y quantised inside this kernel (ecamp_layernorm_fwd_q8)."""
    _chk(x, gamma, beta)
    rows, cols = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    fused = residual is not None or drop_p > 0.0
    z = torch.empty_like(x) if fused else None
    if q8_site is not None:
        y8 = torch.empty((rows, cols), device=x.device, dtype=torch.uint8)
        call("ecamp_layernorm_fwd_q8", ptr(x), ptr(residual), ptr(z), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, cols,
             float(eps), float(drop_p), seed, offset, ptr(y8), ptr(q8_site[0]), ptr(q8_site[1]), code(x.dtype), stream())
        return y, (z if fused else x), mean, rstd, y8
    call("ecamp_layernorm_fwd", ptr(x), ptr(residual), ptr(z), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, cols,
         float(eps), float(drop_p), seed, offset, code(x.dtype), stream())
    return y, (z if fused else x), mean, rstd


def layernorm_bwd(dy, z, mean, rstd, gamma, ggamma, gbeta, dres=None, drop_p=0.0, seed=0, offset=0, want_drop=False):
    """-> dz (+dres) [, dz through the dropout mask];  ggamma/gbeta (f32) are accumulated."""
    rows, cols = dy.shape
    assert dy.is_contiguous() and z.is_contiguous()
    dz = torch.empty_like(dy)
    dxd = torch.empty_like(dy) if want_drop else None
    call("ecamp_layernorm_bwd", ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dz), ptr(dxd), ptr(ggamma),
         ptr(gbeta), rows, cols, float(drop_p), seed, offset, code(dy.dtype), stream())
    return (dz, dxd) if want_drop else dz


# --------------------------------------------------------------------------------------------- attention
def _st(t3):
    return _I64x3(*t3)


def attn_fwd(q, k, v, B, H, Tq, Tk, hd, qs, ks, vs, scale, key_mask=None, drop_p=0.0, seed=0, offset=0, want_mask=False):
    """q/k/v: tensors whose storage is addressed with element strides (batch, token, head); returns (o [B,Tq,H*hd], lse) and, with
    `want_mask`, the dropout keep-mask as bits (uint8 tensor, or None where the bit form does not apply) for `attn_bwd`."""
    _chk(q, k, v)
    o = torch.empty((B, Tq, H * hd), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    bits = None
    if want_mask and drop_p > 0.0:
        nb = int(_lib.load().ecamp_attn_mask_bytes(B, H, Tq, Tk, hd, code(q.dtype)))
        if nb > 0:
            bits = torch.empty((nb,), device=q.device, dtype=torch.uint8)
    call("ecamp_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(key_mask), B, H, Tq, Tk, hd, _st(qs), _st(ks), _st(vs),
         _st((Tq * H * hd, H * hd, hd)), float(scale), float(drop_p), seed, offset, code(q.dtype), ptr(bits), stream())
    return (o, lse, bits) if want_mask else (o, lse)


def attn_probs(q, k, B, H, Tq, Tk, hd, qs, ks, scale, key_mask=None):
    """softmax(scale * q k^T [+ finfo.min on masked keys]) -> f32 [B, H, Tq, Tk]  (Visualization/module/context_fusion.py:45-57)."""
    _chk(q, k)
    probs = torch.empty((B, H, Tq, Tk), device=q.device, dtype=torch.float32)
    call("ecamp_attn_probs", ptr(q), ptr(k), ptr(key_mask), ptr(probs), B, H, Tq, Tk, hd, _st(qs), _st(ks), float(scale), code(q.dtype), stream())
    return probs


def attn_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, Tq, Tk, hd, qs, ks, vs, dqs, dks, dvs, scale, key_mask=None, drop_p=0.0,
             seed=0, offset=0, drop_bits=None):
    delta = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    os_ = (Tq * H * hd, H * hd, hd)
    assert do.is_contiguous() and o.is_contiguous()
    call("ecamp_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), ptr(key_mask),
         B, H, Tq, Tk, hd, _st(qs), _st(ks), _st(vs), _st(os_), _st(os_), _st(dqs), _st(dks), _st(dvs), float(scale), float(drop_p),
         seed, offset, code(q.dtype), ptr(drop_bits), stream())


# --------------------------------------------------------------------------------------------- misc
def add(a, b):
    _chk(a, b)
    y = torch.empty_like(a)
    call("ecamp_add", ptr(a), ptr(b), ptr(y), a.numel(), code(a.dtype), stream())
    return y


def mul_gelu_grad(dy, pre):
    _chk(dy, pre)
    dx = torch.empty_like(dy)
    call("ecamp_gelu_bwd", ptr(dy), ptr(pre), ptr(dx), dy.numel(), code(dy.dtype), stream())
    return dx


def cast(src, dst):
    _chk(src, dst)
    call("ecamp_cast", ptr(src), ptr(dst), src.numel(), code(src.dtype), code(dst.dtype), stream())
    return dst


def scale_(x, alpha=1.0, alpha_dev=None):
    """x *= alpha * alpha_dev[0] in place (f32 arithmetic, any activation dtype)."""
    _chk(x, alpha_dev)
    assert x.is_contiguous() and x.numel() % 4 == 0
    call("ecamp_scale", ptr(x), ptr(x), x.numel(), float(alpha), ptr(alpha_dev), code(x.dtype), stream())
    return x


def zero_blocks_(g, flags):
    """Zero the 64-element blocks of the f32 arena `g` whose byte in `flags` (uint8, len = g.numel() / 64) is non-zero."""
    _chk(g, flags)
    call("ecamp_zero_blocks", ptr(g), ptr(flags), g.numel(), stream())


def zero_(t):
    _chk(t)
    call("ecamp_zero", ptr(t), t.numel() * t.element_size(), stream())
    return t


def zeros(shape, device, dtype=torch.float32):
    return zero_(torch.empty(shape, device=device, dtype=dtype))


def bcast_add(x, g):
    B, S, H = x.shape
    y = torch.empty_like(x)
    call("ecamp_bcast_add", ptr(x), ptr(g), ptr(y), B, S, H, code(x.dtype), stream())
    return y


def seq_sum(x, s0, s1, scale):
    B, S, H = x.shape
    out = torch.empty((B, H), device=x.device, dtype=x.dtype)
    call("ecamp_seq_sum", ptr(x), ptr(out), B, S, H, s0, s1, float(scale), code(x.dtype), stream())
    return out


def seq_bcast(g, y, s0, s1, scale, mode):
    B, S, H = y.shape
    call("ecamp_seq_bcast", ptr(g), ptr(y), B, S, H, s0, s1, float(scale), mode, code(y.dtype), stream())
    return y


def dropout_mask(shape, device, p, seed, offset):
    """Development ABI: the keep-mask (uint8, 1 = kept) of a tensor of `shape` under dropout(p) keyed by (seed, offset) -- what the
    LayerNorm / embedding / attention kernels regenerate from the same pair (element index = flat index of the contiguous tensor)."""
    out = torch.empty(shape, device=device, dtype=torch.uint8)
    call("ecamp_dropout_mask", ptr(out), out.numel(), float(p), seed, offset, stream())
    return out


def uniform(shape, device, seed, offset):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("ecamp_uniform", ptr(out), out.numel(), seed, offset, stream())
    return out


# --------------------------------------------------------------------------------------------- image side
IMG_MEAN, IMG_STD = 0.4721, 0.3037   # the reference's Normalize (pretrain_datasets.py:52): one value for the three identical channels


_IMG_LUT = {}


def image_lut(device):
    """lut[u] = ((float)u / 255 - mean) / std: ToTensor + Normalize of every byte value in their own f32 arithmetic (256 floats, built once)."""
    key = str(device)
    t = _IMG_LUT.get(key)
    if t is None:
        t = _IMG_LUT[key] = torch.arange(256, dtype=torch.float32).div(255.0).sub(IMG_MEAN).div(IMG_STD).to(device)
    return t


def is_u8_image(t):
    """The compact image schema: uint8 [B, H, W] (or [B, 1, H, W]) grayscale crops instead of normalised f32 [B, 3, H, W]."""
    return t.dtype == torch.uint8


def bicubic_resize(src, Hd, Wd):
    """f32 [B,C,Hs,Ws] -> f32 [B,C,Hd,Wd]; a uint8 [B,Hs,Ws] / [B,1,Hs,Ws] crop -> the normalised, resized f32 [B,3,Hd,Wd]."""
    _chk(src)
    if is_u8_image(src):
        src = src.reshape(src.shape[0], src.shape[-2], src.shape[-1])
        assert src.is_contiguous()
        B, Hs, Ws = src.shape
        dst = torch.empty((B, 3, Hd, Wd), device=src.device, dtype=torch.float32)
        call("ecamp_bicubic_resize_u8", ptr(src), ptr(dst), B, Hs, Ws, Hd, Wd, ptr(image_lut(src.device)), stream())
        return dst
    B, C, Hs, Ws = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous()
    dst = torch.empty((B, C, Hd, Wd), device=src.device, dtype=torch.float32)
    call("ecamp_bicubic_resize", ptr(src), ptr(dst), B * C, Hs, Ws, Hd, Wd, stream())
    return dst


def mask_indices(noise, len_keep):
    _chk(noise)
    B, L = noise.shape
    ids_restore = torch.empty((B, L), device=noise.device, dtype=torch.int32)
    ids_keep = torch.empty((B, len_keep), device=noise.device, dtype=torch.int32)
    mask = torch.empty((B, L), device=noise.device, dtype=torch.float32)
    call("ecamp_mask_indices", ptr(noise), B, L, len_keep, ptr(ids_restore), ptr(ids_keep), ptr(mask), stream())
    return ids_restore, ids_keep, mask


def im2col_gather(imgs, ids_keep, p, dtype):
    B, C, R, _ = imgs.shape
    Lk = ids_keep.shape[1]
    out = torch.empty((B * (Lk + 1), C * p * p), device=imgs.device, dtype=dtype)
    call("ecamp_im2col_gather", ptr(imgs), ptr(ids_keep), ptr(out), B, Lk, C, R, p, code(dtype), stream())
    return out


def assemble_tokens_(x, cls, pos, ids_keep, B, Lk, D):
    call("ecamp_assemble_tokens", ptr(x), ptr(cls), ptr(pos), ptr(ids_keep), B, Lk, D, code(x.dtype), stream())
    return x


def unshuffle_fwd(y, ids_restore, mask_token, dpos, B, L, Lk, D):
    xd = torch.empty((B, L + 1, D), device=y.device, dtype=y.dtype)
    call("ecamp_unshuffle_fwd", ptr(y), ptr(ids_restore), ptr(mask_token), ptr(dpos), ptr(xd), B, L, Lk, D, code(y.dtype), stream())
    return xd


def unshuffle_bwd(dxd, ids_restore, ids_keep, gmask_token, B, L, Lk, D):
    dy = torch.empty((B, Lk + 1, D), device=dxd.device, dtype=dxd.dtype)
    call("ecamp_unshuffle_bwd", ptr(dxd), ptr(ids_restore), ptr(ids_keep), ptr(dy), ptr(gmask_token), B, L, Lk, D, code(dxd.dtype),
         stream())
    return dy


def unpatchify_mim(pred, imgs, mask, loss_sum, B, R, p):
    pred_img = torch.empty((B, 3, R, R), device=pred.device, dtype=torch.float32)
    call("ecamp_unpatchify_mim", ptr(pred), ptr(imgs), ptr(mask), ptr(pred_img), ptr(loss_sum), B, R, p, code(pred.dtype), stream())
    return pred_img


def img_loss_bwd(pred_img, imgs, mask, dsr, gm_gs, B, R, p, dtype):
    L = (R // p) ** 2
    dpred = torch.empty((B * (L + 1), p * p * 3), device=pred_img.device, dtype=dtype)
    call("ecamp_img_loss_bwd", ptr(pred_img), ptr(imgs), ptr(mask), ptr(dsr), ptr(gm_gs), ptr(dpred), B, R, p, code(dtype), stream())
    return dpred


def sr_fwd(pred_img, big, column, row, w1, b1, w2, b2, loss_sum, super_patch, window, mode=0):
    """mode 0: f32 stencils (parity); 1: bf16 matrix cores."""
    B, _, R, _ = pred_img.shape
    call("ecamp_sr_fwd", ptr(pred_img), ptr(big), ptr(image_lut(big.device) if is_u8_image(big) else None), ptr(column), ptr(row), ptr(w1), ptr(b1), ptr(w2), ptr(b2),
         ptr(loss_sum), B, R, super_patch, window, int(mode), stream())


def sr_image(pred_img, w1, b1, w2, b2):
    """super_res(pred_img): f32 [B,3,2R,2R] (parity / visualisation; the training step never materialises it)."""
    B, _, R, _ = pred_img.shape
    out = torch.empty((B, 3, 2 * R, 2 * R), device=pred_img.device, dtype=torch.float32)
    call("ecamp_sr_image", ptr(pred_img), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out), B, R, stream())
    return out


def sr_bwd(pred_img, big, column, row, w1, b1, w2, b2, gw_ws, super_patch, window, mode=0):
    B, _, R, _ = pred_img.shape
    dsr = torch.empty((B, 3, R, R), device=pred_img.device, dtype=torch.float32)
    call("ecamp_sr_bwd", ptr(pred_img), ptr(big), ptr(image_lut(big.device) if is_u8_image(big) else None), ptr(column), ptr(row), ptr(w1), ptr(b1), ptr(w2), ptr(b2),
         ptr(dsr), ptr(gw_ws), B, R, super_patch, window, int(mode), stream())
    return dsr


def scaled_accum(ws, grad, scale_dev, idx):
    call("ecamp_scaled_accum", ptr(ws), ptr(grad), ptr(scale_dev), idx, ws.numel(), stream())


# --------------------------------------------------------------------------------------------- report side
def bert_embed_fwd(ids, type_ids, word, pos, typ, gamma, beta, eps, dtype, drop_p=0.0, seed=0, offset=0):
    _chk(ids, word)
    B, S = ids.shape
    H = word.shape[1]
    z = torch.empty((B * S, H), device=ids.device, dtype=dtype)
    e = torch.empty((B * S, H), device=ids.device, dtype=dtype)
    mean = torch.empty(B * S, device=ids.device, dtype=torch.float32)
    rstd = torch.empty(B * S, device=ids.device, dtype=torch.float32)
    call("ecamp_bert_embed_fwd", ptr(ids), ptr(type_ids), ptr(word), ptr(pos), ptr(typ), ptr(gamma), ptr(beta), ptr(z), ptr(e),
         ptr(mean), ptr(rstd), B, S, H, float(eps), float(drop_p), seed, offset, code(dtype), stream())
    return e, z, mean, rstd


def bert_embed_bwd(de, z, mean, rstd, gamma, ids, type_ids, gword, gpos, gtype, ggamma, gbeta, B, S, H, drop_p=0.0, seed=0,
                   offset=0, pad_id=0, hot=(2, 3)):
    call("ecamp_bert_embed_bwd", ptr(de), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(ids), ptr(type_ids), ptr(gword), ptr(gpos),
         ptr(gtype), ptr(ggamma), ptr(gbeta), B, S, H, pad_id, hot[0], hot[1], float(drop_p), seed, offset, code(de.dtype), stream())


def ce_fwd_bwd_(logits, labels, weights, loss_sum, gain=1.0):
    """In place: logits -> gain * d(mean weighted CE)/d logits (unit upstream gradient); loss_sum += sum_i w_i CE_i.
    `gain` (a power of two the caller divides out of the upstream gradient again): the gradient is written in the logits' own format before
    the loss scale is known, and w (p - onehot) / M is 1e-9 for M = 32768 rows -- zero in IEEE half."""
    _chk(logits, labels, weights)
    M, V = logits.shape
    call("ecamp_ce_fwd_bwd", ptr(logits), ptr(labels), ptr(weights), ptr(loss_sum), M, V, logits.stride(0), float(gain) / M,
         code(logits.dtype), stream())
    return logits


# --------------------------------------------------------------------------------------------- optimizer side
def sumsq(x, out):
    call("ecamp_sumsq", ptr(x), x.numel(), ptr(out), stream())


def adamw(p, g, m, v, p16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    call("ecamp_adamw", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p16), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
         float(wd), int(step), float(grad_scale), stream())


def adamw_grouped(p, g, m, v, p16, block_group, lrs, wds, beta1, beta2, eps, step, grad_scale=1.0, grad_sumsq=None, ctl=None):
    """grad_sumsq: optional zeroed f32[1]; receives sum((g * grad_scale)^2) over the updated elements (global grad-norm, fused).
    ctl: optional device f32[4] from loss_scale_update -- grad scale, skip flag and bias corrections are then read on the device."""
    n = len(lrs)
    arr = ctypes.c_float * n
    call("ecamp_adamw_grouped", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p16), ptr(block_group), p.numel(), n, arr(*lrs), arr(*wds),
         float(beta1), float(beta2), float(eps), int(step), float(grad_scale), ptr(grad_sumsq), ptr(ctl), stream())


def loss_scale_update(sumsq_, state, opt_step, ctl, norm_out, growth, backoff, interval, beta1, beta2):
    """GradScaler's unscale_ / step / update on the device (no host read): see include/ecamp_hip.h."""
    _chk(sumsq_, state, opt_step, ctl)
    call("ecamp_loss_scale_update", ptr(sumsq_), ptr(state), ptr(opt_step), ptr(ctl), ptr(norm_out), float(growth), float(backoff), int(interval),
         float(beta1), float(beta2), stream())


# --------------------------------------------------------------------------------------------- side-stream weight gradients
# Weight gradients are leaves of the backward dependency graph: nothing in backward waits for them.  Launching them on a
# second HIP stream lets their workgroups fill the tail-quantisation gaps of the dgrad / attention / LayerNorm kernels on the
# main stream (and vice versa).  The main stream re-joins at the end of backward (autograd engine callback).
OVERLAP_WGRAD = __import__("os").environ.get("ECAMP_OVERLAP_WGRAD", "1") != "0"
_side = {}


def side_stream(device):
    st = _side.get(device)
    if st is None:
        st = _side[device] = torch.cuda.Stream(device=device)
    return st


def _join_side():
    dev = torch.cuda.current_device()
    for key in (torch.device("cuda", dev), ("branch", dev)):
        st = _side.get(key)
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)
    _side["cb"] = False


# The image decoder + pixel losses and the report side both hang off the encoder's latent and nothing else: run on two streams,
# the workgroups of one fill the tail rounds of the other's persistent GEMMs (forward AND backward: autograd replays each
# stage's backward on the stream its forward ran on).  Measured on MI355X at B=256, same-box A/B of 30-step runs on three boxes
# (round 3, after the phase-schedule GEMM and the one-launch-per-block weight gradients): step 37.4-37.5 -> 36.6-36.65 ms, forward
# 14.3 -> 14.0, forward+backward 35.9 -> 35.1 (in round 2 the same switch bought 0.15 ms: the slower GEMMs left fewer gaps that
# the weight-gradient stream did not already fill).  On by default; ECAMP_OVERLAP_BRANCHES=0 serialises the two branches.
OVERLAP_BRANCHES = __import__("os").environ.get("ECAMP_OVERLAP_BRANCHES", "1") != "0"


def side_streams(device):
    """Every stream other than the main one that may hold unfinished gradient work of `device` (wgrad side stream, branch stream)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    out = []
    for key in (torch.device("cuda", idx), ("branch", idx)):
        st = _side.get(key)
        if st is not None:
            out.append(st)
    return out


def branch_stream(device):
    key = ("branch", torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=device)
    return st


# ---- tensors a second stream reads ------------------------------------------------------------------------------------------------
# `t.record_stream(side)` hands the question "when may this block be reused?" to the caching allocator: it answers with an event and keeps
# the block out of circulation until a LATER allocation call finds the event complete.  Which call that is depends on how far the host
# runs ahead of the GPU at that moment, so the allocator's steady state never quite arrives: round 6's bench record showed hipMalloc calls
# inside the timed steps in every run (7-107 even with the host's lead bounded), and one burst of them on a fresh box cost the host 87 ms
# and the GPU a step of 2x the median.  Instead the tensors are HELD here (a Python reference) until events recorded behind the step on
# every stream have completed, then dropped: the block goes back to the allocator the ordinary way, reusable at once by its own stream, and
# the step's allocation pattern is the same every step.  `seal()` is called where a step ends (FusedAdamW.pace) and by `hold` itself every
# 512 tensors or 8 GB (loops that never reach an optimizer); `reap()` drops what the GPU has finished.  ECAMP_HOLD_TENSORS=0: record_stream as before.
HOLD_TENSORS = __import__("os").environ.get("ECAMP_HOLD_TENSORS", "1") != "0"
_held_cur = []
_held_done = __import__("collections").deque()   # ([events], [tensors], bytes) in seal order
_held_bytes = 0                                   # bytes of the sealed sets still held
_cur_bytes = 0                                    # bytes of the open set
_cur_readers = []                                 # the streams named as readers of the open set (hold's first argument)


def held_bytes():
    """Bytes of the tensors held behind steps the GPU has not finished (FusedAdamW.pace lets the host run fewer steps ahead when that is large)."""
    return _held_bytes


def hold(st, *tensors):
    """`st` (a side stream) reads `tensors`, which live in another stream's pool: keep them alive until that has happened.  Contract: every
    read of them by `st` has ALREADY been queued when this is called (the weight-gradient launches) -- a tensor that a side stream reads again
    later (the branch stream's backward nodes) must use record_stream, whose event is taken when the tensor is freed."""
    if not HOLD_TENSORS:
        for t in tensors:
            if t is not None:
                t.record_stream(st)
        return
    global _cur_bytes
    if all(st is not r for r in _cur_readers):
        _cur_readers.append(st)
    for t in tensors:
        if t is not None:
            _held_cur.append(t)
            _cur_bytes += t.numel() * t.element_size()
    if len(_held_cur) >= 512 or _cur_bytes > (8 << 30):   # loops that never reach an optimizer step (evaluation, tests): bounded all the same
        seal()
    elif _held_done and (len(_held_cur) & 31) == 0:
        reap()


def seal():
    """Close the current set of held tensors behind an event on every stream that may read them (main, weight-gradient, branch); drop the
    sets whose events have all completed."""
    global _held_cur, _held_bytes, _cur_bytes
    if _held_cur:
        dev = _held_cur[0].device
        evs = []
        streams = [torch.cuda.current_stream(dev)]
        for st in list(_cur_readers) + side_streams(dev):   # every stream named as a reader, and the side streams whatever was named
            if all(st is not q and st != q for q in streams):
                streams.append(st)
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        del _cur_readers[:]
        _held_done.append((evs, _held_cur, _cur_bytes))
        _held_bytes += _cur_bytes
        _held_cur, _cur_bytes = [], 0
    reap()


def reap(block_first=False):
    """Drop held sets the GPU is done with (oldest first).  block_first: wait for the oldest one (tests)."""
    global _held_bytes
    while _held_done:
        evs, _, nb = _held_done[0]
        if block_first:
            for e in evs:
                e.synchronize()
            block_first = False
        if not all(e.query() for e in evs):
            break
        _held_done.popleft()
        _held_bytes -= nb


def wgrad_group_async(items, workgroups=0):
    """wgrad_group on the weight-gradient side stream (see linear_wgrad_async)."""
    if not OVERLAP_WGRAD:
        return wgrad_group(items, workgroups=workgroups)
    dev = items[0][0].device
    st = side_stream(dev)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        wgrad_group(items, workgroups=workgroups)
    for dy, x, _, _, _ in items:
        hold(st, dy, x)
    if not _side.get("cb"):
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side)
            _side["cb"] = True
        except RuntimeError:  # not inside a backward pass: join immediately
            _join_side()


def linear_wgrad_async(dy, x, gw, alpha=1.0, alpha_dev=None, gb=None, accumulate=True):
    if not OVERLAP_WGRAD:
        return linear_wgrad(dy, x, gw, alpha, alpha_dev, gb, accumulate)
    st = side_stream(dy.device)
    st.wait_stream(torch.cuda.current_stream())  # dy, x (and earlier accumulations into gw) are ready
    with torch.cuda.stream(st):
        linear_wgrad(dy, x, gw, alpha, alpha_dev, gb, accumulate)
    hold(st, dy, x, alpha_dev)  # the caching allocator must not recycle these while the side stream reads them
    if not _side.get("cb"):
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side)
            _side["cb"] = True
        except RuntimeError:  # not inside a backward pass: join immediately
            _join_side()


# --------------------------------------------------------------------------------------------- dataset image transform on the device
def resample_crops_workspace_bytes(B, out, kmax, tmp_rows):
    return int(_lib.load().ecamp_resample_crops_workspace_bytes(int(B), int(out), int(kmax), int(tmp_rows)))


def resample_crops_u8(flat, table, out_t, out, kmax, tmp_rows, max_h, ws, err):
    """B crops (flat uint8 + int64 table [B, 6], both on the device) -> out_t uint8 [B, out, out]: Pillow's crop().resize(BICUBIC) + flip,
    byte for byte (csrc/augment.hip; pretrain_datasets.py:47-52)."""
    _chk(flat, table, out_t, ws, err)
    assert flat.dtype == torch.uint8 and table.dtype == torch.int64 and table.is_contiguous() and out_t.dtype == torch.uint8 and out_t.is_contiguous()
    call("ecamp_resample_crops_u8", ptr(flat), ptr(table), ptr(out_t), table.shape[0], int(out), int(kmax), int(tmp_rows), int(max_h), ptr(ws), ws.numel(),
         ptr(err), stream())
    return out_t
