#!/usr/bin/env python3
"""Randomised shapes through the GEMM and attention entry points (both 16-bit builds), against f32 torch on the same device: the kernel
SELECTION (128^2 / eight-wave / four-wave kernels, split counts, head-resident or streaming attention) depends on the shape, and the tests
pin a few dozen shapes of each.  Prints one line per failure and a summary; exit code 1 on any failure.
    python tools/fuzz_kernels.py [--cases 150] [--seed 0]"""
import argparse, math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ecamp_amd import _lib, hip_ops as o
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=150); ap.add_argument("--seed", type=int, default=0); args = ap.parse_args()
dev = torch.device("cuda:0")
rng = random.Random(args.seed)
TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3}
fails = []
def err(got, ref): 
    got, ref = got.detach().float(), ref.detach().float()
    if not torch.isfinite(got).all(): return float("inf")
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-3))     # (+1e-3: one key makes dQ and dK exactly zero in the reference)
def chk(tag, case, got, ref, tol):
    e = err(got, ref)
    if not e <= tol:
        fails.append((tag, case, e)); print("FAIL %-28s %s rel err %.3e (tol %.1e)" % (tag, case, e, tol), flush=True)
def randn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, device=dev) * scale).to(dt)
ngemm = natt = nln = nce = ngrp = nrefused = 0
for half, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    _lib.set_half(half)
    tol = TOL[dt]
    for c in range(args.cases):
        # ---- GEMM: forward epilogues, data gradient (+ gelu' / residual), weight gradient (+ bias gradient), random shapes (multiples of 8)
        M = rng.choice([8 * rng.randint(1, 40), 8 * rng.randint(40, 700), rng.choice([12800, 32768, 6304, 4096, 2048])])
        N = 8 * rng.randint(1, rng.choice([16, 96, 400]))
        K = 8 * rng.randint(1, rng.choice([16, 96, 400]))
        q8 = rng.choice([-1, -1, 0, 2]); q16 = rng.choice([-1, -1, 0, 2, 3])
        case = "%s M=%d N=%d K=%d q8=%d q16=%d" % (half, M, N, K, q8, q16)
        o.set_option("q8_mode", q8); o.set_option("q16_mode", q16)
        try:
            x, w, b = randn(M, K, dt=dt), randn(N, K, scale=K ** -0.5, dt=dt), randn(N)
            r = randn(M, N, dt=dt)
            xf, wf = x.float(), w.float()
            ref = xf @ wf.T
            chk("fwd", case, o.linear_fwd(x, w), ref, tol)
            chk("fwd+bias+res", case, o.linear_fwd(x, w, b, residual=r), ref + b + r.float(), tol)
            act = rng.choice([1, 2])
            y, pre = o.linear_fwd(x, w, b, act=act, save_pre=True)
            pref = (ref + b).to(dt).float()
            chk("fwd gelu(act=%d)" % act, case, y, F.gelu(pref), tol)
            dy = randn(M, N, dt=dt)
            dref = dy.float() @ wf
            chk("dgrad", case, o.linear_dgrad(dy, w), dref, tol)
            g = randn(M, K, dt=dt)
            gp = g.float().requires_grad_(True); F.gelu(gp).sum().backward()
            chk("dgrad*gelu'+res", case, o.linear_dgrad(dy, w, gmul=g, residual=x), dref * gp.grad + xf, tol)
            gw0 = randn(N, K); gw = gw0.clone(); gb = torch.zeros(N, device=dev)
            o.linear_wgrad(dy, x, gw, alpha=0.5, gb=gb, accumulate=True)
            chk("wgrad+acc", case, gw, gw0 + 0.5 * (dy.float().T @ xf), 1e-2)
            chk("wgrad bias", case, gb, 0.5 * dy.float().sum(0), 1e-2)
            ngemm += 1
        except Exception as e:   # an argument the library refuses is fine if it says so; anything else is a failure
            if "EcampHipError" not in type(e).__name__: fails.append(("exception", case, repr(e)[:200])); print("FAIL exception", case, repr(e)[:300], flush=True)
        finally:
            o.set_option("q8_mode", -1); o.set_option("q16_mode", -1)
        # ---- attention: packed / separate layouts, key mask, dropout (only checked for finiteness and against the no-dropout mean), cross offset
        hd = rng.choice([32, 64, 128]); H = rng.randint(1, 6); B = rng.randint(1, 5)
        Tq = rng.randint(1, 260); Tk = rng.randint(1, 260) if rng.random() < 0.5 else Tq
        masked = rng.random() < 0.5; p = rng.choice([0.0, 0.0, 0.1])
        case = "%s B=%d H=%d Tq=%d Tk=%d hd=%d mask=%d p=%.1f" % (half, B, H, Tq, Tk, hd, masked, p)
        try:
            D = H * hd
            q, k, v, do = randn(B, Tq, D, dt=dt), randn(B, Tk, D, dt=dt), randn(B, Tk, D, dt=dt), randn(B, Tq, D, dt=dt)
            km = None
            if masked:
                lens = torch.randint(1, Tk + 1, (B,), device=dev)
                km = (torch.arange(Tk, device=dev)[None, :] < lens[:, None]).int()
                k = torch.where(km[:, :, None].bool(), k, (k.float() * 30).to(dt))     # padded keys with large scores must not matter
            seed, off = 12345 + c, 7
            keep = o.dropout_mask((B, H, Tq, Tk), dev, p, seed, off).float() if p > 0 else torch.ones(B, H, Tq, Tk, device=dev)
            qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
            sp = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)
            sc = (sp(qr, Tq) @ sp(kr, Tk).transpose(-1, -2)) / math.sqrt(hd)
            if km is not None: sc = sc + (1.0 - km[:, None, None, :].float()) * torch.finfo(torch.float32).min
            out = ((sc.softmax(-1) * keep / (1 - p)) @ sp(vr, Tk)).permute(0, 2, 1, 3).reshape(B, Tq, D)
            out.backward(do.float())
            qs, ks = (Tq * D, D, hd), (Tk * D, D, hd)
            res = o.attn_fwd(q, k, v, B, H, Tq, Tk, hd, qs, ks, ks, 1 / math.sqrt(hd), km, p, seed, off, want_mask=True)
            og, lse, bits = res
            chk("attn fwd", case, og, out, tol)
            for b_ in ([bits, None] if bits is not None else [None]):
                dq, dk, dv = torch.empty_like(q), torch.zeros_like(k), torch.zeros_like(v)
                o.attn_bwd(q, k, v, og, do, lse, dq, dk, dv, B, H, Tq, Tk, hd, qs, ks, ks, qs, ks, ks, 1 / math.sqrt(hd), km, p, seed, off, drop_bits=b_)
                # (one valid key: the softmax is 1, dS = p (dP - delta) is exactly 0 in the reference and the rounding of O in delta here -- a zero
                # reference would make any relative error infinite: dQ / dK are then measured against their natural size |dO . V| |K| scale)
                nat = float((do.float().abs().max() * v.float().abs().max() * hd) * k.float().abs().max() / math.sqrt(hd))
                for nm, got_, ref_ in (("attn dq", dq, qr.grad), ("attn dk", dk, kr.grad)):
                    if float(ref_.abs().max()) < 1e-3 * nat:
                        e_ = float((got_.float() - ref_).abs().max()) / nat
                        if not e_ <= 2 * tol: fails.append((nm, case, e_)); print("FAIL %-28s %s abs err / natural size %.3e" % (nm, case, e_), flush=True)
                    else:
                        chk(nm, case, got_, ref_, 2 * tol)
                chk("attn dv", case, dv, vr.grad, 2 * tol)
            natt += 1
        except Exception as e:
            if "EcampHipError" not in type(e).__name__: fails.append(("exception", case, repr(e)[:200])); print("FAIL exception", case, repr(e)[:300], flush=True)
        # ---- grouped weight gradients: 1-4 layers sharing a row count, ragged shapes, overwrite / accumulate, with / without bias gradient
        rows_g = rng.choice([256 * rng.randint(1, 8), 8 * rng.randint(32, 1600), 12800, 6304])
        nl = rng.randint(1, 4)
        case = "%s wgrad_group rows=%d layers=%d" % (half, rows_g, nl)
        try:
            items, refs = [], []
            for i in range(nl):
                n_out, k_in = 8 * rng.randint(2, 200), 8 * rng.randint(2, 200)
                dy, x = randn(rows_g, n_out, scale=0.5, dt=dt), randn(rows_g, k_in, dt=dt)
                acc = rng.random() < 0.5
                base = randn(n_out, k_in) if acc else torch.zeros(n_out, k_in, device=dev)
                gbias = torch.ones(n_out, device=dev) if rng.random() < 0.6 else None
                items.append((dy, x, base.clone(), gbias, acc))
                refs.append((base + dy.float().T @ x.float(), None if gbias is None else 1.0 + dy.float().sum(0)))
                case += " [%dx%d%s%s]" % (n_out, k_in, "+" if acc else "", "b" if gbias is not None else "")
            if o.wgrad_group_supported(items):
                o.wgrad_group(items)
                for (dy, x, gw, gbias, acc), (rw, rb) in zip(items, refs):
                    chk("wgrad_group dW", case, gw, rw, 1e-2)
                    if gbias is not None: chk("wgrad_group db", case, gbias, rb, 1e-2)
                ngrp += 1
            else:
                nrefused += 1
        except Exception as e:
            if "EcampHipError" not in type(e).__name__: fails.append(("exception", case, repr(e)[:200])); print("FAIL exception", case, repr(e)[:300], flush=True)
            else: nrefused += 1
        # ---- LayerNorm (+ residual, + dropout under the library's own mask) forward / backward, and the weighted cross-entropy
        rows = rng.choice([rng.randint(1, 64), rng.randint(64, 3000), 12800]); cols = rng.choice([64, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048])
        use_res = rng.random() < 0.5; p = rng.choice([0.0, 0.1])
        case = "%s LN rows=%d cols=%d res=%d p=%.1f" % (half, rows, cols, use_res, p)
        try:
            x, res_ = randn(rows, cols, dt=dt), (randn(rows, cols, dt=dt) if use_res else None)
            g, b_, dy = 1 + 0.1 * randn(cols), 0.1 * randn(cols), randn(rows, cols, dt=dt)
            seed, off = 999 + c, 3
            keep = o.dropout_mask((rows, cols), dev, p, seed, off).float() if p > 0 else torch.ones(rows, cols, device=dev)
            xr = x.float().requires_grad_(True); rr = res_.float().requires_grad_(True) if use_res else None
            gr, br = g.clone().requires_grad_(True), b_.clone().requires_grad_(True)
            zr = xr * keep / (1 - p) + (rr if use_res else 0)
            zr.retain_grad()
            yr = F.layer_norm(zr, (cols,), gr, br, 1e-6); yr.backward(dy.float())
            y, z, mean, rstd = o.layernorm_fwd(x, g, b_, 1e-6, residual=res_, drop_p=p, seed=seed, offset=off)
            chk("ln fwd", case, y, yr, tol)
            gg, gb2 = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
            din = randn(rows, cols, dt=dt) if rng.random() < 0.5 else None     # a gradient arriving on the residual stream: ADDED to dz
            want = zr.grad + (din.float() if din is not None else 0)
            if p > 0:
                dz, dxd = o.layernorm_bwd(dy, z, mean, rstd, g, gg, gb2, dres=din, drop_p=p, seed=seed, offset=off, want_drop=True)
                chk("ln dz (+dres)", case, dz, want, 2 * tol)
                chk("ln dx (through dropout)", case, dxd, want * keep / (1 - p), 2 * tol)
            else:
                dz = o.layernorm_bwd(dy, z, mean, rstd, g, gg, gb2, dres=din)
                chk("ln dz (+dres)", case, dz, want, 2 * tol)
            chk("ln dgamma", case, gg, gr.grad, 2e-2); chk("ln dbeta", case, gb2, br.grad, 2e-2)
            nln += 1
        except Exception as e:
            if "EcampHipError" not in type(e).__name__: fails.append(("exception", case, repr(e)[:200])); print("FAIL exception", case, repr(e)[:300], flush=True)
            else: nrefused += 1
        Mc = rng.choice([rng.randint(1, 70), rng.randint(70, 5000)]); V = rng.choice([64, 256, 1000, 4096, 20000, 30000, 30522])
        case = "%s CE M=%d V=%d" % (half, Mc, V)
        try:
            lg = (randn(Mc, V) * 3).to(dt); lab = torch.randint(0, V, (Mc,), device=dev); lab[::5] = -100; wts = torch.rand(Mc, device=dev) * 2
            lr_ = lg.float().requires_grad_(True)
            loss = (F.cross_entropy(lr_, lab, reduction="none", ignore_index=-100) * wts).mean(); loss.backward()
            gain = 64.0 if dt == torch.float16 else 1.0
            s_ = torch.zeros(1, device=dev); ld = lg.clone(); o.ce_fwd_bwd_(ld, lab, wts, s_, gain=gain)
            chk("ce loss", case, s_ / Mc, loss.view(1), 1e-4); chk("ce dlogits", case, ld.float() / gain, lr_.grad, tol)
            nce += 1
        except Exception as e:
            if "EcampHipError" not in type(e).__name__: fails.append(("exception", case, repr(e)[:200])); print("FAIL exception", case, repr(e)[:300], flush=True)
            else: nrefused += 1
_lib.set_half("bf16")
print("fuzz: %d GEMM cases, %d attention cases, %d grouped weight-gradient cases, %d LayerNorm cases, %d cross-entropy cases (%d shapes refused by the library), %d failures"
      % (ngemm, natt, ngrp, nln, nce, nrefused, len(fails)), flush=True)
sys.exit(1 if fails else 0)
