#!/usr/bin/env python3
"""HBM-roofline table of the memory-bound kernels of the bench step (BASELINE.json configs[1]: B=256, S=128, bf16):

    python tools/hbm_table.py gpurun_out/kernel_stats_<tag>_serialized.txt > profiles/r05_hbm_kernels.txt

For every non-GEMM kernel that is >= 0.1 % of the serialized step: launches per step, ALGORITHMIC bytes per step (the minimal traffic of
SURVEY.md 8(d): every input tensor read once, every output written once, at the storage width the kernel is given -- masks and
intermediates that never have to exist are not counted), its serialized microseconds per step (tools/step_kernels.sh: rocprofv3
--kernel-trace of bench.py with every kernel alone on the GPU), the achieved TB/s and the fraction of the 8 TB/s HBM3E peak
(MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming copy reaches, shown as `of copy`).  Kernels whose floor is not HBM (the attention
kernels' matrix + softmax work, the SR head's stencils) are listed all the same: the column says how far their time is from what moving
their operands would cost.
"""
import re
import sys

B, S, T, TD, D, DD, HB = 256, 128, 50, 197, 768, 512, 768     # pairs, report tokens, encoder / decoder tokens, widths
V, R = 30000, 224
MB = 1e6
bf, f4 = 2, 4
ROWS_E, ROWS_D, ROWS_T = B * T, B * TD, B * S               # 12800, 50432, 32768
NPARAM = 183.17e6

# kernel-name prefix -> (launches per step, algorithmic bytes per step, what is counted)
K = {}


def add(name, calls, nbytes, what):
    K[name] = (calls, nbytes, what)


ln_e = ROWS_E * D * bf
ln_t = ROWS_T * HB * bf
ln_d = ROWS_D * DD * bf
add("ln_fwd_kernel<unsigned short, 2, 8>", 41, 25 * 2 * ln_e + 15 * 4 * ln_t + 1 * 2 * ln_t,
    "25 encoder LN (x in, y out) + 15 BERT post-LN (dense out + residual in; z + y out) + MLM transform LN")
add("ln_fwd_kernel<unsigned short, 1, 8>", 9, 9 * 2 * ln_d, "9 decoder LN on 50432 x 512 (x in, y out)")
add("ln_bwd_kernel<unsigned short, 3, 4, 16, true>", 41, 25 * 4 * ln_e + 15 * 4 * ln_t + 1 * 3 * ln_t,
    "25 encoder (dy, z, residual grad in; dz out) + 15 BERT (dy, z in; dz, dropped dz out) + transform LN (dy, z in; dz out)")
add("ln_bwd_kernel<unsigned short, 1, 8, 16, true>", 9, 9 * 4 * ln_d, "9 decoder LN backward (dy, z, residual grad in; dz out)")
qkv_e = B * 12 * T * 64 * bf
qkv_d = B * 16 * TD * 32 * bf
qkv_t = B * 6 * S * 128 * bf
kv_x = B * 6 * 49 * 128 * bf
add("attn_head_fwd_kernel<64, 0>", 12, 12 * 4 * qkv_e, "encoder T=50, hd 64: q, k, v in, o out")
add("attn_head_fwd_kernel<32, 0>", 4, 4 * 4 * qkv_d, "decoder T=197, hd 32: q, k, v in, o out")
add("attn_head_fwd_kernel<128, 1>", 8, 7 * 4 * qkv_t + (2 * qkv_t + 2 * kv_x) + 8 * B * 6 * S * 32,
    "7 report self-attentions S=128, hd 128 + the cross-attention onto 49 image tokens; + the dropout mask bits (32 B per row and head)")
add("attn_head_bwd_kernel<64, 0>", 12, 12 * 8 * qkv_e, "q, k, v, o, dO in; dq, dk, dv out")
add("attn_head_bwd_kernel<32, 0>", 4, 4 * 8 * qkv_d, "q, k, v, o, dO in; dq, dk, dv out")
add("attn_head_bwd_kernel<128, 1>", 8, 7 * 8 * qkv_t + (4 * qkv_t + 4 * kv_x) + 8 * B * 6 * S * 32,
    "7 self (8 tensors of 50 MB) + cross (q, o, dO, dq + k, v, dk, dv of 49 keys) + mask bits")
add("adamw_grouped_kernel", 1, 28 * NPARAM, "p, g, m, v read (16 B) + p, m, v written (12 B) per parameter; the bf16 shadow (+2 B) not counted")
img224, img448 = B * 3 * R * R * f4, B * 3 * 4 * R * R * f4
add("sr_pair_fwd_kernel", 1, img224 + img448, "pred_img (f32 224^2) + big (f32 448^2) in; loss scalar out")
add("sr_pair_bwd_kernel", 1, img224 + img448 + img224, "pred_img + big in; d loss / d pred_img (f32 224^2) out")
add("ce_fwd_bwd_row_kernel<4, 1024>", 1, 2 * ROWS_T * V * bf, "bf16 logits [32768, 30000] read once, their gradient written over them")
add("bicubic_half_kernel", 1, img448 + img224, "f32 448^2 image in, 224^2 out")
add("wgrad_group_reduce_kernel", 22, None, "f32 slabs of the grouped weight gradients in, gradient arena out (slab count per tile varies)")
add("splitk_reduce_kernel", 13, None, "split-K slabs in, gradient arena out")
add("bert_embed_bwd_kernel<unsigned short, 3>", 1, 2 * ln_t + ROWS_T * HB * f4 * 0, "de, z in (bf16); embedding-row gradients by atomics (not counted)")
add("bert_embed_fwd_kernel<unsigned short, 3>", 1, 2 * ln_t, "z, e out (bf16); ids + table rows in (cached)")
add("img_loss_bwd_kernel<unsigned short>", 1, 2 * img224 + img224 + B * 196 * 768 * bf, "pred_img, imgs, dsr in (f32 224^2); d pred out (bf16 [B,196,768])")
add("unpatchify_mim_kernel<unsigned short>", 1, B * 196 * 768 * bf + 2 * img224, "pred (bf16) + imgs in; pred_img (f32) out; masked-MSE scalar")
add("zero_blocks_kernel", 1, 0.13 * 4 * NPARAM, "the atomically accumulated 13 % of the f32 gradient arena zeroed")

PEAK, COPY = 8.0, 6.3


def main():
    rows, nsteps = {}, None
    for line in open(sys.argv[1]):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    if "adamw_grouped_kernel" in rows:
        nsteps = rows["adamw_grouped_kernel"][0]
    total = sum(v[1] for v in rows.values()) / nsteps
    print("# %s: %d optimizer steps, %.2f ms of kernels per step (serialized: one kernel at a time)" % (sys.argv[1], nsteps, total / 1e3))
    print("# configs[1]: B=256, S=128, bf16.  HBM3E peak %.1f TB/s; a streaming copy reaches ~%.1f TB/s (MI355X_MICROARCH.md)." % (PEAK, COPY))
    print("%-48s %6s %10s %9s %7s %7s %8s  %s" % ("kernel", "calls", "alg MB", "us/step", "TB/s", "of peak", "of copy", "algorithmic bytes = "))
    tot_us = 0.0
    for name, (calls, nbytes, what) in K.items():
        hit = [k for k in rows if k.startswith(name)]
        if not hit:
            continue
        n, us = rows[hit[0]]
        us_step = us / nsteps
        tot_us += us_step
        if nbytes is None:
            print("%-48s %6.1f %10s %9.1f %7s %7s %8s  %s" % (name[:48], n / nsteps, "-", us_step, "-", "-", "-", what))
            continue
        tbs = nbytes / (us_step * 1e-6) / 1e12
        print("%-48s %6.1f %10.0f %9.1f %7.2f %6.0f%% %7.0f%%  %s" % (name[:48], n / nsteps, nbytes / MB, us_step, tbs, 100 * tbs / PEAK, 100 * tbs / COPY, what))
    gemm = sum(v[1] for k, v in rows.items() if k.startswith("gemm_")) / nsteps
    print("# listed non-GEMM kernels: %.2f ms per step; GEMM family %.2f ms; everything else %.2f ms" % (tot_us / 1e3, gemm / 1e3, (total - tot_us - gemm) / 1e3))


if __name__ == "__main__":
    main()
