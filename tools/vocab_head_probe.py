#!/usr/bin/env python3
"""The vocabulary head priced against its fused alternative:

    python tools/vocab_head_probe.py > gpurun_out/r05_vocab_head.txt

Shape of configs[1]: logits [32768, 30000] bf16 (1.97 GB), hidden 768.  Measured, every pass on buffers that were just written by the pass
before (as in the step; nothing of 1.97 GB stays in the 256 MB Infinity Cache):
  A  the product's chain: forward GEMM (bias epilogue) -> ecamp_ce_fwd_bwd (one read of the logits, gradient written over them)
  B  the floor of ANY pass that reads the logits once and writes the gradient once (what a CE pass fed with ready row statistics would be):
     an in-place elementwise multiply over the same buffer
  C  what an epilogue that evaluates exp() per logit costs the GEMM: the same GEMM with the GELU + saved-derivative epilogue (an upper bound:
     it also stores a second [32768, 30000] tensor, which a statistics epilogue would not)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd import hip_ops as o

dev = torch.device("cuda:0")
M, V, H = 32768, 30000, 768
bf = torch.bfloat16


def timeit(fn, n=6):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(M, H, device=dev, generator=g) * 2 - 1).to(bf)
    w = ((torch.rand(V, H, device=dev, generator=g) * 2 - 1) * H ** -0.5).to(bf)
    bias = torch.zeros(V, device=dev)
    logits = torch.empty(M, V, device=dev, dtype=bf)
    pre = torch.empty(M, V, device=dev, dtype=bf)
    labels = torch.randint(0, V, (M,), device=dev, generator=g)
    wts = torch.ones(M, device=dev)
    loss = torch.zeros(1, device=dev)
    gb = 2.0 * M * V * 2 / 1e9

    t_gemm = timeit(lambda: o.gemm(x, w, logits, M, V, H, True, H, True, H, V, bias=bias))
    t_gemm_e1 = timeit(lambda: o.gemm(x, w, logits, M, V, H, True, H, True, H, V, bias=bias, pre_out=pre, ldp=V, act=2))

    def chain():
        o.gemm(x, w, logits, M, V, H, True, H, True, H, V, bias=bias)
        o.ce_fwd_bwd_(logits, labels, wts, loss)
    t_chain = timeit(chain)

    def chain_mul():
        o.gemm(x, w, logits, M, V, H, True, H, True, H, V, bias=bias)
        logits.mul_(0.5)
    t_chain_mul = timeit(chain_mul)
    t_ce, t_mul = t_chain - t_gemm, t_chain_mul - t_gemm
    print("# vocabulary head, logits [%d, %d] bf16 = %.2f GB; us (median of 6)" % (M, V, M * V * 2 / 1e9))
    print("A  forward GEMM, bias epilogue                       %8.1f   (%.0f TFLOP/s)" % (t_gemm, 2.0 * M * V * H / t_gemm / 1e6))
    print("A  ecamp_ce_fwd_bwd behind it (chain - GEMM)         %8.1f   (%.2f TB/s of read + write)" % (t_ce, gb / t_ce * 1e3))
    print("B  in-place multiply behind it (chain - GEMM)        %8.1f   (%.2f TB/s): the floor of a CE pass that is handed its row statistics" % (t_mul, gb / t_mul * 1e3))
    print("C  forward GEMM, GELU + saved-derivative epilogue    %8.1f   (+%.1f over the bias epilogue: exp-class arithmetic per logit AND a second 1.97 GB store)" %
          (t_gemm_e1, t_gemm_e1 - t_gemm))
    print("#")
    print("# a fused form can return at most A(ce) - B = %.1f us and pays the epilogue arithmetic for it" % (t_ce - t_mul))


if __name__ == "__main__":
    main()
