#!/usr/bin/env python3
"""Development probe: would the encoder's forward pass gain from running two half-batches on two streams (the 150-tile GEMMs of a
12800-row batch leave 41 % of the CUs idle; a second independent chain could fill them)?  Forward only, no autograd."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecamp_amd.module import model_ecamp
from ecamp_amd.functions import VitBlockFn
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = model_ecamp.ecamp(compute_dtype=torch.bfloat16).to(dev); m.prepare(); m.eval()
B, T, D = 256, 50, 768
x = torch.randn(B * T, D, device=dev).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def full():
    y = x
    for blk in m.blocks:
        y = VitBlockFn.apply(y, blk, m, B, T, m.num_heads)
    return y

def halves(parts=2):
    cur = torch.cuda.current_stream()
    outs = []
    streams = [s1, s2]
    hb = B // parts
    for i in range(parts):
        st = streams[i % 2]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            y = x[i * hb * T:(i + 1) * hb * T]
            for blk in m.blocks:
                y = VitBlockFn.apply(y, blk, m, hb, T, m.num_heads)
            outs.append(y)
    for st in streams:
        cur.wait_stream(st)
    return outs

def interleaved(parts=2):
    """the same two chains, enqueued block by block (the host alternates between the streams, as one Function per block would)"""
    cur = torch.cuda.current_stream()
    streams = [s1, s2]
    hb = B // parts
    ys = [x[i * hb * T:(i + 1) * hb * T] for i in range(parts)]
    for st in streams: st.wait_stream(cur)
    for blk in m.blocks:
        for i in range(parts):
            with torch.cuda.stream(streams[i % 2]):
                ys[i] = VitBlockFn.apply(ys[i], blk, m, hb, T, m.num_heads)
    for st in streams: cur.wait_stream(st)
    return ys

def timeit(fn, n=10):
    with torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    a = full(); b = torch.cat(halves()); torch.cuda.synchronize()
    print("max |full - halves| =", float((a.float() - b.float()).abs().max()))
for r in range(2):
    print("encoder forward, 12 blocks: full batch %.3f ms | two halves on two streams %.3f ms | interleaved enqueue %.3f ms | four quarters %.3f ms" %
          (timeit(full), timeit(halves), timeit(interleaved), timeit(lambda: halves(4))))
