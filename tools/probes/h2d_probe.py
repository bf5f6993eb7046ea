"""Host->HBM copy rate of one batch of images (616 MB pinned) alone and while the compute stream is saturated with GEMMs."""
import torch, time
dev = torch.device("cuda:0")
x = torch.empty(256, 3, 448, 448).pin_memory()
y = torch.empty_like(x, device=dev)
s = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(3): b = a @ a
torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
for n in range(3):
    e0, e1 = ev(), ev()
    with torch.cuda.stream(s):
        e0.record(s); y.copy_(x, non_blocking=True); e1.record(s)
    s.synchronize()
    print("H2D pinned alone: %.2f ms = %.1f GB/s" % (e0.elapsed_time(e1), x.numel() * 4 / e0.elapsed_time(e1) / 1e6))
g0, g1 = ev(), ev()
g0.record()
for _ in range(20): b = a @ a
g1.record(); torch.cuda.synchronize()
print("20 GEMMs alone: %.2f ms" % g0.elapsed_time(g1))
for n in range(2):
    e0, e1, g0, g1 = ev(), ev(), ev(), ev()
    g0.record()
    for _ in range(5): b = a @ a
    with torch.cuda.stream(s):
        e0.record(s); y.copy_(x, non_blocking=True); e1.record(s)
    for _ in range(15): b = a @ a
    g1.record(); torch.cuda.synchronize()
    print("overlapped: copy %.2f ms, 20 GEMMs %.2f ms" % (e0.elapsed_time(e1), g0.elapsed_time(g1)))
