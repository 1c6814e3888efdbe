"""Does a pinned H2D copy on a side stream run on an SDMA engine or as a blit kernel on the CUs?  Times a persistent
GEMM loop alone and with concurrent 33 MB H2D copies."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from hero_amd import functional as HF
dev = torch.device("cuda", 0)
x = torch.randn(12000, 768, device=dev).bfloat16(); w = (torch.randn(3072, 768, device=dev) * 0.05).bfloat16()
h = torch.randn(32, 60, 4352).pin_memory(); d = torch.empty_like(h, device=dev)
side = torch.cuda.Stream()
def gemms(n=100):
    for _ in range(n): HF.k_linear(x, w)
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
gemms(10)
print("100 GEMMs alone            %.2f ms" % T(gemms))
def copies(n=10):
    with torch.cuda.stream(side):
        for _ in range(n): d.copy_(h, non_blocking=True)
print("10 H2D copies alone        %.2f ms" % T(copies))
def both():
    copies(10); gemms(100)
print("100 GEMMs + 10 H2D copies  %.2f ms" % T(both))
