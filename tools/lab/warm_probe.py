#!/usr/bin/env python3
"""Is the first second of GEMM work slower (clock ramp)?  The same K,K GEMM timed ten times in a row."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L
from ln_bench import timeit
dt = torch.bfloat16
M, N, K = 12000, 768, 3072
a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt); bias = torch.randn(N, device="cuda")
c = torch.empty(M, N, device="cuda", dtype=dt)
fn = lambda: HF.k_gemm(a, b, c, M, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.BF16, bias=bias)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    t = timeit(fn, n=20)
    print("t = %5.2f s  %6.1f us" % (time.perf_counter() - t0, t), flush=True)
c2 = torch.empty(M, N, device="cuda", dtype=dt)
fn2 = lambda: HF.k_gemm(a, b, c2, M, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.BF16, bias=bias)
print("fresh C:", timeit(fn2, n=20), timeit(fn, n=20))
