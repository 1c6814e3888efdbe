#!/usr/bin/env python3
"""Reference point for DESIGN.md: the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the shapes of the HERO step,
next to hero_gemm / hero_wgrad_group on the same box.  Plain GEMMs only (no fused epilogue on the library side)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF


def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dt = torch.bfloat16
    M = 12000
    print("forward / dgrad shapes  y[M,N] = x[M,K] w[N,K]^T   (us, TF/s):  library | hero_gemm (no epilogue)")
    for N, K in [(3072, 768), (768, 3072), (768, 768), (2304, 768), (768, 2304)]:
        x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        y = torch.empty(M, N, device="cuda", dtype=dt)
        fl = 2.0 * M * N * K
        a = timeit(lambda: torch.matmul(x, w.t(), out=y))
        b = timeit(lambda: HF.k_linear(x, w))
        print("  N=%4d K=%4d   %6.1f us %6.0f | %6.1f us %6.0f" % (N, K, a, fl / a / 1e6, b, fl / b / 1e6))
    print("weight-gradient shapes  dW[N,K] += dY[M,N]^T x[M,K] (fp32 out):  library (bf16 in, fp32 addmm_) | hero k_wgrad")
    for N, K in [(3072, 768), (768, 3072), (768, 768), (2304, 768)]:
        x = torch.randn(M, K, device="cuda").to(dt); dy = torch.randn(M, N, device="cuda").to(dt)
        out = torch.zeros(N, K, device="cuda")
        outb = torch.zeros(N, K, device="cuda", dtype=dt)
        fl = 2.0 * M * N * K
        a = timeit(lambda: torch.matmul(dy.t(), x, out=outb))       # bf16 output (the library's plain form)
        b = timeit(lambda: HF.k_wgrad(dy, x, out=out, beta=1.0))
        print("  N=%4d K=%4d   %6.1f us %6.0f | %6.1f us %6.0f" % (N, K, a, fl / a / 1e6, b, fl / b / 1e6))


if __name__ == "__main__":
    main()
