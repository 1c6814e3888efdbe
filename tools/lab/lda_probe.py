#!/usr/bin/env python3
"""Does the leading dimension of a 3072-wide operand matter?  (A row stride of 6144 B = 24 x 256 B maps the rows of a
panel onto few L2 channels if the channel index is taken from low address bits.)  K,K GEMM M = 12000, N = 768, K = 3072
with lda / ldb = 3072 vs padded, and the reverse shape for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L
from ln_bench import timeit
dt = torch.bfloat16
M = 12000
for N, K in ((768, 3072), (3072, 768), (768, 768), (2304, 768)):
    for pad_a, pad_b, pad_c in ((0, 0, 0), (64, 0, 0), (64, 64, 0), (32, 32, 0), (128, 128, 0), (0, 0, 64), (64, 64, 64)):
        a = torch.randn(M, K + pad_a, device="cuda").to(dt)
        b = (torch.randn(N, K + pad_b, device="cuda") * 0.05).to(dt)
        c = torch.empty(M, N + pad_c, device="cuda", dtype=dt)
        bias = torch.randn(N, device="cuda")
        fn = lambda: HF.k_gemm(a, b, c, M, N, K, K + pad_a, K + pad_b, N + pad_c, L.LAYOUT_K, L.LAYOUT_K, L.BF16, bias=bias)
        fn(); torch.cuda.synchronize()
        err = (c[:, :N].float() - (a[:, :K].float() @ b[:, :K].float().t() + bias)).abs().max().item()
        t = timeit(fn, n=20)
        print("N %4d K %4d  lda +%3d ldb +%3d ldc +%3d : %6.1f us %6.0f TF/s (err %.2g)" % (N, K, pad_a, pad_b, pad_c, t, 2.0 * M * N * K / t / 1e6, err), flush=True)
