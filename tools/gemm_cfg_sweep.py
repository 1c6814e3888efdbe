#!/usr/bin/env python3
"""Forward GEMM (bias epilogue) time per forced tile geometry on the step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L
from gemm_bench import timeit
for (M, N, K) in [(12000, 3072, 768), (12000, 2304, 768), (12000, 768, 3072), (12000, 768, 2304), (12000, 768, 768),
                  (1920, 3072, 768), (1920, 2304, 768), (1920, 768, 3072), (1920, 768, 768), (1920, 768, 4352), (480, 768, 768)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = []
    for cfg in (0, 1, 3):
        L.lib().hero_gemm_force_config(cfg)
        t = timeit(lambda: HF.k_linear(x, w, b, residual=res), n=30)
        out.append("%d: %6.1f us %5.0f TF/s" % (cfg, t, 2.0 * M * N * K / t / 1e6))
    L.lib().hero_gemm_force_config(-1)
    print("M=%5d N=%4d K=%4d  " % (M, N, K) + "  |  ".join(out), flush=True)
