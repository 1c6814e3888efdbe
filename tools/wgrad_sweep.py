#!/usr/bin/env python3
"""Sweep the reduction split of the wgrad GEMM (dW = dY^T X, fp32 atomics merge) per step shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L
from gemm_bench import timeit

for (M, N, K) in [(12000, 3072, 768), (12000, 768, 3072), (12000, 2304, 768), (12000, 768, 768),
                  (1920, 768, 4352), (1920, 3072, 768), (1920, 2304, 768), (1920, 768, 768), (480, 768, 768)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    out = torch.zeros(N, K, device="cuda")
    res = []
    for s in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14):
        t = timeit(lambda: HF.k_gemm(dy, x, out, N, K, M, N, K, K, L.LAYOUT_O, L.LAYOUT_O, L.BF16, out_f32=True,
                                     beta=1.0, split_k=s), n=20)
        res.append("%d:%.0f" % (s, t))
    print("M=%5d N=%4d K=%4d tiles=%3d  %s   [heuristic %d]" % (M, N, K, ((N + 127) // 128) * ((K + 127) // 128), " ".join(res),
                                                     HF._split_for(N, K, M, 64)), flush=True)
