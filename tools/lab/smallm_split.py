#!/usr/bin/env python3
"""Small-M, long-K GEMMs (the Temporal Transformer's FFN2 and its gradients: M = 1920, N = 768, K = 3072 / 2304 / 4352):
the existing split-K of the 4-wave kernels (fp32 atomics, a stand-in for a slab store) under forced tile geometries, next to
the production launch and the vendor library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from hero_amd import functional as HF, _lib as L
dt = torch.bfloat16


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        with torch.cuda.graph(g, stream=gs):
            for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(gs)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps / 3


for M, N, K in [(1920, 768, 3072), (1920, 768, 2304), (1920, 768, 4352), (1920, 768, 768), (480, 768, 3072)]:
    x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(dt)
    out = torch.zeros(M, N, device="cuda")
    print("%d x %d x %d: production bias+res %.1f us, plain %.1f us, library %.1f us" % (
        M, N, K, t(lambda: HF.k_linear(x, w, b, residual=res)), t(lambda: HF.k_linear(x, w)), t(lambda: F.linear(x, w))), flush=True)
    for cfg, name in ((3, "64x64"), (0, "128x128"), (1, "192x128")):
        L.lib().hero_gemm_force_config(cfg)
        row = []
        for S in (1, 2, 3, 4, 6, 8):
            fn = lambda: HF.k_gemm(x, w, out, M, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.BF16, out_f32=True, beta=1.0, split_k=S)
            row.append("S=%d %5.1f" % (S, t(fn)))
        print("   %-8s fp32 out, split-K atomics:  %s" % (name, "  ".join(row)), flush=True)
    L.lib().hero_gemm_force_config(-1)
