#!/usr/bin/env python3
"""The 64-row wave-specialised geometries (hero_gemm_force_config 13: 64 x 128 tiles, 6-deep ring; 14: 64 x 192, 4-deep)
against the 4-wave launch (8 = never wave-specialised) and the heuristic, on the small-M shapes of the step, per epilogue;
relative L2 error against an fp32 matmul for every variant."""
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


print("%6s %5s %5s %-8s  4-wave   64x128   64x192  default  library   (us)   rel.err 4-wave / 64x128 / 64x192" % ("M", "N", "K", "epilogue"))
for M, N, K in [(1920, 768, 3072), (1920, 768, 2304), (1920, 768, 4352), (1920, 768, 768), (1920, 768, 1536), (2560, 768, 3072), (1000, 768, 3072), (1920, 1024, 4096), (3104, 768, 768), (3104, 768, 3072), (3104, 768, 2304)]:
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, K, device="cuda", generator=g).to(dt); w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(N, device="cuda", generator=g); res = torch.randn(M, N, device="cuda", generator=g).to(dt)
    ref0 = x.float() @ w.float().t()
    for name, fn, ref in (("none", lambda: HF.k_linear(x, w), ref0), ("bias+res", lambda: HF.k_linear(x, w, b, residual=res), ref0 + b + res.float())):
        ts, errs = [], []
        for cfg in (8, 13, 14, -1):
            L.lib().hero_gemm_force_config(cfg)
            y = fn()
            if cfg != -1: errs.append(((y.float() - ref).norm() / ref.norm()).item())
            ts.append(t(fn))
        L.lib().hero_gemm_force_config(-1)
        lib = t(lambda: F.linear(x, w))
        print("%6d %5d %5d %-8s %7.1f  %7.1f  %7.1f  %7.1f  %7.1f          %.2e / %.2e / %.2e" % (M, N, K, name, *ts, lib, *errs), flush=True)

print("\nfp32, M <= 32 (the loss head's Linear layers): skinny kernel (default) / 64 x 64 MFMA tiles (cfg 3), us")
for M, N, K in [(32, 768, 768), (32, 1920, 768), (32, 768, 3072)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    t_new = t(lambda: HF.k_linear(x, w, b))
    L.lib().hero_gemm_force_config(3)
    t_old = t(lambda: HF.k_linear(x, w, b))
    L.lib().hero_gemm_force_config(-1)
    print("%4d %5d %5d  %6.1f  %6.1f   library %6.1f" % (M, N, K, t_new, t_old, t(lambda: F.linear(x, w, b))), flush=True)

print("\nM = 480 (queries), bf16: 4-wave / 64x128 / 64x192 / default, us")
for M, N, K in [(480, 768, 768), (480, 2304, 768), (480, 768, 2304), (480, 3072, 768), (480, 768, 3072)]:
    x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt); b = torch.randn(N, device="cuda")
    ts = []
    for cfg in (8, 13, 14, -1):
        L.lib().hero_gemm_force_config(cfg)
        ts.append(t(lambda: HF.k_linear(x, w, b)))
    L.lib().hero_gemm_force_config(-1)
    print("%4d %5d %5d  %6.1f  %6.1f  %6.1f  %6.1f" % (M, N, K, *ts), flush=True)
