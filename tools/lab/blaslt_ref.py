#!/usr/bin/env python3
"""What the vendor library (hipBLASLt / rocBLAS behind torch.nn.functional.linear) does on the K,K shapes of the step, next to
hero_gemm (no epilogue / bias): is a library GEMM + separate elementwise kernel ever the better deal?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from hero_amd import functional as HF
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


print("%6s %5s %5s   hero none  hero bias   lib none   lib bias   (us; TF/s of the faster plain one)" % ("M", "N", "K"))
for M, N, K in [(12000, 2304, 768), (12000, 3072, 768), (12000, 768, 768), (12000, 768, 3072), (12000, 768, 2304), (1920, 768, 3072), (1920, 768, 768), (1920, 3072, 768)]:
    x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda"); bb = b.to(dt)
    r = [t(lambda: HF.k_linear(x, w)), t(lambda: HF.k_linear(x, w, b)), t(lambda: F.linear(x, w)), t(lambda: F.linear(x, w, bb))]
    print("%6d %5d %5d  %9.1f  %9.1f  %9.1f  %9.1f   %6.0f" % (M, N, K, *r, 2.0 * M * N * K / min(r[0], r[2]) / 1e6), flush=True)
