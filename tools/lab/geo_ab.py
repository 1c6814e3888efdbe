#!/usr/bin/env python3
"""192 x 192 (hero_gemm_force_config 9) against 128 x 192 (10) wave-specialised tiles over row counts around the tile-round
boundaries of a ragged batch: where does the smaller tile win?  (fitting the cost ratio used by gemm_ws_run)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import functional as HF, _lib as L
dt = torch.bfloat16
def t(fn, reps=30):
    end = time.time() + 0.2
    while time.time() < end: fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps
print("%6s %5s %5s  tiles33 tiles23   t33 us   t23 us  ratio  default" % ("M", "N", "K"))
for N, K in ((768, 768), (768, 3072), (2304, 768), (3072, 768)):
    for M in (12000, 12480, 13000, 13600, 14450, 15400, 16400, 18000, 24000):
        x = (torch.randn(M, K, device="cuda")).to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        b = torch.zeros(N, device="cuda"); res = torch.zeros(M, N, device="cuda", dtype=dt)
        r = {}
        for cfg in (9, 10, -1):
            L.lib().hero_gemm_force_config(cfg)
            r[cfg] = t(lambda: HF.k_linear(x, w, b, residual=res))
        L.lib().hero_gemm_force_config(-1)
        t33 = -(-M // 192) * -(-N // 192); t23 = -(-M // 128) * -(-N // 192)
        print("%6d %5d %5d  %7d %7d  %7.1f  %7.1f  %5.2f  %7.1f   rounds %d / %d" % (M, N, K, t33, t23, r[9], r[10], r[10] / r[9], r[-1], -(-t33 // 256), -(-t23 // 256)))
