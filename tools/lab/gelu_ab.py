#!/usr/bin/env python3
"""A/B of the FFN1 forward (bias + GELU + saved pre-activation) and its backward (gelu' + column sums) GEMMs:
run once per library build (HERO_HIP_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import functional as HF, _lib as L
M, N, K = 12000, 3072, 768
dt = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
b = torch.zeros(N, device="cuda"); aux = torch.empty(M, N, device="cuda", dtype=dt)
dy = torch.randn(M, K, device="cuda").to(dt); wt = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
u = torch.randn(M, N, device="cuda").to(dt); cs = torch.zeros(N, device="cuda")
def t(fn, reps=50):
    end = time.time() + 0.3
    while time.time() < end: fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps
print(os.environ.get("HERO_HIP_LIB", "product"),
      "fwd bias+gelu+aux %.1f us" % t(lambda: HF.k_linear(x, w, b, act=L.ACT_GELU, aux=aux)),
      "| fwd bias %.1f us" % t(lambda: HF.k_linear(x, w, b)),
      "| bwd gelu'+colsum %.1f us" % t(lambda: HF.k_dgrad_t(dy, wt, act=L.ACT_GELU_BWD, aux=u, colsum=cs)),
      "| bwd plain %.1f us" % t(lambda: HF.k_dgrad_t(dy, wt)))
