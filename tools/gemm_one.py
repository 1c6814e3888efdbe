#!/usr/bin/env python3
"""Run ONE GEMM shape/flavour repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L
kind, cfg, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
L.lib().hero_gemm_force_config(cfg)
dt = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
dy = torch.randn(M, N, device="cuda").to(dt); out = torch.zeros(N, K, device="cuda")
for _ in range(5):
    if kind == "fwd": HF.k_linear(x, w)
    elif kind == "dgrad": HF.k_dgrad(dy, w)
    else: HF.k_wgrad(dy, x, out=out, beta=1.0)
torch.cuda.synchronize()
