#!/usr/bin/env python3
"""hero_colsum micro-benchmark (graph-captured kernel time) at the shapes of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import functional as HF
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ln_bench import timeit
for rows, cols in ((12000, 2304), (12000, 768), (1920, 2304)):
    dy = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    out = torch.zeros(cols, device="cuda")
    ref = dy.float().sum(0)
    got = HF.k_colsum(dy)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(lambda: HF.k_colsum(dy, out=out, beta=1.0))
    print("colsum %5d x %4d: %5.1f us (%.2f TB/s) rel err %.2g" % (rows, cols, t, rows * cols * 2 / t / 1e6, err))
