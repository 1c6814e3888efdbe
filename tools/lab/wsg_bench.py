#!/usr/bin/env python3
"""hero_wgrad_group on the four weight gradients of a BertLayer at 12000 rows (graph-captured kernel time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import _lib as L
from ln_bench import timeit
rows = 12000; dt = torch.bfloat16
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
dys = [torch.randn(rows, n, device="cuda").to(dt) for n, _ in shapes]
xs = [torch.randn(rows, k, device="cuda").to(dt) for _, k in shapes]
outs = [torch.zeros(n, k, device="cuda") for n, k in shapes]
pr = (L.WgradProblem * 4)(*[L.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), shapes[i][0], shapes[i][1],
                                            shapes[i][0], shapes[i][1], shapes[i][1], 4) for i in range(4)])
run = lambda: L.check(L.lib().hero_wgrad_group(pr, 4, rows, L.BF16, L.stream()))
run(); torch.cuda.synchronize()
err = max(((outs[i] - dys[i].float().t() @ xs[i].float()).abs().max() / (dys[i].float().t() @ xs[i].float()).abs().max()).item() for i in range(4))
t = timeit(run, n=20)
fl = sum(2.0 * rows * n * k for n, k in shapes)
print("hero_wgrad_group: %.1f us, %.0f TF/s, max rel err %.2g" % (t, fl / t / 1e6, err))
