#!/usr/bin/env python3
"""hero_wgrad_batch (whole tiles, all layers in one launch) vs hero_wgrad_group (stream-K, one launch per layer) on the
weight gradients of the cross-modal stack (6 layers, 12000 rows) and the Temporal Transformer (3 layers, 1920 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hero_amd import _lib as L
from ln_bench import timeit
dt = torch.bfloat16
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
for rows, layers in ((12000, 6), (1920, 3)):
    dys = [torch.randn(rows, n, device="cuda").to(dt) for _ in range(layers) for n, _ in shapes]
    xs = [torch.randn(rows, k, device="cuda").to(dt) for _ in range(layers) for _, k in shapes]
    outs = [torch.zeros(n, k, device="cuda") for _ in range(layers) for n, k in shapes]
    n = len(dys)
    pr = (L.WgradProblem * n)(*[L.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), outs[i].shape[0], outs[i].shape[1],
                                               outs[i].shape[0], outs[i].shape[1], outs[i].shape[1], 4) for i in range(n)])
    buf = np.zeros(8 + 8 * 256 * 16, dtype=np.int32)
    words = L.lib().hero_wgrad_batch_plan(pr, n, rows, buf.ctypes.data, buf.size)
    plan = torch.from_numpy(buf[:words].copy()).cuda()
    batch = lambda: L.check(L.lib().hero_wgrad_batch(pr, n, rows, L.BF16, plan.data_ptr(), words, L.stream()))
    def group():
        for g0 in range(0, n, 4):
            sub = (L.WgradProblem * 4)(*[pr[i] for i in range(g0, g0 + 4)])
            L.check(L.lib().hero_wgrad_group(sub, 4, rows, L.BF16, L.stream()))
    fl = sum(2.0 * rows * o.shape[0] * o.shape[1] for o in outs)
    for name, fn in (("batch (whole tiles)", batch), ("group (stream-K per layer)", group)):
        fn(); torch.cuda.synchronize()
        t = timeit(fn, n=10)
        print("rows %5d layers %d  %-28s %8.1f us  %6.0f TF/s  (rounds %d, tail slices %d)" % (rows, layers, name, t, fl / t / 1e6, buf[2], buf[7]))
