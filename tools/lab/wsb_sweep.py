#!/usr/bin/env python3
"""hero_wgrad_batch timing sweep: rows x layer count x shape mix (run with HERO_HIP_LIB=tools/lab/libhero_<variant>.so for
the ablations of build_variants.sh).  usage: wsb_sweep.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hero_amd import _lib as L
from ln_bench import timeit
dt = torch.bfloat16
LAYER = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
tag = sys.argv[1] if len(sys.argv) > 1 else "product"
quick = tag != "product"


def run(rows, shapes, label, same_tile=False):
    dys = [torch.randn(rows, n, device="cuda").to(dt) for n, _ in shapes]
    xs = [torch.randn(rows, k, device="cuda").to(dt) for _, k in shapes]
    outs = [torch.zeros(n, k, device="cuda") for n, k in shapes]
    n = len(dys)
    pr = (L.WgradProblem * n)(*[L.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), outs[i].shape[0], outs[i].shape[1],
                                               outs[i].shape[0], outs[i].shape[1], outs[i].shape[1], 4) for i in range(n)])
    buf = np.zeros(8 + 8 * 256 * 16, dtype=np.int32)
    words = L.lib().hero_wgrad_batch_plan(pr, n, rows, buf.ctypes.data, buf.size)
    if same_tile:             # every workgroup on the SAME tile (all operand reads hit the L2): timing only, use a noepi build
        it = buf[8:words].reshape(-1, 8)
        it[:, 0], it[:, 1], it[:, 2] = 0, 0, 0
    plan = torch.from_numpy(buf[:words].copy()).cuda()
    fn = lambda: L.check(L.lib().hero_wgrad_batch(pr, n, rows, L.BF16, plan.data_ptr(), words, L.stream()))
    fn(); torch.cuda.synchronize()
    t = timeit(fn, n=6)
    fl = sum(2.0 * rows * o.shape[0] * o.shape[1] for o in outs)
    steps = (rows + 63) // 64
    print("[%s] %-34s rows %5d tiles %4d rounds %d slices %d: %8.1f us %6.0f TF/s  (%.3f us per k-step of a full round)"
          % (tag, label, rows, buf[6], buf[2], buf[7], t, fl / t / 1e6, t / (buf[6] / 256.0 * steps)), flush=True)


for rows in ((12032,) if quick else (3008, 6016, 12032)):
    run(rows, LAYER * 4, "4 layers (3 full rounds)")
if "noepi" in tag or "nomfma" in tag:
    run(12032, LAYER * 4, "4 layers, every WG the same tile", same_tile=True)
if not quick:
    run(12000, LAYER * 6, "6 layers (benched step)")
    run(1920, LAYER * 3, "3 layers (Temporal Transformer)")
run(12032, [(768, 3072)] * 12, "12 x (768, 3072)")
run(12032, [(3072, 768)] * 12, "12 x (3072, 768)")
run(12032, [(2304, 768)] * 16, "16 x (2304, 768)")
run(12032, [(768, 768)] * 32, "32 x (768, 768)")
