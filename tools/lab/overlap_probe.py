#!/usr/bin/env python3
"""Can the batched weight-gradient launch (LDS-bound, matrix pipe half idle, ~2 TB/s of HBM) run UNDER the HBM-bound kernels of
the backward chain?  Stream A: hero_wgrad_batch over the D2 cross-modal shapes (6 layers x 4 weights, 12000 rows).  Stream B, the
"chain" of the same 6 layers: (small) 2 x LayerNorm backward + attention backward per layer; (full) the same plus the four dgrad
GEMMs.  Serial = A then B on one stream; concurrent = A on a side stream while B runs; wall time to both done (events)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import hero_amd
from hero_amd import _lib as Lb
from hero_amd import functional as HF

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1)


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * scale).to(dtype)


rows, layers = 12000, 6
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
dys, xs, outs = [], [], []
for l in range(layers):
    for n, k in shapes:
        dys.append(rnd(rows, n)); xs.append(rnd(rows, k)); outs.append(torch.zeros(n, k, device=dev))
n = len(dys)
probs = (Lb.WgradProblem * n)()
for i in range(n):
    m_, k_ = outs[i].shape
    probs[i] = Lb.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), m_, k_, m_, k_, k_, 4)
buf = np.zeros(8 + 8 * 256 * 16, dtype=np.int32)
words = Lb.lib().hero_wgrad_batch_plan(probs, n, rows, buf.ctypes.data, buf.size)
plan = torch.from_numpy(buf[:words].copy()).cuda()


def run_a():
    Lb.check(Lb.lib().hero_wgrad_batch(probs, n, rows, Lb.BF16, plan.data_ptr(), words, Lb.stream()))


# the chain's operands
S, Lq, H, D = 500, 24, 12, 768
y = rnd(rows, D); dout = rnd(rows, D); gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
_, mean, rstd, _ = HF.k_ln_fwd(y, gamma, beta, 1e-12, y.dtype, rows, D)
qkv = rnd(rows, 3 * D, scale=0.5)
mask_add = torch.zeros(S, Lq, device=dev)
ctxt, saved = HF.k_attn_fwd(qkv, mask_add, S, Lq, H)
dctx = rnd(rows, D)
W1t = rnd(D, 3072, scale=0.02); W2t = rnd(3072, D, scale=0.02); Wot = rnd(D, D, scale=0.02); Wqkvt = rnd(D, 3 * D, scale=0.02)
d3072 = rnd(rows, 3072); dqkv_in = rnd(rows, 3 * D)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)


def run_b(full):
    for _ in range(layers):
        HF.k_ln_bwd(y, dout, gamma, mean, rstd, dgamma=dg, dbeta=db, grad_beta=1.0, want_params=False)
        if full:
            HF.k_dgrad_t(dout, W2t)          # 12000 x 768 -> 3072  (N = 3072, K = 768)
            HF.k_dgrad_t(d3072, W1t)         # 12000 x 3072 -> 768
        HF.k_ln_bwd(y, dout, gamma, mean, rstd, dgamma=dg, dbeta=db, grad_beta=1.0, want_params=False)
        if full:
            HF.k_dgrad_t(dout, Wot)
        HF.k_attn_bwd(qkv, saved, dctx, S, Lq, H, ctx=ctxt, mask_add=mask_add)
        if full:
            HF.k_dgrad_t(dqkv_in, Wqkvt)
    HF.colsum_flush()


side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def timed(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def serial(full):
    run_a(); run_b(full)


def concurrent(full, a_first=True):
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        run_a()
        done = torch.cuda.Event()
        done.record(side)
    run_b(full)
    main.wait_event(done)


print("A alone (wgrad batch, %d problems): %.3f ms" % (n, timed(run_a)))
for full in (False, True):
    tb = timed(lambda: run_b(full))
    ts = timed(lambda: serial(full))
    tc = timed(lambda: concurrent(full))
    print("chain %-5s: B alone %.3f ms, serial A+B %.3f ms, concurrent %.3f ms  (gain %.3f ms)" % ("full" if full else "small", tb, ts, tc, ts - tc))
