#!/usr/bin/env python3
"""Same-box A/B of what rides on the batched weight-gradient launch (gemm_wsb_kernel 1.317 -> 1.396 ms between
rounds 3 and 4, 0.9 -> 1.02 GB per launch).  Variants, all in ONE process, alternating:
  prod     round-4/5 default: the QKV AND FFN1 bias sums ride on hero_wgrad_batch's loader waves
  b1epi    round-3 state: the FFN1 bias sum comes from the gelu' GEMM epilogue's fp32 atomics, only the QKV one rides
  noride   no bias sum rides (deferred column sums of their own)
For each: the replayed micro-step (ms, 40 steps) and the wsb launches by HIP events (hero_prof slot 9, 4 eager steps)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import hero_amd
from hero_amd import functional as HF, _lib as L
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_ab_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev)
VARIANTS = {"prod": (True, False), "b1epi": (True, True), "noride": (False, False)}


def run(name):
    HF.WGRAD_DBIAS_RIDE[0], HF.B1_EPILOGUE[0] = VARIANTS[name]
    model = bench.build_model(dev, cfgp)
    tr = TrainStep(model, use_graph=True, static_usage=True, uniform_shapes=True)
    tr.prepare(batch)
    for _ in range(8):
        tr.micro_step(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        tr.micro_step(batch)
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / 40
    tr.use_graph = False
    L.check(L.lib().hero_prof_enable(1))
    for _ in range(4):
        tr.micro_step(batch)
    torch.cuda.synchronize()
    out = {}
    for slot in (8, 9):
        ms, fl, n = C.c_double(), C.c_double(), C.c_longlong()
        L.check(L.lib().hero_prof_read(slot, C.byref(ms), C.byref(fl), C.byref(n)))
        out[slot] = (ms.value / 4, n.value / 4)
    L.check(L.lib().hero_prof_enable(0))
    print("%-7s step %.3f ms | wsb (slot 9) %.3f ms in %.0f launches | K,K 192x192 (slot 8) %.3f ms in %.0f launches" % (
        name, step_ms, out[9][0], out[9][1], out[8][0], out[8][1]), flush=True)
    del tr, model
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    torch.cuda.empty_cache()


for rep in range(2):
    for v in ("prod", "b1epi", "noride"):
        run(v)
