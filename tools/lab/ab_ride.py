#!/usr/bin/env python3
"""Same-process A/B of functional.WGRAD_RIDE_MAX_ROWS (bias sums riding on hero_wgrad_batch or as column sums of their own) on the
replayed D2 / D2r micro-step: alternating, twice, 60 steps each."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import hero_amd
from hero_amd import functional as HF
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_ab_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))


def run(cap, ragged, d4=False):
    HF.WGRAD_RIDE_MAX_ROWS[0] = cap
    if d4:
        batch = make_batch("D4", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, videos=256)
    else:
        batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, ragged=ragged)
    model = bench.build_model(dev, cfgp)
    tr = TrainStep(model, use_graph=not d4, static_usage=True, uniform_shapes=not d4)
    tr.prepare(batch)
    n = 6 if d4 else 60
    for _ in range(2 if d4 else 10):
        tr.micro_step(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        tr.micro_step(batch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    del batch
    del tr, model
    HF.reset_caches()
    torch.cuda.empty_cache()
    return ms


BIG = 1 << 30
if "D4" in sys.argv[1:]:
    for rep in range(2):
        for cap in (BIG, 0):
            print("D4 (256 videos) ride %s: %.3f ms" % ("on " if cap else "off", run(cap, False, d4=True)), flush=True)
else:
    for ragged in (False, True):
        for rep in range(2):
            for cap in (BIG, 0):
                print("%s ride %s: %.3f ms" % ("D2r" if ragged else "D2 ", "on " if cap else "off", run(cap, ragged)), flush=True)
