#!/usr/bin/env python3
"""Same-process A/B of a functional.* lab switch on the replayed D2 / D2r micro-step: alternating, twice, 60 steps each.
python tools/lab/ab_switch.py NAME VALUE_A VALUE_B   (e.g. B1_PARTIALS 1 0)"""
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
name, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def run(val, ragged):
    getattr(HF, name)[0] = type(getattr(HF, name)[0])(val)
    batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, ragged=ragged)
    model = bench.build_model(dev, cfgp)
    tr = TrainStep(model, use_graph=True, static_usage=True, uniform_shapes=True)
    tr.prepare(batch)
    for _ in range(10):
        tr.micro_step(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        tr.micro_step(batch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    del tr, model, batch
    HF.reset_caches()
    torch.cuda.empty_cache()
    return ms


for ragged in (False, True):
    for rep in range(2):
        for v in (va, vb):
            print("%s %s=%d: %.3f ms" % ("D2r" if ragged else "D2 ", name, v, run(v, ragged)), flush=True)
