#!/usr/bin/env python3
"""Which PyTorch-side (aten) ops still run in the training micro-step, by GPU time (eager mode)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hero_amd
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_prof_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
model = bench.build_model(dev, cfgp)
tr = TrainStep(model, use_graph=False)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev)
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(4):
        tr.micro_step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t / 4.0, e.count / 4.0, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total device time per step (us):", round(tot))
for t, n, k in rows[:70]:
    print("%9.1f us  %6.1f calls  %s" % (t, n, k[:110]))
