#!/usr/bin/env python3
"""Who issues the device-to-device copies (__amd_rocclr_copyBuffer) of a micro-step: aten ops whose device activity
is a Memcpy, with their autograd range and first hero_amd frame.  `ragged` = the D2r batch."""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import hero_amd
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch
from torch.profiler import profile, ProfilerActivity

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_prof_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
model = bench.build_model(dev, cfgp)
tr = TrainStep(model, use_graph=False, static_usage=True)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, ragged="ragged" in sys.argv)
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        tr.micro_step(batch)
    torch.cuda.synchronize()
count, dtime = collections.Counter(), collections.Counter()
for e in prof.events():
    ks = [k for k in (e.kernels or []) if "emcpy" in k.name or "copyBuffer" in k.name]
    if not ks:
        continue
    chain, p = [], e.cpu_parent
    while p is not None:
        chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
        p = p.cpu_parent
    frame = ""
    for f in (e.stack or []):
        if "hero_amd/" in f or "bench.py" in f:
            frame = f.split("/root/repo/")[-1] if "/root/repo/" in f else f
            break
    key = (e.name, " < ".join(chain[:4]), frame[:100], str([s for s in (e.input_shapes or []) if s])[:70])
    count[key] += 1
    dtime[key] += sum(k.duration for k in ks)
print("%7s %8s  op / parents / frame / shapes" % ("calls", "us/step"))
for key, t in sorted(dtime.items(), key=lambda kv: -kv[1])[:30]:
    print("%7.1f %8.1f  %s | %s | %s | %s" % (count[key] / N, t / N, *key))
print("total: %.1f copies, %.1f us per step" % (sum(count.values()) / N, sum(dtime.values()) / N))
