#!/usr/bin/env python3
"""Who issues the remaining PyTorch-side fill / copy / add kernels of the micro-step, INCLUDING the ones the autograd
engine's worker thread issues (tools/torch_ops_sources.py sees the calling thread only): for every such aten op the
chain of enclosing profiler ranges (autograd node names) and the first hero_amd frame of its Python stack."""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
tr = TrainStep(model, use_graph=False)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, ragged="ragged" in sys.argv)   # "ragged": the D2r batch
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        tr.micro_step(batch)
    torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::sum", "aten::cat",
        "aten::index_select", "aten::where", "aten::div", "aten::sub", "aten::neg", "aten::_to_copy", "aten::clone")
count = collections.Counter()
dtime = collections.Counter()
for e in prof.events():
    if e.name not in WANT:
        continue
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0:
        continue
    chain, p = [], e.cpu_parent
    while p is not None:
        if not p.name.startswith("aten::") or p.name in ("aten::zeros", "aten::zeros_like", "aten::ones", "aten::full", "aten::to", "aten::contiguous"):
            chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
        p = p.cpu_parent
    frame = ""
    for f in (e.stack or []):
        if "hero_amd/" in f or "bench.py" in f:
            frame = f.split("/root/repo/")[-1] if "/root/repo/" in f else f
            break
    shapes = str([sh for sh in (e.input_shapes or []) if sh])[:60]
    key = (e.name, " < ".join(chain[:3]), frame[:90] + " " + shapes)
    count[key] += 1
    dtime[key] += dt
print("%7s %8s  op / enclosing ranges / first hero_amd frame" % ("calls", "us/step"))
for key, n in sorted(count.items(), key=lambda kv: -dtime[kv[0]])[:70]:
    print("%7.1f %8.1f  %s | %s | %s" % (n / N, dtime[key] / N, key[0], key[1], key[2]))
print("total: %.1f calls, %.1f us per step" % (sum(count.values()) / N, sum(dtime.values()) / N))
