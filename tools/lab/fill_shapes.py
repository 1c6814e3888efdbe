"""Which tensors does a D2 micro-step zero-fill?  aten::zero_/fill_/zeros with shapes and python stacks (eager)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

dev = torch.device("cuda", 0)
path = "/tmp/hero_fill.json"
json.dump(bench.HERO_BASE, open(path, "w"))
model = bench.build_model(dev, path)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev)
tr = TrainStep(model, use_graph=False, static_usage=True)
tr.prepare(batch)
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU],
                            record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        tr.micro_step(batch)
    torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::add", "aten::add_", "aten::mul") and e.device_time_total > 3:
        print("%-16s %8.1f us  %s" % (e.name, e.device_time_total, e.input_shapes))
        for s in (e.stack or [])[:6]:
            if "hero_amd" in s or "bench" in s:
                print("        ", s)
