#!/usr/bin/env python3
"""Which Python lines issue the remaining PyTorch-side kernels of the micro-step (eager)."""
import json, os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hero_amd
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch
from torch.utils._python_dispatch import TorchDispatchMode

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
count = collections.Counter()
SKIP = ("view", "reshape", "detach", "alias", "as_strided", "expand", "t.default", "transpose", "permute", "slice", "select",
        "unsqueeze", "squeeze", "_unsafe_view", "empty", "unbind", "split", "is_", "sym_", "numel", "size", "stride", "narrow")

class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/hero_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    count[(name.replace("aten.", ""), "%s:%d" % (os.path.relpath(fr.filename, "/root/repo"), fr.lineno))] += 1
                    break
        return func(*args, **(kwargs or {}))

with Mode():
    tr.micro_step(batch)
    tr.micro_step(batch)
torch.cuda.synchronize()
for (op, where), n in sorted(count.items(), key=lambda kv: -kv[1])[:60]:
    print("%5.1f /step  %-32s %s" % (n / 2.0, op, where))
print("total dispatched compute ops per step:", sum(count.values()) / 2.0)
