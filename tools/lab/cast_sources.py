#!/usr/bin/env python3
"""Who calls hero_cast / hero_transpose_cast inside a steady-state micro-step (they should be rare: weight copies are
refreshed by ONE hero_copy_multi per optimiser step)."""
import collections, json, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import hero_amd
from hero_amd import _lib as L
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_prof_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
model = bench.build_model(dev, cfgp)
tr = TrainStep(model, use_graph=False, static_usage=True)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev)
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
lib = L.lib()
log = collections.Counter()


class Wrap:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        fr = [f for f in traceback.extract_stack()[:-1] if "hero_amd/" in f.filename]
        key = (self.name, " < ".join("%s:%d" % (f.filename.split("hero_amd/")[-1], f.lineno) for f in fr[-4:]), str(a[2]) if self.name == "hero_cast" else "%sx%s" % (a[2], a[3]))
        log[key] += 1
        return self.fn(*a)


class LibProxy:
    def __init__(self, real):
        object.__setattr__(self, "_real", real)

    def __getattr__(self, k):
        v = getattr(self._real, k)
        return Wrap(k, v) if k in ("hero_cast", "hero_transpose_cast") else v


proxy = LibProxy(lib)
L.lib = lambda: proxy
N = 2
for _ in range(N):
    tr.micro_step(batch)
torch.cuda.synchronize()
for key, n in sorted(log.items(), key=lambda kv: -kv[1]):
    print("%5.1f  %s  n=%s  %s" % (n / N, key[0], key[2], key[1]))
