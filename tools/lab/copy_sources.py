#!/usr/bin/env python3
"""Who makes large device-to-device copies in a micro-step (__amd_rocclr_copyBuffer in the kernel trace): wraps
Tensor.copy_ / clone / contiguous / torch.cat and the autograd engine's accumulations are visible as the rest.
`ragged` = the D2r batch."""
import collections, json, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
tr = TrainStep(model, use_graph=False, static_usage=True)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, ragged="ragged" in sys.argv)
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
log = collections.Counter()
size = collections.Counter()


def frame():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "hero_amd/" in f.filename and "copy_sources" not in f.filename:
            return "%s:%d %s" % (f.filename.split("hero_amd/")[-1], f.lineno, f.name)
    return "?"


def wrap(name, fn, big):
    def w(*a, **k):
        r = fn(*a, **k)
        t = a[0] if torch.is_tensor(a[0]) else (r if torch.is_tensor(r) else None)
        n = big(a, r)
        if n >= (1 << 20):
            key = (name, frame())
            log[key] += 1
            size[key] += n
        return r
    return w


torch.Tensor.copy_ = wrap("copy_", torch.Tensor.copy_, lambda a, r: a[0].numel() * a[0].element_size() if a[0].is_cuda else 0)
torch.Tensor.clone = wrap("clone", torch.Tensor.clone, lambda a, r: r.numel() * r.element_size() if r.is_cuda else 0)
_c = torch.Tensor.contiguous
torch.Tensor.contiguous = wrap("contiguous", _c, lambda a, r: (r.numel() * r.element_size()) if (r.is_cuda and r.data_ptr() != a[0].data_ptr()) else 0)
N = 2
for _ in range(N):
    tr.micro_step(batch)
torch.cuda.synchronize()
print("%6s %9s  op / first hero_amd frame" % ("calls", "MB/step"))
for key, n in sorted(size.items(), key=lambda kv: -kv[1])[:25]:
    print("%6.1f %9.1f  %s | %s" % (log[key] / N, n / N / 2 ** 20, *key))
