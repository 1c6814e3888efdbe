#!/usr/bin/env python3
"""Which host thread count suits the CPU oracle on the GPU box (bench.py cpu_baseline)?  One warm-up + 2 timed per setting."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hero_amd.synth import make_batch
from oracle import hero_oracle as O

cfg_path = "/tmp/hero_probe.json"
json.dump(bench.HERO_BASE, open(cfg_path, "w"))
model = bench.build_model(torch.device("cpu"), cfg_path)
P = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point() and not k.endswith("pad")) for k, v in model.state_dict().items()}
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, videos=8)
ocfg = O.cfg_from_json(bench.HERO_BASE)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    ts = []
    for i in range(3):
        for p in P.values(): p.grad = None
        t0 = time.perf_counter()
        sum(O.vsm_losses(batch, P, ocfg, p_drop=0.1)).backward()
        ts.append(time.perf_counter() - t0)
    print("threads %3d: %.2f s -> %.2f videos/s" % (n, min(ts[1:]), 8 / min(ts[1:])), flush=True)
