#!/usr/bin/env python3
"""Host-side (Python) cost of one eager micro-step: cProfile over a few steps, top entries by own time."""
import cProfile, json, os, pstats, sys, time
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
N = 8
t0 = time.perf_counter()
for _ in range(N):
    tr.micro_step(batch)
t1 = time.perf_counter()          # host issue time only (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.2f ms/step, with drain %.2f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.micro_step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = []
for (fn, ln, name), (cc, nc, tt, ct, _) in st.stats.items():
    rows.append((tt / N * 1e3, ct / N * 1e3, nc / N, "%s:%d(%s)" % (os.path.basename(fn), ln, name)))
rows.sort(reverse=True)
for tt, ct, nc, nm in rows[:45]:
    print("%7.3f ms own  %7.3f ms cum  %7.1f calls  %s" % (tt, ct, nc, nm))
