#!/usr/bin/env python3
"""Who issues the device-to-device copies (__amd_rocclr_copyBuffer in the kernel trace) of a micro-step: chrome trace of
two steps; every hipMemcpy* runtime call is matched to the CPU ops that enclose it.  `ragged` = the D2r batch, `D4` = configs[4] shapes (96 videos)."""
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
if "D4" in sys.argv:       # configs[4] shapes at a small batch: 96 videos x 256 frames
    batch = make_batch("D4", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, videos=96)
else:
    batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, ragged="ragged" in sys.argv)
for _ in range(4):
    tr.micro_step(batch)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    for _ in range(N):
        tr.micro_step(batch)
    torch.cuda.synchronize()
prof.export_chrome_trace("/tmp/hero_trace.json")
tr_ = json.load(open("/tmp/hero_trace.json"))["traceEvents"]
cpu_ops = [e for e in tr_ if e.get("cat") in ("cpu_op", "user_annotation", "python_function") and "dur" in e]
rt = [e for e in tr_ if e.get("cat") in ("cuda_runtime", "hip_runtime", "cuda_driver") and "emcpy" in e.get("name", "")]
dev_copies = {e["args"].get("correlation"): e for e in tr_ if e.get("cat") in ("gpu_memcpy", "gpu_memset") and "args" in e}
count, dur, size = collections.Counter(), collections.Counter(), collections.Counter()
for r in rt:
    ts, tid = r["ts"], r.get("tid")
    encl = [o for o in cpu_ops if o.get("tid") == tid and o["ts"] <= ts <= o["ts"] + o["dur"]]
    encl.sort(key=lambda o: o["dur"])
    names = [o["name"].replace("autograd::engine::evaluate_function: ", "bwd:") for o in encl[:5]]
    shapes = str(encl[0].get("args", {}).get("Input Dims", ""))[:60] if encl else ""
    d = dev_copies.get(r.get("args", {}).get("correlation"))
    key = (r["name"], " < ".join(names), shapes)
    count[key] += 1
    if d is not None:
        dur[key] += d.get("dur", 0)
        size[key] += d.get("args", {}).get("bytes", 0) or 0
print("%6s %8s %8s  runtime call | enclosing ops | input dims" % ("calls", "us/step", "MB/step"))
for key, n in sorted(count.items(), key=lambda kv: -dur[kv[0]])[:30]:
    print("%6.1f %8.1f %8.1f  %s | %s | %s" % (n / N, dur[key] / N, size[key] / N / 2 ** 20, *key))
