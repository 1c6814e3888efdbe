"""Which groups does the deferred weight-gradient queue launch in the D4 step (256 videos)?  Logs every hero_wgrad_batch call of one eager
micro-step: rows, problems, tiles, plan rounds / slices, and its duration by HIP events."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import hero_amd
from hero_amd import functional as HF, _lib as L
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_d4_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
model = bench.build_model(dev, cfgp)
tr = TrainStep(model, use_graph=False, static_usage=True)
wl = sys.argv[2] if len(sys.argv) > 2 else "D4"
batch = make_batch(wl, vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev, videos=int(sys.argv[1]) if len(sys.argv) > 1 else 256)
for _ in range(2):
    tr.micro_step(batch)
torch.cuda.synchronize()
log = []
orig = L.lib().hero_wgrad_batch


class Wrap:
    def __getattr__(self, k):
        return getattr(real, k)


real = L.lib()
calls = []
_plan = HF._wgrad_plan


def plan_logged(probs, n, rows, device):
    r = _plan(probs, n, rows, device)
    tiles = sum(-(-probs[i].M // 192) * -(-probs[i].N // 192) for i in range(n))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    calls.append([rows, n, tiles, None if r is None else int(r[0][2].item()), None if r is None else int(r[0][7].item()), e0, e1, _WQB()])
    return r


def _WQB():
    return HF._WQ_BYTES[0] >> 20


if "noride" in sys.argv:
    HF.WGRAD_DBIAS_RIDE[0] = False                   # bias sums as column sums of their own instead of riding on the loader waves
HF._wgrad_plan = plan_logged
_wf = HF.wgrad_flush


def flush_logged():
    n0 = len(calls)
    torch.cuda.current_stream().synchronize() if False else None
    s = torch.cuda.Event(enable_timing=True); s.record()
    _wf()
    e = torch.cuda.Event(enable_timing=True); e.record()
    for c in calls[n0:]:
        c[5], c[6] = s, e


HF.wgrad_flush = flush_logged
tr.micro_step(batch)
tr.micro_step(batch)
torch.cuda.synchronize()
for c in calls:
    print("rows %7d problems %2d tiles %4d plan rounds %s slices %s  queue %5d MB  flush %.2f ms" % (c[0], c[1], c[2], c[3], c[4], c[7], c[5].elapsed_time(c[6])))
