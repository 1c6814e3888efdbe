"""Timing of the StaticBatchFeeder path (hero_amd/loader.py): step / commit / prefetch alone and together."""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
import bench as B
import hero_amd
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch
from hero_amd.loader import StaticBatchFeeder, pin_batch
from hero_amd import functional as HFm
HF_refresh = lambda: HFm.refresh_memo([t for t in fd.static.values() if torch.is_tensor(t)])
hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
json.dump(B.HERO_BASE, open("/tmp/hb.json", "w"))
model = B.build_model(dev, "/tmp/hb.json")
tr = TrainStep(model, use_graph=True, static_usage=True)
host = [pin_batch(make_batch("D2", vfeat_dim=4352, vocab=50272, seed=1 + 100 * i)) for i in range(4)]
fd = StaticBatchFeeder(host[0], dev)
tr.prepare(fd.static)
fd.capture()
def T(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("step only        %.3f ms" % T(lambda i: tr.micro_step(fd.static)))
def c1(i):
    fd.prefetch(host[i % 4]); fd.commit()
print("prefetch+commit  %.3f ms" % T(c1))

fd.prefetch(host[0])
def full(i):
    b = fd.commit(); tr.micro_step(b); fd.prefetch(host[(i + 1) % 4])
print("commit+prefetch+step %.3f ms" % T(full))
fd.commit()
# pieces of the commit (eager, GPU time by events)
def ev(name, fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print("  %-22s %.1f us (eager, incl. launch gaps)" % (name, e0.elapsed_time(e1) / n * 1e3))
ev("payload D2D", lambda: [fd.static[k].copy_(fd.stage[0][k]) for k in fd.shapes])
ev("load_lengths", lambda: fd.dc.load_lengths(fd.stage_len[0], src_device=True))
ev("rebuild", lambda: fd.dc.rebuild(c_v_feats=fd.static["c_v_feats"]))
ev("refresh_memo", lambda: HF_refresh())
# round 5: where do the +0.6 ms of the fed step (7.00 vs 6.40 ms in the driver's run) come from?
ev("commit graph replay", lambda: fd._graph[0].replay())
def host_ms(name, fn, n=20):
    torch.cuda.synchronize(); t = 0.0
    for i in range(n):
        t0 = time.perf_counter(); fn(i); t += time.perf_counter() - t0
        torch.cuda.synchronize()
    print("  host time of %-18s %.3f ms per call" % (name, t / n * 1e3))
fd.prefetch(host[1])
def hc(i):
    fd.commit(); fd.prefetch(host[i % 4])
host_ms("commit + prefetch", hc)
fd.commit()
host_ms("prefetch alone", lambda i: (fd.prefetch(host[i % 4]), fd.commit()) and None)
host_ms("step replay launch", lambda i: tr.micro_step(fd.static))
fd.skip_h2d = True
fd.prefetch(host[0])
print("commit+prefetch(no H2D)+step %.3f ms" % T(full))
fd.commit()
fd.skip_h2d = False
fd.prefetch(host[0])
print("commit+prefetch+step (again) %.3f ms" % T(full, n=40))
fd.commit()
print("step only (again) %.3f ms" % T(lambda i: tr.micro_step(fd.static), n=40))
