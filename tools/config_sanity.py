#!/usr/bin/env python3
"""Run a few training micro-steps of HERO-base (bf16) on the other BASELINE.json shapes: the ragged TVR
variant (SURVEY 8d D2 'ragged') and the long-video stress shape D4 (256-frame Temporal Transformer),
and report loss / time / memory.  Sanity + sizing tool, not the headline bench."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hero_amd
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_sanity_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
for name, kw in (("D2 ragged", dict(name="D2", ragged=True)), ("D4 long video x8", dict(name="D4", videos=8)),
                 ("D4 long video x32", dict(name="D4", videos=32))):
    model = bench.build_model(dev, cfgp)
    tr = TrainStep(model, use_graph=False)
    batch = make_batch(vfeat_dim=bench.VFEAT, vocab=50272, seed=3, device=dev, **kw)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        loss = tr.micro_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        loss = tr.micro_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    B = batch["c_v_feats"].shape[0]
    print("%-20s videos=%d frames<=%d subs=%d  loss %.4f  %.1f ms/step  %.0f videos/s  peak mem %.1f GB"
          % (name, B, batch["c_v_feats"].shape[1], batch["f_attn_masks"].shape[0], float(loss), dt * 1e3, B / dt,
             torch.cuda.max_memory_allocated() / 2**30), flush=True)
    hero_amd.functional.set_grad_sink(None)
    hero_amd.functional.clear_weight_cache()
    del tr, model, batch
    torch.cuda.empty_cache()
