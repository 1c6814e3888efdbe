"""Per-task view of the D3 (configs[3]) micro-step: eager launches under torch.profiler, the kernels that are NOT this
library's (aten / vendor GEMM / copies) by total time, and the step's kernel sum.  python tools/lab/d3_task_profile.py [task ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from hero_amd.step import TrainStep  # noqa: E402
from hero_amd.synth import make_pretrain_batches  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = json.loads(json.dumps(bench.HERO_BASE))
    cfg["f_config"]["vocab_size"] = 50265
    path = "/tmp/hero_d3_prof.json"
    with open(path, "w") as f:
        json.dump(cfg, f)
    model = bench.build_model(dev, path, pretraining=True)
    batches = make_pretrain_batches("D2", vfeat_dim=bench.VFEAT, vocab=50265, seed=1, device=dev)
    trainer = TrainStep(model, opts=dict(learning_rate=3e-5, gradient_accumulation_steps=2), task="vsm", use_graph=False)
    tasks = sys.argv[1:] or ["mlm", "mfm-nce", "fom", "vsm"]
    for task in tasks:
        trainer.prepare(batches[task], task)
        for _ in range(4):
            trainer.micro_step(batches[task], task)
        torch.cuda.synchronize()
        n = 4
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
            for _ in range(n):
                trainer.micro_step(batches[task], task)
            torch.cuda.synchronize()
        ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
        total = sum(e.device_time_total for e in ev) / n
        ours = ("hero", "gemm_ws", "colsum_multi", "gemm_skinny", "attn_", "ce_", "k_")
        other = [e for e in ev if not any(s in e.key for s in ours)]
        print("== %s: kernel time %.3f ms / micro-step, not ours %.3f ms in %.1f launches" %
              (task, total / 1e3, sum(e.device_time_total for e in other) / n / 1e3, sum(e.count for e in other) / n))
        for e in sorted(other, key=lambda e: -e.device_time_total)[:14]:
            print("   %7.1f us/step  %5.1f x  %s" % (e.device_time_total / n, e.count / n, e.key[:150]))
        for e in sorted(ev, key=lambda e: -e.device_time_total)[:int(os.environ.get("TOPN", "6"))]:
            print("   top  %7.1f us/step  %5.1f x  %s" % (e.device_time_total / n, e.count / n, e.key[:120]))


if __name__ == "__main__":
    main()
