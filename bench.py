#!/usr/bin/env python3
"""Headline benchmark: videos/sec of a HERO-base TVR finetune training micro-step (forward + VSM/VCMR
loss + backward, gradient sync + clip + AdamW on every 2nd micro-step as in
config/train-tvr-8gpu.json) on synthetic TVR-shaped batches, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HERO_BASE = {  # config/hero_finetune.json
    "f_config": dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                     hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                     max_position_embeddings=514, num_attention_heads=12, num_hidden_layers=6,
                     type_vocab_size=2, vocab_size=50272),
    "c_config": dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                     hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                     max_position_embeddings=514, num_attention_heads=12, num_hidden_layers=3,
                     type_vocab_size=2),
    "q_config": dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                     hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                     num_attention_heads=12, max_position_embeddings=514, num_hidden_layers=0,
                     type_vocab_size=1, vocab_size=50272),
}
VFEAT = 4352
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
SLOT_NAMES = {4: "gemm_glds_kernel<bf16> (K-contiguous operands: forward x W^T and dgrad dY (W^T)^T)",
              5: "gemm_kernel<bf16,K,O>", 7: "gemm_glds_tr_kernel (bf16 wgrad dY^T X)", 6: "gemm_kernel<bf16,O,K>"}


def algorithmic_flops_per_video(sh):
    """per BertLayer forward 24 M d^2 + 4 M L d; projections 2 M 4352 d;
    step = 3 x forward (backward = 2 x forward)."""
    d, B = 768, sh["videos"]
    Lf = sh["fps"] + sh["toks"]
    Mf, Mq, Mc = B * sh["subs"] * Lf, B * sh["qtoks"], B * sh["frames"]
    layer = lambda M, L: 24 * M * d * d + 4 * M * L * d                    # noqa: E731
    fwd = 6 * layer(Mf, Lf) + 6 * layer(Mq, sh["qtoks"]) + 3 * layer(Mc, sh["frames"])
    fwd += 2 * (B * sh["subs"] * sh["fps"]) * VFEAT * d + 2 * Mc * VFEAT * d
    fwd += 2 * Mq * d * d + 8 * Mq * d * d + 4 * Mq * sh["qtoks"] * d      # query head proj + attention
    return 3.0 * fwd / B


def build_model(device, cfg_path):
    from hero_amd.model import HeroForVcmr
    from hero_amd.utils.misc import set_dropout
    torch.manual_seed(0)
    model = HeroForVcmr.from_pretrained(
        cfg_path, {}, vfeat_dim=VFEAT, max_frm_seq_len=100, lw_neg_ctx=8.0, lw_neg_q=8.0,
        lw_st_ed=0.01, ranking_loss_type="hinge", use_hard_negative=False, hard_pool_size=20,
        margin=0.1, use_all_neg=True, drop_svmr_prob=0.0)
    model.to(device)
    set_dropout(model, 0.1)
    model.train()
    return model


def cpu_baseline(model, cfg, sample_videos=8, reps=3):
    """The CPU oracle (kind 'port': verified == reference in tests/test_oracle_golden.py) timed on
    this box's host cores on a bounded sample of the same workload."""
    from hero_amd.synth import make_batch
    from oracle import hero_oracle as O
    P = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point() and not k.endswith("pad"))
         for k, v in model.state_dict().items()}
    batch = make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1, videos=sample_videos)
    ocfg = O.cfg_from_json(cfg)
    times = []
    for i in range(reps + 1):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        losses = O.vsm_losses(batch, P, ocfg, p_drop=0.1)
        sum(losses).backward()
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return {"value": sample_videos / med, "unit": "videos/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d-video slice of the D2 batch, fwd+loss+bwd fp32, dropout 0.1, 1 warm-up + %d timed, "
                      "median %.2f s" % (sample_videos, reps, med)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (N=1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if os.environ.get("HERO_BENCH_ONE_DEVICE"):      # plumbing check of the N>1 path on a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HERO_BENCH_BACKEND", "nccl")        # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    import hero_amd
    from hero_amd import _lib as L
    from hero_amd.step import TrainStep
    from hero_amd.synth import SHAPES, make_batch
    hero_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    cfg_path = "/tmp/hero_finetune_bench_%d.json" % rank
    with open(cfg_path, "w") as f:
        json.dump(HERO_BASE, f)
    model = build_model(device, cfg_path)
    trainer = TrainStep(model, use_graph=(world == 1 and not args.no_graph),
                        static_usage=True)     # drop_svmr_prob = 0: every step uses the same parameters
    batch = make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank, device=device)
    sh = SHAPES["D2"]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    trainer.prepare(batch)                           # hipGraph capture is setup, never inside the timed region
    for _ in range(args.warmup):
        trainer.micro_step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.micro_step(batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)

    # ---- roofline leg: HIP events around every GEMM launch over extra, identical steps ----------
    roof = None
    if rank == 0:
        L.check(L.lib().hero_prof_enable(1))
        trainer.use_graph = False                    # events cannot be recorded inside a replay
        for _ in range(args.profile_steps):
            trainer.micro_step(batch)
        torch.cuda.synchronize()
        best = None
        for slot in range(8):
            ms, fl, n = C.c_double(), C.c_double(), C.c_longlong()
            L.check(L.lib().hero_prof_read(slot, C.byref(ms), C.byref(fl), C.byref(n)))
            if n.value and (best is None or ms.value > best[1]):
                best = (slot, ms.value, fl.value, n.value)
        L.check(L.lib().hero_prof_enable(0))
        if best:
            slot, ms, fl, n = best
            ach = fl / (ms * 1e-3) / 1e12
            peak = BF16_PEAK_TFLOPS if slot >= 4 else 157.3
            traffic, tsrc = None, None
            try:                                   # HBM bytes per launch from the committed PMC passes
                pj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
                pm = json.load(open(pj))
                want = {4: "gemm_glds_kernel<unsigned short", 5: "gemm_kernel<unsigned short, 0, 1",
                        7: "gemm_glds_tr_kernel"}.get(slot)
                hits = [v for k, v in pm.items() if want and want in k]
                if hits:
                    tot = sum(h["launches"] for h in hits)
                    traffic = sum(h["hbm_bytes_per_launch_corrected"] * h["launches"] for h in hits) / tot
                    tsrc = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 gfx950 correction)"
            except Exception:
                pass
            roof = {"bound": "mfma", "kernel": SLOT_NAMES.get(slot, "gemm slot %d" % slot),
                    "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc, "launches": n,
                    "avg_launch_us": round(ms * 1e3 / n, 2),
                    "flops_per_launch": fl / n}
    elif world > 1:
        for _ in range(args.profile_steps):
            trainer.micro_step(batch)
    launch_mode = "hipGraph replay" if (world == 1 and not args.no_graph) else "eager"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, HERO_BASE)

    if rank == 0:
        vps = sh["videos"] * world * args.steps / dt
        fl_video = algorithmic_flops_per_video(sh)
        out = {
            "metric": "videos/sec training step, HERO-base TVR-shaped batch",
            "value": round(vps, 2), "unit": "videos/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: HERO-base TVR finetune micro-step (train-tvr-8gpu.json shapes: "
                                   "32 videos x 60 frames, 15 subs x (4 frames + 20 tokens), 15-token query; "
                                   "vfeat 4352; fwd + VSM loss + bwd, all-reduce/clip/AdamW every 2nd micro-step)",
                       "global_batch": sh["videos"] * world, "parallelism": "dp%d" % world,
                       "dropout": 0.1, "grad_accum": 2, "launch": launch_mode},
            "step_tflops": round(vps * fl_video / 1e12, 1),
            "step_frac_of_bf16_peak": round(vps * fl_video / 1e12 / world / BF16_PEAK_TFLOPS, 4),
            "final_loss": loss_val,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
