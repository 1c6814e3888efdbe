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
SLOT_NAMES = {4: "gemm_glds_kernel<bf16> (4-wave tiles, K-contiguous operands: the small-M forward / dgrad GEMMs)",
              5: "gemm_kernel<bf16,K,O>", 7: "gemm_glds_tr_kernel (bf16 wgrad dY^T X, 128x128 tiles)", 6: "gemm_kernel<bf16,O,K>",
              8: "gemm_ws_kernel<3,3,K,K> (wave-specialised persistent 192x192 tiles: forward x W^T and dgrad dY (W^T)^T "
                 "of the M = 12000 row batch, fused epilogues)",
              9: "gemm_ws_kernel<3,3,O,O> (wave-specialised 192x192 wgrad dY^T X)",
              10: "gemm_ws_kernel<2,3,K,K> (wave-specialised 128x192 tiles: the 1920-row GEMMs with N >= 2304)"}
def _latest_profile(suffix):
    """The newest committed PMC summary (profiles/rNN_<suffix>); it is stamped with the kernel-source hash it was taken with."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return os.path.basename(hits[-1]) if hits else "r04_" + suffix


PROFILE_TRAFFIC = os.environ.get("HERO_PROFILE_TRAFFIC") or _latest_profile("pmc_traffic.json")
PROFILE_MFMA = os.environ.get("HERO_PROFILE_MFMA") or _latest_profile("pmc_mfma.json")
HBM_PEAK_GBPS = 8000.0        # HBM3E spec, MI355X_MICROARCH.md (6.3 TB/s achievable)


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


def algorithmic_flops_of_batch(batch):
    """Same accounting on an arbitrary (ragged) batch: per sequence of L valid tokens one BertLayer forward is
    24 L d^2 + 4 L^2 d; padded positions do not count."""
    d = 768
    layer = lambda lens: sum(24 * int(n) * d * d + 4 * int(n) * int(n) * d for n in lens)     # noqa: E731
    lf = batch["f_attn_masks"].sum(1).tolist()
    lq = batch["query_attn_masks"].sum(1).tolist()
    lc = batch["c_attn_masks"].sum(1).tolist()
    fwd = 6 * layer(lf) + 6 * layer(lq) + 3 * layer(lc)
    n_fv = int((batch["f_attn_masks"][:, :batch["f_v_feats"].shape[1]] != 0).sum())
    fwd += 2 * n_fv * VFEAT * d + 2 * int(sum(lc)) * VFEAT * d
    fwd += sum(10 * int(n) * d * d + 4 * int(n) * int(n) * d for n in lq)
    return 3.0 * fwd


def build_model(device, cfg_path, pretraining=False):
    if pretraining:
        return build_pretraining_model(device, cfg_path)
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


def build_pretraining_model(device, cfg_path):
    """config/hero_pretrain.json + pretrain.py:60-80: vocabulary 50265 padded to 50272 by pad_vocab()."""
    from hero_amd.model import HeroForPretraining
    from hero_amd.utils.misc import set_dropout
    torch.manual_seed(0)
    model = HeroForPretraining.from_pretrained(
        cfg_path, {}, vfeat_dim=VFEAT, max_frm_seq_len=100, lw_neg_ctx=8.0, lw_neg_q=8.0,
        lw_st_ed=0.01, ranking_loss_type="hinge", use_hard_negative=False, hard_pool_size=20,
        margin=0.1, use_all_neg=True, drop_svmr_prob=0.0)
    model.v_encoder.f_encoder.pad_vocab()
    model.to(device)
    set_dropout(model, 0.1)
    model.train()
    return model


def cpu_baseline(model, cfg, sample_videos=32, reps=3, threads=16):
    """The CPU oracle (kind 'port': verified == reference in tests/test_oracle_golden.py) timed on
    this box's host cores on the SAME workload: the full 32-video D2 batch (rounds 1-3 timed a 16-video
    slice), 1 warm-up + 3 timed steps of ~4.5 s.  16 threads: measured best on the GPU box
    (tools/cpu_threads_probe.py on 256 logical CPUs: 8 -> 4.6, 16 -> 6.6, 32 -> 5.9, 64 -> 2.9, 128 (torch's default)
    -> 1.3 videos/s; the oracle's small per-subtitle ops do not scale past one CCD)."""
    from hero_amd.synth import make_batch
    from oracle import hero_oracle as O
    P = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point() and not k.endswith("pad"))
         for k, v in model.state_dict().items()}
    batch = make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1, videos=sample_videos)
    ocfg = O.cfg_from_json(cfg)
    before = torch.get_num_threads()
    threads = max(1, min(threads, os.cpu_count() or threads))
    torch.set_num_threads(threads)
    times = []
    try:
        for i in range(reps + 1):
            for p in P.values():
                p.grad = None
            t0 = time.perf_counter()
            losses = O.vsm_losses(batch, P, ocfg, p_drop=0.1)
            sum(losses).backward()
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(before)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return {"value": sample_videos / med, "unit": "videos/s", "cores": threads,
            "kind": "port",
            "sample": "%s D2 batch (%d videos), fwd+loss+bwd fp32, dropout 0.1, 1 warm-up + %d timed, "
                      "median %.2f s" % ("the full" if sample_videos == 32 else "a slice of the", sample_videos, reps, med)}


def _timed(trainer, batch, task, steps, warmup, world):
    """prepare (untimed) + warm-up + `steps` micro-steps bracketed by barrier + synchronise; max over ranks."""
    def sync():
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.cuda.synchronize()
    trainer.prepare(batch, task)
    for _ in range(warmup):
        trainer.micro_step(batch, task)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = trainer.micro_step(batch, task)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=batch["c_v_feats"].device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, float(loss)


def secondary_workload(args, device, world, rank, steps=None, warmup=None):
    """The other BASELINE.json configurations as bench lines of their own (same JSON contract, own `metric`).  Returns the
    line (a dict); `python bench.py --workload W` prints it, the default D2 run attaches short versions as `secondary`."""
    from hero_amd.step import TrainStep
    from hero_amd.synth import SHAPES, make_batch, make_pretrain_batches
    cfg = json.loads(json.dumps(HERO_BASE))
    graph = not torch.distributed.is_initialized() and not args.no_graph
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    steps, warmup = steps + steps % 2, warmup + warmup % 2     # whole accumulation windows
    base = {"unit": "videos/s", "n_gpus": world, "steps": steps, "warmup": warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32",
            "data": "synthetic"}
    cfg_path = "/tmp/hero_bench_%s_%d.json" % (args.workload, rank)

    if args.workload == "D2r":
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        model = build_model(device, cfg_path)
        batch = make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank, device=device, ragged=True)
        trainer = TrainStep(model, use_graph=graph, static_usage=True)
        dt, loss = _timed(trainer, batch, None, steps, warmup, world)
        B = batch["c_v_feats"].shape[0]
        vps = B * world * steps / dt
        fl = algorithmic_flops_of_batch({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()})
        out = dict(base, metric="videos/sec training step, HERO-base, ragged TVR batch (30-100 frames, 8-25 subtitles of "
                                "0-8 frames and 4-40 tokens per video)",
                   value=round(vps, 2), ms_per_step=round(dt / steps * 1e3, 3),
                   config={"workload": "configs[1] ragged variant (SURVEY 8d): %d videos, %d subtitle rows, %d valid of %d "
                                       "cross-modal positions (packed), fwd + VSM loss + bwd, clip/AdamW every 2nd micro-step"
                                       % (B, batch["f_attn_masks"].shape[0], int(batch["f_attn_masks"].sum()),
                                          batch["f_attn_masks"].numel()),
                           "global_batch": B * world, "parallelism": "dp%d" % world, "dropout": 0.1, "grad_accum": 2,
                           "launch": "hipGraph replay" if graph else "eager"},
                   step_tflops=round(vps / B * fl / 1e12, 1),
                   step_frac_of_bf16_peak=round(vps / B * fl / 1e12 / world / BF16_PEAK_TFLOPS, 4), final_loss=loss,
                   peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    elif args.workload == "D3":
        cfg["f_config"]["vocab_size"] = 50265                      # config/hero_pretrain.json; pad_vocab() -> 50272
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        model = build_model(device, cfg_path, pretraining=True)
        batches = make_pretrain_batches("D2", vfeat_dim=VFEAT, vocab=50265, seed=1 + rank, device=device)
        opts = dict(learning_rate=3e-5, gradient_accumulation_steps=2)      # config/pretrain-tv-16gpu.json:33-34
        trainer = TrainStep(model, opts=opts, task="vsm", use_graph=graph)
        mix = {"mlm": 2, "mfm-nce": 2, "fom": 1, "vsm": 2}                  # config/pretrain-tv-16gpu.json:11-14
        B = SHAPES["D2"]["videos"]
        per_task, t_mix, losses = {}, 0.0, {}
        for task, w in mix.items():                                         # fixed schedule, one task per window
            dt, losses[task] = _timed(trainer, batches[task], task, steps, warmup, world)
            per_task[task] = {"videos_per_s": round(B * world * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3)}
            t_mix += w * dt / steps
        vps = sum(mix.values()) * B * world / t_mix
        out = dict(base, metric="videos/sec pre-training step, HERO-base, TV multi-task mix mlm:mfm-nce:fom:vsm = 2:2:1:2",
                   value=round(vps, 2), ms_per_step=round(t_mix / sum(mix.values()) * 1e3, 3),
                   config={"workload": "configs[3]: pretrain-tv-16gpu.json shapes per GPU (32 videos x 60 frames, 15 subs x "
                                       "(4 frames + 20 tokens); 15 % masked tokens / frames, 15 % shuffled frames, 5 queries per "
                                       "video; vocabulary 50265 padded to 50272); each task timed on its own, value = the "
                                       "2:2:1:2 time-weighted mix",
                           "global_batch": B * world, "parallelism": "dp%d" % world, "dropout": 0.1, "grad_accum": 2,
                           "launch": "hipGraph replay" if graph else "eager"},
                   per_task=per_task, final_loss=losses,
                   peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    else:   # D4
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        model = build_model(device, cfg_path)
        trainer = TrainStep(model, use_graph=False, static_usage=True)
        total = torch.cuda.get_device_properties(device).total_memory
        B = args.videos
        if B <= 0:      # probe the per-video footprint on a small batch, then size the batch to ~80 % of HBM
            probe = make_batch("D4", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank, device=device, videos=48)
            trainer.micro_step(probe)
            trainer.micro_step(probe)
            torch.cuda.synchronize()
            del probe
            torch.cuda.empty_cache()
            fixed = torch.cuda.memory_allocated()                   # weights, optimiser state, gradient arena, copies
            per_video = (torch.cuda.max_memory_allocated() - fixed) / 48.0
            B = int((0.84 * total - fixed) / per_video)
        while True:     # the caching allocator's peak is not exactly linear in the batch: back off on OOM
            try:
                torch.cuda.reset_peak_memory_stats()
                batch = make_batch("D4", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank, device=device, videos=B)
                dt, loss = _timed(trainer, batch, None, steps, warmup, world)
                break
            except torch.OutOfMemoryError:
                batch = None
                trainer.arena.zero()
                torch.cuda.empty_cache()
                B = int(B * 0.9)
                if B < 8:
                    raise
        vps = B * world * steps / dt
        fl = algorithmic_flops_of_batch({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()})
        peak = torch.cuda.max_memory_allocated()
        out = dict(base, metric="videos/sec training step, HERO-base, long-video stress (256 frames per video)",
                   value=round(vps, 2), ms_per_step=round(dt / steps * 1e3, 3),
                   config={"workload": "configs[4]: %d videos x 256 frames (64 subs x (4 frames + 20 tokens)), 256-frame "
                                       "Temporal Transformer, fwd + VSM loss + bwd, clip/AdamW every 2nd micro-step" % B,
                           "global_batch": B * world, "parallelism": "dp%d" % world, "dropout": 0.1, "grad_accum": 2,
                           "launch": "eager"},
                   step_tflops=round(vps / B * fl / 1e12, 1),
                   step_frac_of_bf16_peak=round(vps / B * fl / 1e12 / world / BF16_PEAK_TFLOPS, 4), final_loss=loss,
                   peak_mem_gb=round(peak / 2 ** 30, 2), hbm_frac=round(peak / total, 3))
    return out


def feed_run(device, rank, steps, warmup, n_batches=4):
    """D2 through hero_amd.loader.StaticBatchFeeder: distinct pinned host batches, H2D one step ahead (the PCIe-inclusive
    rate; never the headline value)."""
    from hero_amd.loader import StaticBatchFeeder, pin_batch
    from hero_amd.step import TrainStep
    from hero_amd.synth import SHAPES, make_batch
    cfg_path = "/tmp/hero_bench_feed_%d.json" % rank
    with open(cfg_path, "w") as f:
        json.dump(HERO_BASE, f)
    model = build_model(device, cfg_path)
    trainer = TrainStep(model, use_graph=True, static_usage=True, uniform_shapes=True)
    host = [pin_batch(make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank + 100 * i)) for i in range(n_batches)]
    feeder = StaticBatchFeeder(host[0], device)
    trainer.prepare(feeder.static)
    feeder.capture()
    if not feeder._ready:
        feeder.prefetch(host[0])
    k = [0]

    def step():
        b = feeder.commit()
        loss_ = trainer.micro_step(b)                  # the step's graph is LAUNCHED before the host prepares the next batch:
        k[0] += 1                                      # prefetch() sorts the token ids on the host (~0.4 ms) - in front of the
        feeder.prefetch(host[k[0] % len(host)])        # launch that was 0.35 ms of idle GPU per step (round 5, first try)
        return loss_
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(SHAPES["D2"]["videos"] * steps / dt, 2), "unit": "videos/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "warmup": warmup,
            "input": "%d distinct pinned host batches through StaticBatchFeeder (33 MB of frame features per micro-step over "
                     "PCIe, copy stream one step ahead)" % n_batches}


def feed_ragged_run(device, rank, steps, warmup, n_batches=8, n_buckets=3):
    """Ragged TVR batches (every real batch has its own shape) through hero_amd.loader.BucketedBatchFeeder: distinct
    pinned host batches, each padded to one of <= n_buckets bucket shapes, H2D one step ahead, one pair of captured step graphs
    per bucket, the cross-modal layers packed through the feeder's static pack plan.  PCIe-inclusive; never the headline."""
    from hero_amd.loader import BucketedBatchFeeder, batch_dims, pin_batch
    from hero_amd.step import TrainStep
    from hero_amd.synth import SHAPES, make_batch
    cfg_path = "/tmp/hero_bench_feedr_%d.json" % rank
    with open(cfg_path, "w") as f:
        json.dump(HERO_BASE, f)
    model = build_model(device, cfg_path)
    trainer = TrainStep(model, use_graph=True, static_usage=True, uniform_shapes=True)
    raw = [make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank + 100 * i, ragged=True) for i in range(n_batches)]
    dims = [batch_dims(b) for b in raw]
    feeder = BucketedBatchFeeder(BucketedBatchFeeder.derive_buckets(dims, n_buckets=n_buckets), device)
    host = [pin_batch(feeder.pad(b)[1]) for b in raw]          # what a DataLoader worker + its pin thread hand over
    del raw
    k = [0]
    feeder.prefetch(host[0])

    def step():
        b = feeder.commit()
        loss_ = trainer.micro_step(b) if b is not None else trainer.micro_step(feeder.take_eager(), eager=True)
        k[0] += 1
        feeder.prefetch(host[k[0] % len(host)])
        return loss_
    for _ in range(max(warmup, 3 * len(host))):        # every bucket has met its first batch (eager) and a window start (capture)
        step()
    torch.cuda.synchronize()
    before = dict(trainer.counts)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(SHAPES["D2"]["videos"] * steps / dt, 2), "unit": "videos/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "warmup": max(warmup, 3 * len(host)),
            "buckets": [{"rows": b["rows"], "min_rows": b["min_rows"], "served": n} for b, n in zip(feeder.buckets, feeder.served)],
            "graphs": feeder.graphs, "packed_rows_of_the_batches": [d["rows"] for d in dims],
            "replayed_in_timed_region": trainer.counts["replayed"] - before["replayed"],
            "eager_in_timed_region": trainer.counts["eager_in_graph_mode"] - before["eager_in_graph_mode"],
            "input": "%d distinct ragged host batches (N_f ~ U{30..100}, 8-25 subtitles per video, 0-8 frames and 4-40 tokens per subtitle) "
                     "through BucketedBatchFeeder: bucket padding on the host, frame features over PCIe one step ahead, hipGraph replay "
                     "per bucket, packed cross-modal layers (static pack plan rebuilt on the host per batch)" % n_batches}


def live_pmc(timeout_s=75):
    """the HBM-traffic and MFMA-busy counters on the line used to be READ from a committed profile.  When
    rocprofv3 is on PATH, take them in THIS run: three short passes of this very command (2 eager micro-steps each, `--pmc`
    alone with --kernel-trace, separate passes as MI355X_MICROARCH.md prescribes) as subprocesses after the timed region,
    summarised by tools/profile_summary.py into a scratch directory.  Returns (traffic json, mfma json, note) or None - any
    failure (no rocprofv3, a refused counter, a timeout) leaves the line with the stamped committed profile."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_summary as PS
    d = tempfile.mkdtemp(prefix="hero_pmc_")
    import atexit
    atexit.register(shutil.rmtree, d, True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
           "--no-graph", "--no-box-probe", "--profile-steps", "0"]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    try:
        for tag, ctr in (("f", ["FETCH_SIZE"]), ("w", ["WRITE_SIZE"]), ("m", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"])):
            r = subprocess.run(["rocprofv3", "--pmc"] + ctr + ["--kernel-trace", "--output-format", "csv", "-d", os.path.join(d, tag), "-o", tag, "--"] + cmd,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            PS.pmc(os.path.join(d, "f"), os.path.join(d, "w"), os.path.join(d, "traffic.json"))
            PS.mfma(os.path.join(d, "m"), os.path.join(d, "mfma.json"))
        return (os.path.join(d, "traffic.json"), os.path.join(d, "mfma.json"),
                "measured in THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes of this command (2 eager "
                "micro-steps each, subprocesses after the timed region, %.0f s; FETCH x2 gfx950 correction)" % (time.perf_counter() - t0))
    except Exception:                                    # noqa: BLE001 - the committed profile stays the source
        return None


def _forget_previous_models():
    """Between the workloads of one process: the package caches compute copies of the weights per parameter object (and refreshes
    ALL of them after every optimiser step) and memoises tensors derived from batches - a later workload must not pay for, or
    keep alive, its predecessors' models and batches (round 4's secondary.feed ran with the copies of four dead models)."""
    import gc
    from hero_amd import functional as HF_
    HF_.reset_caches()
    gc.collect()
    torch.cuda.empty_cache()


def box_probe(device):
    """What THIS box delivers right now (the same tree ran 6.37-7.49 ms per step on different boxes of the
    pool): ~0.2 s of back-to-back bf16 MFMAs on every CU (hero_probe_mfma: dense TFLOP/s, sustained matrix clock) and a
    1 GiB streaming copy / read (hero_probe_hbm).  Runs before the timed region; the constants 2500 TFLOP/s / 8000 GB/s stay
    the `peak` the fractions are quoted against, the measured figures ride beside them."""
    from hero_amd import _lib as L
    n = 1 << 30
    a = torch.zeros(n // 4, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    tf, ghz, cp, rd = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    st = torch.cuda.current_stream(device).cuda_stream
    L.check(L.lib().hero_probe_mfma(b.data_ptr(), n, C.byref(tf), C.byref(ghz), st))
    L.check(L.lib().hero_probe_hbm(a.data_ptr(), b.data_ptr(), n, C.byref(cp), C.byref(rd), st))
    torch.cuda.synchronize()
    del a, b
    # Third probe: the product GEMM itself on one shape of the step (12000 x 3072 x 768, no epilogue).  The round's boxes with
    # EQUAL MFMA and HBM probes still ran the step 15 % apart (6.15 vs 7.26 ms): what differs between them is the path the
    # persistent GEMM loop is bound by (L2 -> LDS fill), which neither pure probe exercises.
    from hero_amd import functional as HF
    x = torch.randn(12000, 768, device=device).to(torch.bfloat16)
    w = (torch.randn(3072, 768, device=device) * 0.05).to(torch.bfloat16)
    for _ in range(60):
        HF.k_linear(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        HF.k_linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    gemm_us = e0.elapsed_time(e1) * 1000 / 40
    del x, w
    torch.cuda.empty_cache()
    return {"mfma_bf16_tflops": round(tf.value, 1), "mfma_clock_ghz": round(ghz.value, 3),
            "hbm_copy_gbps": round(cp.value, 1), "hbm_read_gbps": round(rd.value, 1),
            "gemm_12000x3072x768_us": round(gemm_us, 1), "gemm_12000x3072x768_tflops": round(2.0 * 12000 * 3072 * 768 / gemm_us / 1e6, 1),
            "how": "hero_probe_mfma: 36 independent v_mfma_f32_32x32x16_bf16 per iteration on 8 waves x every CU, past the clock ramp; "
                   "hero_probe_hbm: 1 GiB nontemporal streaming copy (read + written bytes) and read; gemm: 40 back-to-back hero_gemm launches "
                   "(eager, HIP events)"}


def launch_command(n, argv, n_devices, port=None):
    """The command (and extra environment) `python bench.py --gpus N` turns itself into when N > 1 and no launcher has
    set WORLD_SIZE: one process per GPU under torch.distributed.run on 127.0.0.1, exactly what the driver's own N > 1
    command line is.  On a box with fewer than N GPUs the ranks share device 0 and talk over gloo (RCCL refuses two
    ranks on one device): a PLUMBING run - process group, broadcast, bucketed exchange, negatives, max-over-ranks
    timing, the one JSON line - that says so in `config.parallelism`."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = {"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    if n_devices < n:
        env["HERO_BENCH_ONE_DEVICE"] = "1"
        env["HERO_BENCH_BACKEND"] = os.environ.get("HERO_BENCH_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return cmd, env


def self_launch(args):
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no GPU visible (the hot path has no CPU fallback)")
    cmd, env = launch_command(args.gpus, sys.argv[1:], n_dev)
    sys.stdout.flush()
    os.execve(cmd[0], cmd, dict(os.environ, **env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed micro-steps (default 200 = a 1.2 s timed region; D4: 20 = 11 s)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed micro-steps before them (default 10; D4: 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip the short D2r / D3 / feed runs attached to the D2 line as `secondary`")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the ~0.5 s MFMA / HBM probes of the box")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do not take the three short rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE, MFMA-busy) "
                         "after the timed region; `roofline.traffic / hbm_gbps / mfma_busy` then come from the committed profile "
                         "(stamped with the kernel-source hash).  Implied by --no-secondary (what the profiling scripts pass).")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (N=1)")
    ap.add_argument("--workload", default="D2", choices=["D2", "D2r", "D3", "D4"],
                    help="D2 (default, the headline line): BASELINE configs[1]/[2]; secondary lines: D2r = ragged TVR "
                         "batch (SURVEY 8d), D3 = configs[3] multi-task pre-training, D4 = configs[4] 256-frame videos "
                         "with the batch sized to HBM")
    ap.add_argument("--videos", type=int, default=0, help="D4: videos per step (0 = fill ~80 %% of HBM)")
    ap.add_argument("--feed", type=int, default=0,
                    help="D2: rotate this many DISTINCT pinned host batches through hero_amd.loader.StaticBatchFeeder (H2D of "
                         "the frame features on a copy stream one step ahead, index tensors rebuilt on the device) instead of "
                         "re-running one HBM-resident batch; the headline value is the resident one (0)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "abi"],
                    help="N > 1: what carries the device collectives - torch.distributed's process group (default, DESIGN 6) or "
                         "hero_comm_* of the C ABI (RCCL on a side stream of this process)")
    args = ap.parse_args()
    if args.steps is None:                           # 20 steps were a 0.125 s timed region
        args.steps = 20 if args.workload == "D4" else 200
    if args.warmup is None:
        args.warmup = 4 if args.workload == "D4" else 10

    self_launch(args)                                # plain `python bench.py --gpus N`, N > 1: becomes the launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    one_device = bool(os.environ.get("HERO_BENCH_ONE_DEVICE"))      # plumbing run of the N > 1 path on a box with fewer GPUs
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or (bool(os.environ.get("HERO_DP_FORCE_COLLECTIVES")) and "RANK" in os.environ)
    if dist_on:
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
    if dist_on and args.exchange == "abi":
        from hero_amd.utils import distributed as D_
        D_.set_exchange("abi")                       # collective: the RCCL communicator of hero_comm_* is created here
    if args.workload != "D2":
        out = secondary_workload(args, device, world, rank)
        if rank == 0:
            print(json.dumps(out))
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    box = None
    if not args.no_box_probe:
        try:
            box = box_probe(device)                  # every rank probes its own GPU (rank 0's goes on the line)
        except Exception as e:                       # noqa: BLE001 - a probe never takes the headline down
            box = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    cfg_path = "/tmp/hero_finetune_bench_%d.json" % rank
    with open(cfg_path, "w") as f:
        json.dump(HERO_BASE, f)
    model = build_model(device, cfg_path)
    # drop_svmr_prob = 0: every step uses the same parameters; every rank's batch has the same shape.
    # N = 1: the step is captured in hipGraphs and replayed.  N > 1: the eager run (RCCL all-reduces issued from the
    # backward hooks) is timed FIRST - it is the mode every multi-rank test covers - and then, unless HERO_DP_GRAPH=0,
    # the step is captured WITH its collectives and timed again under a watchdog; the line reports the faster of the
    # two measured runs and says which (`config.launch`).  Each run is exactly --steps micro-steps between barriers.
    trainer = TrainStep(model, use_graph=(not dist_on and not args.no_graph), static_usage=True, uniform_shapes=True)
    feeder, host_batches = None, []
    if args.feed > 0:
        from hero_amd.loader import StaticBatchFeeder, pin_batch
        host_batches = [pin_batch(make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank + 100 * i)) for i in range(args.feed)]
        feeder = StaticBatchFeeder(host_batches[0], device)
        batch = feeder.static
    else:
        batch = make_batch("D2", vfeat_dim=VFEAT, vocab=50272, seed=1 + rank, device=device)
    sh = SHAPES["D2"]
    fed = [0]

    def step():
        if feeder is None:
            return trainer.micro_step(batch)
        b = feeder.commit()                                              # the batch prefetched during the previous step
        loss_ = trainer.micro_step(b)                                    # launch first, then the host-side work of the next batch
        fed[0] += 1
        feeder.prefetch(host_batches[fed[0] % len(host_batches)])        # next one: copy stream, overlaps this step
        return loss_

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def timed_run():
        trainer.prepare(batch)                       # hipGraph capture is setup, never inside the timed region
        if feeder is not None:
            feeder.capture()
            if not feeder._ready:
                feeder.prefetch(host_batches[0])
        for _ in range(args.warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_ = step()
        sync()
        dt_ = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt_], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, float(loss_)

    dt, loss_val = timed_run()
    graph_mode = trainer.use_graph
    # (the captured run needs device-side collectives: over gloo - the plumbing run - they are host-staged and cannot be captured)
    try_graph = (dist_on and not args.no_graph and os.environ.get("HERO_DP_GRAPH", "1") not in ("", "0")
                 and torch.distributed.get_backend() == "nccl")
    eager_ms = dt / args.steps * 1e3

    # ---- N > 1: what the collectives cost, so that the line itself shows that RCCL saw N ranks ---------------------
    comm = None
    if dist_on:
        accum = trainer.opts.gradient_accumulation_steps
        ones = torch.ones(1, device=device)
        torch.distributed.all_reduce(ones)                                 # every rank contributes a 1
        ar_ms = trainer.arena.probe_allreduce_ms(reps=3)                   # the bucketed all-reduce of one optimiser step, alone
        while trainer.micro % accum:
            step()
        trainer.arena.mute = True                                          # same K steps without the gradient exchange
        try:
            dt_mute, _ = timed_run()
        finally:
            trainer.arena.mute = False
        from hero_amd.utils import distributed as D
        D.broadcast_tensors([p.data for p in model.parameters()], 0)       # the muted steps let the replicas drift: re-align
        HF_ = __import__("hero_amd.functional", fromlist=["x"])
        HF_.notify_weights_updated()
        HF_.refresh_weight_cache()
        mute_ms = dt_mute / args.steps * 1e3
        comm = {"backend": torch.distributed.get_backend() + (" (RCCL)" if torch.distributed.get_backend() == "nccl" else ""),
                "exchange": "hero_comm_* (C ABI, side stream)" if trainer.arena.backend == "abi" else "torch.distributed process group",
                "ranks_seen": int(ones.item()), "wire_dtype": trainer.arena.compress or "f32",
                "buckets": len(trainer.arena.buckets), "payload_mb_per_opt_step": round(trainer.arena.wire_bytes() / 2 ** 20, 1),
                "allreduce_ms_per_opt_step": round(ar_ms, 3),
                "eager_ms_per_step": round(eager_ms, 3), "eager_ms_per_step_without_grad_exchange": round(mute_ms, 3),
                "exposed_ms_per_opt_step": round(max(0.0, eager_ms - mute_ms) * accum, 3),
                "forward_allgather": "negatives of the VSM loss (3 padded all-gathers per micro-step, in both runs)",
                "graph_ms_per_step": None}

    # ---- roofline leg: HIP events around every GEMM launch over extra, identical steps ----------
    roof = None
    if rank == 0:
        L.check(L.lib().hero_prof_enable(1))
        trainer.use_graph = False                    # events cannot be recorded inside a replay
        for _ in range(args.profile_steps):
            trainer.micro_step(batch)
        torch.cuda.synchronize()
        best = None
        for slot in range(11):
            ms, fl, n = C.c_double(), C.c_double(), C.c_longlong()
            L.check(L.lib().hero_prof_read(slot, C.byref(ms), C.byref(fl), C.byref(n)))
            if n.value and (best is None or ms.value > best[1]):
                best = (slot, ms.value, fl.value, n.value)
        L.check(L.lib().hero_prof_enable(0))
        if best:
            slot, ms, fl, n = best
            ach = fl / (ms * 1e-3) / 1e12
            peak = BF16_PEAK_TFLOPS if slot >= 4 else 157.3      # slots 0-3 are the fp32 parity-mode kernels
            traffic, tsrc = None, None
            live = live_pmc() if (world == 1 and not dist_on and not args.no_live_pmc and not args.no_secondary and not args.feed) else None
            want = {4: "gemm_glds_kernel<unsigned short", 5: "gemm_kernel<unsigned short, 0, 1",
                    7: "gemm_glds_tr_kernel", 8: "gemm_ws_kernel<3, 3, false", 9: "gemm_ws",
                    10: "gemm_ws_kernel<2, 3, false"}.get(slot)
            try:                                   # HBM bytes per launch from the committed PMC passes, stamped with the
                pj = live[0] if live else os.path.join(ROOT, "profiles", PROFILE_TRAFFIC)     # kernel sources they were taken with
                pm = json.load(open(pj))
                meta = pm.pop("_meta", {})
                hits = [v for k, v in pm.items() if want and want in k and (slot != 9 or ", false," not in k)]
                if hits:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    from profile_summary import csrc_sha16
                    same = meta.get("csrc_sha16") == csrc_sha16()
                    tot = sum(h["launches"] for h in hits)
                    traffic = sum(h["hbm_bytes_per_launch_corrected"] * h["launches"] for h in hits) / tot
                    tsrc = live[2] if live else (
                        "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH x2 gfx950 correction), "
                        "taken with kernel sources %s = %s" % (PROFILE_TRAFFIC, meta.get("csrc_sha16"),
                                                              "the sources of this build" if same else "NOT this build's sources (stale)"))
            except Exception:
                pass
            # north_star: "rocprof-reported HBM GB/s and MFMA utilisation against chip peak" - counters cannot be read
            # inside the run, so both come from the committed PMC passes of the same command (stamped like `traffic`):
            # HBM bytes per launch / this run's live launch duration, and SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x
            # GPU-active cycles), launch-weighted over the family's instantiations.
            hbm_gbps = mfma_busy = None
            if traffic:
                hbm_gbps = traffic / (ms * 1e-3 / n) / 1e9
            try:
                mm = json.load(open(live[1] if live else os.path.join(ROOT, "profiles", PROFILE_MFMA)))
                mm.pop("_meta", None)
                mh = [v for k, v in mm.items() if want and want in k and (slot != 9 or ", false," not in k)]
                if mh:
                    mfma_busy = sum(h["mfma_util"] * h["launches"] for h in mh) / sum(h["launches"] for h in mh)
            except Exception:
                pass
            roof = {"bound": "mfma", "kernel": SLOT_NAMES.get(slot, "gemm slot %d" % slot),
                    "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc, "launches": n,
                    "avg_launch_us": round(ms * 1e3 / n, 2),
                    "flops_per_launch": fl / n,
                    "hbm_gbps": round(hbm_gbps, 1) if hbm_gbps else None,
                    "hbm_frac_of_peak": round(hbm_gbps / HBM_PEAK_GBPS, 4) if hbm_gbps else None,
                    "mfma_busy": round(mfma_busy, 4) if mfma_busy is not None else None,
                    "peak_measured": box.get("mfma_bf16_tflops") if box and slot >= 4 else None,
                    "frac_of_peak_measured": round(ach / box["mfma_bf16_tflops"], 4) if box and box.get("mfma_bf16_tflops") and slot >= 4 else None,
                    "hbm_peak_measured": box.get("hbm_copy_gbps") if box else None,
                    "clock_ghz": box.get("mfma_clock_ghz") if box else None,
                    "counters_source": live[2] if live else
                    "profiles/%s + profiles/%s (rocprofv3 --pmc passes of this command, committed)" % (PROFILE_TRAFFIC, PROFILE_MFMA)}
    elif world > 1:
        for _ in range(args.profile_steps):
            trainer.micro_step(batch)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, HERO_BASE)

    def line(dt_, loss_, launch_mode):
        vps = sh["videos"] * world * args.steps / dt_
        fl_video = algorithmic_flops_per_video(sh)
        return {
            "metric": "videos/sec training step, HERO-base TVR-shaped batch",
            "value": round(vps, 2), "unit": "videos/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_ / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: HERO-base TVR finetune micro-step (train-tvr-8gpu.json shapes: "
                                   "32 videos x 60 frames, 15 subs x (4 frames + 20 tokens), 15-token query; "
                                   "vfeat 4352; fwd + VSM loss + bwd, all-reduce/clip/AdamW every 2nd micro-step)",
                       "global_batch": sh["videos"] * world,
                       "parallelism": "dp%d" % world + (" (PLUMBING run: the %d ranks share ONE device, %s transport - not a scaling "
                                                        "number)" % (world, torch.distributed.get_backend()) if one_device and world > 1 else ""),
                       "dropout": 0.1, "grad_accum": 2, "launch": launch_mode,
                       "input": ("%d distinct pinned host batches rotated through StaticBatchFeeder (33 MB of frame features per "
                                 "micro-step over PCIe on a copy stream, one step ahead; index tensors rebuilt on the device)" % args.feed)
                       if args.feed else "one batch resident in HBM"},
            "step_tflops": round(vps * fl_video / 1e12, 1),
            "step_frac_of_bf16_peak": round(vps * fl_video / 1e12 / world / BF16_PEAK_TFLOPS, 4),
            "step_frac_of_peak_measured": round(vps * fl_video / 1e12 / world / box["mfma_bf16_tflops"], 4) if box and box.get("mfma_bf16_tflops") else None,
            "final_loss": loss_,
            "box": box,
            "roofline": roof,
            "cpu_baseline": cpu,
            "comm": comm,
        }

    out = line(dt, loss_val, "hipGraph replay" if graph_mode else "eager")
    hard_exit = False
    if try_graph:
        # Second measured run of the same K steps: the step captured WITH its RCCL collectives.  A watchdog prints the
        # eager line and leaves if the captured run does not come back (a wedged collective cannot be recovered from
        # inside the process); whichever run was faster is reported, the other one's time goes into `config`.
        import threading
        finished = threading.Event()

        def bail():
            if not finished.is_set():
                if rank == 0:
                    out["config"]["launch"] += " (the hipGraph run with captured RCCL collectives timed out)"
                    print(json.dumps(out), flush=True)
                os._exit(0)
        timer = threading.Timer(float(os.environ.get("HERO_DP_GRAPH_TIMEOUT", "240")), bail)
        timer.daemon = True
        timer.start()
        try:
            while trainer.micro % trainer.opts.gradient_accumulation_steps:     # odd --steps / --warmup: finish the window
                step()
            trainer.enable_graph(collectives=True)
            if os.environ.get("HERO_DP_GRAPH_TEST_WEDGE"):      # test hook: a captured run that never comes back
                time.sleep(1e6)
            dt_g, loss_g = timed_run()
            if comm is not None:
                comm["graph_ms_per_step"] = round(dt_g / args.steps * 1e3, 3)
            if dt_g < dt and loss_g == loss_g:
                out = line(dt_g, loss_g, "hipGraph replay (step captured with its RCCL collectives)")
                out["config"]["eager_ms_per_step"] = round(eager_ms, 3)
            else:
                out["config"]["graph_ms_per_step"] = round(dt_g / args.steps * 1e3, 3)
        except Exception as e:                       # noqa: BLE001 - any capture failure: the eager measurement stands
            out["config"]["launch"] += " (hipGraph capture with RCCL collectives failed: %s)" % type(e).__name__
            hard_exit = True                         # the process group may be unusable: no orderly teardown
        finished.set()
        timer.cancel()
    # ---- N = 1: the other configurations ride on the line (short runs, after everything that is timed above) -------
    if rank == 0 and world == 1 and not dist_on and not args.no_secondary and not args.feed:
        del trainer, model, batch
        _forget_previous_models()
        sec, t_sec = {}, time.perf_counter()
        ns = argparse.Namespace(**vars(args))
        for w in ("D2r", "D3", "D4"):
            try:
                ns.workload = w
                if w == "D4":                        # configs[4] at a BOUNDED size (256 videos x 256 frames: ~70 GB, ~0.16 s per
                    ns.videos = 256                  # step); the HBM-filling run (979 videos, 90 % of HBM) is `--workload D4`
                    r = secondary_workload(ns, device, world, rank, steps=2, warmup=2)
                else:
                    r = secondary_workload(ns, device, world, rank, steps=20, warmup=4)
                sec[w] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "per_task", "peak_mem_gb",
                                            "step_frac_of_bf16_peak") if k in r}
                if w == "D4":
                    sec[w]["videos"] = ns.videos
                sec[w]["launch"] = r["config"]["launch"]
            except Exception as e:                       # noqa: BLE001 - a secondary line never takes the headline down
                sec[w] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            _forget_previous_models()
        try:
            sec["feed"] = feed_run(device, rank, steps=min(args.steps, 100), warmup=6)
        except Exception as e:                           # noqa: BLE001
            sec["feed"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        _forget_previous_models()
        try:
            sec["feed_ragged"] = feed_ragged_run(device, rank, steps=min(args.steps, 60), warmup=6)
        except Exception as e:                           # noqa: BLE001
            sec["feed_ragged"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        sec["wall_s"] = round(time.perf_counter() - t_sec, 1)
        out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if hard_exit:
        os._exit(0)
    if dist_on:
        torch.distributed.destroy_process_group()

if __name__ == "__main__":
    main()
