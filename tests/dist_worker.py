"""Worker of tests/test_gpu_distributed.py: one rank of a data-parallel run with the REAL HIP kernels.
The GPU box has one MI355X, so either (a) two ranks share cuda:0 and the collectives go through gloo
(host-staged) - everything above the transport (GradArena's use/done counting from the HIP backward,
bucket launches, averaging folded into the optimiser, cross-rank negatives) is the production code path
of bench.py --gpus N - or (b) ONE rank runs over the real RCCL backend with HERO_DP_FORCE_COLLECTIVES=1:
process-group setup on the device, asynchronous all-reduces on RCCL's stream while backward continues,
the bf16 wire buffers, wait / barrier / teardown."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    import hero_amd
    if len(sys.argv) > 4 and sys.argv[4] == "abi":     # gradient buckets / broadcast / negatives through hero_comm_* (C ABI)
        from hero_amd.utils import distributed as D_
        D_.set_exchange("abi")
    from hero_amd.step import TrainStep
    from hero_amd.synth import make_batch
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    from tests.util import load_tiny
    model, _, _ = load_tiny("cuda")
    if rank == 1:                                      # the constructor's broadcast from rank 0 must undo this
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    set_dropout(model, 0.0)
    model.train()
    wire = sys.argv[2] if len(sys.argv) > 2 else "none"          # gradient wire format: fp32 (exact check) or bf16
    tol = 1e-5 if wire == "none" else 2.0 ** -6
    trainer = TrainStep(model, opts={"gradient_accumulation_steps": 2, "learning_rate": 1e-3}, use_graph=False,
                        bucket_bytes=64 << 10,        # small buckets -> many overlapped all-reduces
                        grad_compress=None if wire == "none" else wire)
    named = dict(model.named_parameters())
    if len(sys.argv) > 4 and sys.argv[4] == "feeder":
        return feeder_run(trainer, model, named, rank, world, dev, out_dir)
    if len(sys.argv) > 4 and sys.argv[4] == "multiq":
        return multiq_run(trainer, model, named, rank, world, dev, out_dir)
    # (1) parameters were broadcast from rank 0
    chk = torch.stack([p.detach().double().sum() for p in named.values()])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(both[0], b) for b in both), "parameter broadcast failed"

    batch = make_batch("D1", vfeat_dim=96, vocab=160, seed=1 + rank, device=dev)
    errs = []
    for fuse in (True, False):       # False: the cross-modal layers run twice per forward (two uses per parameter)
        model.fuse_query_pass = fuse
        # (2) local gradients of two accumulated micro-steps WITHOUT synchronisation
        trainer.arena.zero()
        for _ in range(2):
            trainer.arena.set_sync(False)
            trainer._fwd_bwd(batch)
            trainer.arena.finish()
        local = trainer.arena.flat.clone()
        gl = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gl, local)
        want_sum = sum(gl)
        # (3) the same two micro-steps through the production path: buckets all-reduced from the hooks
        trainer.arena.zero()
        trainer.arena.set_sync(False)
        trainer._fwd_bwd(batch)
        trainer.arena.set_sync(True)
        trainer._fwd_bwd(batch)
        trainer.arena.finish()
        got = trainer.arena.flat.clone()
        err = (got - want_sum).abs().max().item() / max(want_sum.abs().max().item(), 1e-12)
        if err >= tol:
            bad = []
            for n_, p_ in named.items():
                s_, e_ = trainer.arena.slices[p_]
                d_ = (got[s_:e_] - want_sum[s_:e_]).abs().max().item()
                if d_ > 1e-6 * max(want_sum.abs().max().item(), 1e-12):
                    bad.append((round(d_ / max(want_sum[s_:e_].abs().max().item(), 1e-12), 4), n_, trainer.arena.bucket_of[p_]))
            print("RANK", rank, "mismatching parameters:", sorted(bad, reverse=True)[:12], flush=True)
        assert err < tol, "bucketed all-reduce != sum of local gradients (rel %g, wire %s)" % (err, wire)
        errs.append(err)
    err = max(errs)
    nb = len(trainer.arena.buckets)
    # (4) full optimiser steps keep the replicas identical
    trainer.arena.zero()
    losses = []
    for _ in range(4):
        losses.append(float(trainer.micro_step(batch)))
    chk = torch.stack([p.detach().double().sum() for p in named.values()])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(both[0], b) for b in both), "replicas diverged after optimiser steps"
    assert all(l == l and abs(l) < 1e4 for l in losses)
    torch.cuda.synchronize()
    json.dump({"rank": rank, "buckets": nb, "rel_err": err, "losses": losses, "backend": dist.get_backend(), "exchange": trainer.arena.backend,
               "collectives": bool(__import__("hero_amd.utils.distributed", fromlist=["x"]).collectives_active())},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.destroy_process_group()


def multiq_run(trainer, model, named, rank, world, dev, out_dir):
    """Round 6: VSM batches with several queries per video (data/vsm.py:105-145) under data parallelism - the fused HIP head
    on the (query, video) pairs with cross-rank negatives (model/pretrain.py:383-401, 427-451) == the PyTorch head: the
    three (global) losses and, after the bucketed exchange, every gradient."""
    from hero_amd.synth import make_pretrain_batches
    batch = make_pretrain_batches("D1", vfeat_dim=96, vocab=160, seed=1 + rank, device=dev, queries_per_video=3)["vsm"]
    nq, nv = batch["query_input_ids"].shape[0], batch["c_attn_masks"].shape[0]
    assert nq == 3 * nv
    res = []
    for fused in (True, False):
        model.fused_head = fused
        model.q_feat_attn.fused_pool = fused
        trainer.arena.zero()
        trainer.arena.set_sync(True)
        loss = trainer._fwd_bwd(batch)
        trainer.arena.finish()
        torch.cuda.synchronize()
        res.append((float(loss), trainer.arena.flat.clone()))
    (lf, gf), (lt, gt) = res
    assert abs(lf - lt) < 1e-5 * max(1.0, abs(lt)), (lf, lt)
    err = (gf - gt).abs().max().item() / max(gt.abs().max().item(), 1e-12)
    assert err < 2e-4, "fused several-queries head != PyTorch head under cross-rank negatives (rel %g)" % err
    both = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, torch.tensor([lf], device=dev, dtype=torch.float64))
    json.dump({"rank": rank, "loss": lf, "losses_all": [float(b) for b in both], "rel_err": err, "nq": nq, "nv": nv},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.destroy_process_group()


def feeder_run(trainer, model, named, rank, world, dev, out_dir):
    """Round 6: two ranks, each feeding its OWN ragged batches through a BucketedBatchFeeder (eager launches: the mode every
    multi-rank test covers).  The buckets are derived from both ranks' batches (every padded dimension shared, the packed row
    capacities differ), so at any step the ranks may sit on different buckets - different GEMM row counts, different pack
    plans - while exchanging the same gradient buckets and the same padded negatives; the replicas must stay identical."""
    from hero_amd.loader import BucketedBatchFeeder, batch_dims
    from tests.test_gpu_loader import _ragged_batches
    host = _ragged_batches(4, seed0=40 + 10 * rank)
    dims = [batch_dims(h) for h in host]
    every = [None] * world
    dist.all_gather_object(every, dims)
    buckets = BucketedBatchFeeder.derive_buckets([d for ds in every for d in ds], n_buckets=3, row_quantum=16)
    feeder = BucketedBatchFeeder(buckets, dev)
    used, losses = [], []
    feeder.prefetch(host[0])
    for n in range(8):
        b = feeder.commit()
        assert b is not None and b.get("_static_plan")
        used.append(int(b["_bucket"]))
        losses.append(float(trainer.micro_step(b)))
        feeder.prefetch(host[(n + 1) % len(host)])
    chk = torch.stack([p.detach().double().sum() for p in named.values()])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(both[0], b_) for b_ in both), "replicas diverged after optimiser steps on bucketed ragged batches"
    assert all(l == l and abs(l) < 1e4 for l in losses)
    torch.cuda.synchronize()
    json.dump({"rank": rank, "buckets_used": used, "n_buckets": len(buckets), "losses": losses, "backend": dist.get_backend()},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
