import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0); dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.step import TrainStep
    from hero_amd.synth import make_batch
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    from tests.util import load_tiny
    model, _, _ = load_tiny("cuda")
    set_dropout(model, 0.0); model.train()
    trainer = TrainStep(model, opts={"gradient_accumulation_steps": 2, "learning_rate": 1e-3}, use_graph=False, bucket_bytes=64 << 10, grad_compress=None)
    names = {p: n for n, p in model.named_parameters()}
    arena = trainer.arena
    log = []
    orig_launch = arena._launch
    def dbg_launch(b):
        import traceback
        log.append(("LAUNCH", b, arena._launched[b], list(arena._pending[:4]), [f.name for f in traceback.extract_stack()[-6:-1]])); return orig_launch(b)
    arena._launch = dbg_launch
    orig_done = arena.done
    def dbg_done(p):
        log.append(("DONE", names[p].split("encoder.")[-1], arena.bucket_of.get(p), arena._uses.get(p)))
        return orig_done(p)
    arena.done = dbg_done
    orig_flush = HF.wgrad_flush
    def dbg_flush():
        log.append(("FLUSH", len(HF._WQ), [tuple(e[2].shape) for e in HF._WQ]))
        return orig_flush()
    HF.wgrad_flush = dbg_flush
    batch = make_batch("D1", vfeat_dim=96, vocab=160, seed=1 + rank, device=dev)
    model.fuse_query_pass = True
    arena.zero()
    log.append(("PENDING0", list(arena._pending[:6]), [b[2] for b in arena.buckets[:6]]))
    arena.set_sync(False); trainer._fwd_bwd(batch)
    log.append(("PENDING1", list(arena._pending[:6])))
    log.append(("---- micro 2",))
    arena.set_sync(True); trainer._fwd_bwd(batch)
    log.append(("---- finish",))
    arena.finish()
    if rank == 0:
        for e in log: print(e)
    dist.destroy_process_group()
main()
