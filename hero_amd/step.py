"""Training micro-step harness reproducing the reference loop (train_vcmr.py:202-262):

  loss = loss_st_ed + loss_neg_ctx + loss_neg_q -> mean -> backward            every micro-step
  on every `gradient_accumulation_steps`-th micro-step:
      gradient all-reduce (average over ranks)  ->  lr = lr0 * warmup_linear(step)  ->
      clip_grad_norm_(1.0)  ->  AdamW.step()  ->  zero_grad()

MI355X-first differences: no host synchronisation inside the step (the reference calls .item() four
times per micro-step), gradients live in one flat arena that is all-reduced in buckets while
backward is still running, clipping + averaging + AdamW are one kernel pass, and — for static
shapes on one GPU — the whole micro-step (forward, loss, backward, optimiser) is captured once in
a hipGraph and replayed, which removes the ~900 host-side launches per step.
"""
import os
from types import SimpleNamespace

import torch

from . import functional as HF
from .optim import build_optimizer, get_lr_sched
from .utils import distributed as D

TVR_OPTS = dict(  # config/train-tvr-8gpu.json
    learning_rate=1e-4, lr_mul=1.0, weight_decay=0.01, optim="adamw", betas=[0.9, 0.98],
    grad_norm=1.0, warmup_steps=500, num_train_steps=5000, gradient_accumulation_steps=2,
    train_batch_size=32, dropout=0.1, lw_neg_q=8.0, lw_neg_ctx=8.0, lw_st_ed=0.01, margin=0.1)


def qkv_groups(model):
    """Gradient-arena groups: each attention block's query/key/value weights (and biases) back to
    back, so that their gradients are one [3D, D] GEMM (+ one column sum) in the backward."""
    from .model.layers import BertSelfAttention
    groups = []
    for m in model.modules():
        if isinstance(m, BertSelfAttention):
            groups.append((m.query.weight, m.key.weight, m.value.weight))
            groups.append((m.query.bias, m.key.bias, m.value.bias))
    return groups


class _packing:
    """Restrict the packed (ragged) formulation of BertEncoder for the duration: "all" = no restriction, "static" = only
    plans a feeder registered with fixed buffer sizes (BertEncoder.register_static_plan), "none" = padded formulation."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from .model.layers import BertEncoder
        self.prev, BertEncoder.packing_mode = BertEncoder.packing_mode, self.mode

    def __exit__(self, *exc):
        from .model.layers import BertEncoder
        BertEncoder.packing_mode = self.prev


def _packing_mode_of(batch):
    """A batch whose BUFFERS are rewritten between steps (hero_amd.loader.StaticBatchFeeder, DeviceCollate batches fed to a
    captured step) must not run a pack plan derived from its masks on the host: the packed row COUNT would be the capture
    batch's on every later batch.  It runs padded - or, when its feeder owns a static plan (fixed row capacity,
    refreshed with every batch: `_static_plan`), packed through that plan.  Enforced for eager steps on such batches too, so
    both modes compute the same thing."""
    if not hasattr(batch, "get") or not batch.get("_static_buffers"):
        return "all"
    return "static" if batch.get("_static_plan") else "none"


class TrainStep:
    def __init__(self, model, opts=None, task="tvr", bucket_bytes=64 << 20, use_graph=False, static_usage=False,
                 grad_compress="bf16", uniform_shapes=False, graph_collectives=None):
        """static_usage: the set of parameters that receive gradients only grows over the run (single-task
        fine-tuning with drop_svmr_prob = 0, as bench.py runs it) - lets the gradient buckets that also
        hold never-used parameters overlap with backward too.  Off by default: the reference configs
        drop the st/ed head on 80 % of the steps (config/train-tvr-8gpu.json:30).
        grad_compress: 'bf16' (default) sends the gradient buckets as bf16 - the reference's payload is fp16
        (train-tvr-8gpu.json:67 "fp16": true) - None keeps fp32 on the wire.
        uniform_shapes: every rank feeds batches of the same padded shape - the cross-rank negatives skip their
        per-forward host-side size exchange (utils/distributed.gather_negatives).
        graph_collectives: with use_graph in a data-parallel run, capture the step INCLUDING the bucketed RCCL
        all-reduces and the forward all-gathers (they are issued on streams that join the capture); default: the
        HERO_DP_GRAPH environment switch, off - N > 1 then runs eagerly, the mode every multi-rank test covers."""
        self.model = model
        self.opts = SimpleNamespace(**{**TVR_OPTS, **(opts or {})})
        self.task = task
        self.optimizer = build_optimizer(model, self.opts)
        self.arena = D.GradArena(list(model.parameters()), bucket_bytes=bucket_bytes,
                                 groups=qkv_groups(model), static_usage=static_usage,
                                 compress=grad_compress if D.collectives_active() else None)
        self.micro = 0
        self.global_step = 0
        D.broadcast_tensors([p.data for p in model.parameters()], 0)     # train_vcmr.py:152
        self.uniform_shapes = bool(uniform_shapes)      # applied around each forward (_fwd_bwd): not a process-wide setting
        if graph_collectives is None:
            graph_collectives = os.environ.get("HERO_DP_GRAPH", "0") not in ("", "0")
        self.use_graph = use_graph and (not D.collectives_active() or (graph_collectives and uniform_shapes))
        self.counts = {"replayed": 0, "eager_in_graph_mode": 0, "captures": 0}
        self._graphs = {}             # task, or (task, bucket) for a feeder's bucket batches -> (plain graph, its loss, boundary graph, its loss, static batch)
        self._window = []             # tasks of the micro-steps accumulated since the last optimiser step
        # compute copies of the weights are cached per optimiser step: anything else that rewrites parameters
        # (checkpoint restore, EMA swap) must invalidate them
        model.register_load_state_dict_post_hook(lambda *_: HF.notify_weights_updated())
        dev = next(model.parameters()).device
        self._step_t = torch.zeros(1, dtype=torch.int32, device=dev)      # device-side optimiser step
        self._lr_t = torch.zeros(8, dtype=torch.float32, device=dev)

    def enable_graph(self, collectives=False):
        """Switch an eager trainer to hipGraph replay (the next micro_step / prepare captures).  collectives: allow
        it in a data-parallel run - needs uniform_shapes (see __init__)."""
        if D.collectives_active() and not (collectives and self.uniform_shapes):
            raise RuntimeError("graph replay of a data-parallel step needs collectives=True and uniform_shapes=True")
        if self.micro % self.opts.gradient_accumulation_steps != 0:
            raise RuntimeError("switch to graph replay on an accumulation boundary")
        self.use_graph = True

    # ---- pieces ------------------------------------------------------------------------------------
    def _fwd_bwd(self, batch, task=None):
        HF.advance_seed()
        # dropout sites restart with every step: a site is then the SAME number in the eager step and in its captured
        # replay (a capture freezes the site numbers it saw; counting on across steps made the two modes draw different
        # masks), the per-step seed word keeps the steps apart - graph replay and eager runs follow one trajectory
        HF.RNG.site = 0
        with _packing(_packing_mode_of(batch)), HF.weights_frozen(), D.uniform_shapes(self.uniform_shapes):
            out = self.model(batch, task=task or self.task, compute_loss=True)
            # 'tvr' / 'vsm': (loss_st_ed, loss_neg_ctx, loss_neg_q) summed (train_vcmr.py:216-226, pretrain.py:283-290);
            # 'mlm' / 'mfm-nce' / 'fom': one loss tensor
            loss = (out[0] + out[1] + out[2]) if isinstance(out, (tuple, list)) else out
            if loss.dim() > 0:                       # train_vcmr.py:226 `.mean()` of per-GPU losses; a 0-dim loss is its own mean
                loss = loss.mean()
            loss.backward()
        return loss.detach()

    def _optimise(self, device_state):
        self.arena.finish()
        untouched = [p for p in self.arena.params if p not in self.arena.touched]
        sumsq = self.optimizer.grad_sumsq(self.arena.flat) if self.opts.grad_norm != -1 else None
        if device_state:
            self._step_t.add_(1)
        self.optimizer.step(grad_sumsq=sumsq, max_grad_norm=float(self.opts.grad_norm),
                            grad_scale=1.0 / D.world_size(), skip=set(untouched),
                            step_tensor=self._step_t if device_state else None,
                            lr_tensor=self._lr_t if device_state else None)
        HF.refresh_weight_cache()                    # every compute copy of the weights in one launch
        self.arena.zero()

    def _set_lr(self):
        self.global_step += 1
        lr = get_lr_sched(self.global_step, self.opts)
        for g in self.optimizer.param_groups:
            g["lr"] = lr                               # train_vcmr.py:248-249 (all groups)
        return lr

    # ---- eager -------------------------------------------------------------------------------------
    def micro_step(self, batch, task=None, eager=False):
        """One forward+backward; optimiser step on accumulation boundaries. Returns the loss
        tensor (device-resident, not synchronised).  task: this micro-step's task (multi-task
        pre-training, pretrain.py:274-350); default: the task given at construction.
        eager: in graph mode, run THIS micro-step with eager launches (a batch no captured graph fits - e.g. one that none
        of a BucketedBatchFeeder's buckets held); it shares the accumulation window and the device-side optimiser state
        with the replayed ones."""
        task = task or self.task
        if self.use_graph:
            if eager:
                return self._eager_step_in_graph_mode(batch, task)
            return self._graph_step(batch, task)
        accum = self.opts.gradient_accumulation_steps
        boundary = (self.micro + 1) % accum == 0
        self.arena.set_sync(boundary)
        loss = self._fwd_bwd(batch, task)
        self.micro += 1
        if boundary:
            self._set_lr()
            self._optimise(device_state=False)
        return loss

    # ---- hipGraph ------------------------------------------------------------------------------------
    def _capture(self, batch, task):
        """Warm up eagerly (all lazy state: kernel attributes, index maps, workspaces, optimiser
        tables), then capture two graphs on the same static batch: plain micro-step and boundary
        micro-step (with clip + AdamW + weight-copy refresh + gradient zeroing)."""
        if getattr(self.model, "drop_svmr_prob", 0) > 0 and getattr(self.model, "lw_st_ed", 0) != 0:
            raise RuntimeError("graph mode cannot capture a step whose loss terms are drawn per step on the host "
                               "(drop_svmr_prob > 0, model/pretrain.py:74-75): the branch taken at capture would be replayed "
                               "forever; run eagerly")
        accum = self.opts.gradient_accumulation_steps
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(2 * accum):
                self.arena.set_sync((i + 1) % accum == 0)
                self._fwd_bwd(batch, task)
                if (i + 1) % accum == 0:
                    lr = self._set_lr()
                    self._lr_t.fill_(lr)
                    self._optimise(device_state=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.micro += 2 * accum
        self._record(batch, task)

    @staticmethod
    def _key(batch, task):
        """Graphs are per task and - for the static batches of a hero_amd.loader.BucketedBatchFeeder - per bucket shape."""
        bucket = batch.get("_bucket") if hasattr(batch, "get") else None
        return task if bucket is None else (task, bucket)

    def _eager_step_in_graph_mode(self, batch, task):
        """One micro-step with eager launches that keeps the graph mode's books: the accumulation window, the device-side
        optimiser step / learning rate the captured boundary graphs read."""
        accum = self.opts.gradient_accumulation_steps
        boundary = (self.micro + 1) % accum == 0
        self.counts["eager_in_graph_mode"] += 1
        self._window.append(task)
        if len(set(self._window)) > 1:
            raise RuntimeError("graph mode: one accumulation window mixes tasks %s" % sorted(set(self._window)))
        if boundary:
            self._window = []
        self.arena.set_sync(boundary)
        loss = self._fwd_bwd(batch, task)
        self.micro += 1
        if boundary:
            self._lr_t.fill_(self._set_lr())
            self._optimise(device_state=True)
        return loss

    def _record(self, batch, task):
        """Capture the two graphs of `task` (every lazy state exists already).  Also used to RE-capture a task whose
        graphs are older than the newest compute copy of a weight: the captured weight-copy refresh covers the copies
        that existed at capture time only, so a task captured later (the tied MLM decoder copy, a head's weights)
        would otherwise read stale copies of weights this task's optimiser step changes."""
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread queries events of earlier collectives: only THIS thread's
        # calls may invalidate the capture
        mode = dict(capture_error_mode="thread_local") if D.collectives_active() else {}
        self.arena.set_sync(False)                   # accumulation micro-step: no collective in this graph
        with torch.cuda.graph(ga, **mode):
            loss_a = self._fwd_bwd(batch, task)
        self.arena.set_sync(True)                    # boundary: bucket all-reduces (RCCL's stream joins the capture)
        with torch.cuda.graph(gb, pool=ga.pool(), **mode):
            loss_b = self._fwd_bwd(batch, task)
            self._optimise(device_state=True)
        key = self._key(batch, task)
        self.counts["captures"] += 1
        self._graphs[key] = (ga, loss_a, gb, loss_b, batch)
        self._graph_gen = getattr(self, "_graph_gen", {})
        self._graph_gen[key] = HF.weight_cache_generation()
        self._active_of = getattr(self, "_active_of", {})
        self._active_of[key] = list(getattr(self.optimizer, "last_active", []))     # for prebuild() before a re-capture

    def prepare(self, batch, task=None):
        """One-time setup outside any timed region: in graph mode the eager warm-up micro-steps and the
        capture of the two graphs of `task` on `batch`'s buffers (a no-op otherwise / when already done)."""
        task = task or self.task
        if self.use_graph:
            if self._key(batch, task) not in self._graphs:
                self._capture(batch, task)
        elif task not in getattr(self, "_prepared", set()):
            for _ in range(self.opts.gradient_accumulation_steps):   # one full cycle: every lazy state exists
                self.micro_step(batch, task)
            self._prepared = getattr(self, "_prepared", set()) | {task}

    def _graph_step(self, batch, task):
        """Replay of the graphs captured for `task`.  The captured region holds everything that was decided on the
        host while capturing: the batch's index / mask / pack-plan tensors, and the set of parameters the optimiser
        skips.  Hence (a) only the captured batch OBJECT is accepted - new data of identical structure (same masks,
        lengths, frame maps) may be copied into its tensors, anything else needs a new capture; (b) all micro-steps
        of one accumulation window must be of the same task."""
        key = self._key(batch, task)
        if key not in self._graphs:
            if not any((k if isinstance(k, str) else k[0]) == task for k in self._graphs):
                self._capture(batch, task)           # first graphs of this task: full eager warm-up (every lazy state), then capture
            else:
                # another BUCKET of a task that is already captured (BucketedBatchFeeder): every lazy state that does not
                # depend on the batch exists.  Its FIRST batch runs eagerly on the new static batch (memoised tensors,
                # weight-gradient plans and workspaces come into being - no extra optimiser steps, no batch seen twice);
                # the bucket's two graphs are captured at the first window boundary it meets - right behind an eager step
                # of its own that closed a window, or the next time it comes up at the start of one (a capture runs the
                # host side of a whole step: it must not land between a window's micro-steps) - i.e. by its second batch.
                seen = self.__dict__.setdefault("_seen_eager", set())
                if key in seen and self.micro % self.opts.gradient_accumulation_steps == 0:
                    torch.cuda.synchronize()
                    self._record(batch, task)
                else:
                    seen.add(key)
                    loss = self._eager_step_in_graph_mode(batch, task)
                    if self.micro % self.opts.gradient_accumulation_steps == 0:
                        # that eager step closed its window: this IS a window start, and the bucket's lazy state exists now -
                        # capture at once (a bucket whose batches always fall on the boundary micro-step - eight batches
                        # cycling under accumulation 2 keep their parity - would otherwise never meet a window start)
                        torch.cuda.synchronize()
                        self._record(batch, task)
                    return loss
        accum = self.opts.gradient_accumulation_steps
        if self.micro % accum == 0 and self._graph_gen[key] != HF.weight_cache_generation():
            torch.cuda.synchronize()                 # window start: re-capture against the current set of weight copies
            if hasattr(self.optimizer, "prebuild"):  # its descriptor table holds the copies' addresses: build the new one eagerly
                self.optimizer.prebuild(self._active_of.get(key, []))
            self._record(self._graphs[key][4], task)
        ga, loss_a, gb, loss_b, static_batch = self._graphs[key]
        if batch is not static_batch:
            raise RuntimeError("graph mode replays the captured batch buffers (and the index / mask tensors derived "
                               "from them at capture): copy new data of the SAME structure into them, or run eagerly")
        accum = self.opts.gradient_accumulation_steps
        boundary = (self.micro + 1) % accum == 0
        self._window.append(task)
        if len(set(self._window)) > 1:
            raise RuntimeError("graph mode: one accumulation window mixes tasks %s; the captured optimiser step only "
                               "updates the parameters of its own task" % sorted(set(self._window)))
        if boundary:
            self._window = []
        self.micro += 1
        self.counts["replayed"] += 1
        if boundary:
            self._lr_t.fill_(self._set_lr())        # stream-ordered scalar fill (no pinned-buffer race)
            gb.replay()
            return loss_b
        ga.replay()
        return loss_a

    # ---- checkpointing ---------------------------------------------------------------------------------
    def _sync_optimizer_steps(self):
        """hipGraph replays advance the optimiser's per-parameter step counts on the device only; bring the host-side
        counters (bias-correction steps in optimizer.state) up to date before they are saved."""
        self.optimizer.sync_steps_from_device()

    def state_dict(self):
        self._sync_optimizer_steps()
        return {"global_step": self.global_step, "micro": self.micro, "optimizer": self.optimizer.state_dict(),
                "optimizer_global_step": self.optimizer._global_step}

    def load_state_dict(self, sd):
        if self._graphs:
            raise RuntimeError("restore the training state before the first graph-mode step (captured graphs hold "
                               "the optimiser's device state)")
        self.optimizer.load_state_dict(sd["optimizer"])
        self.optimizer._global_step = sd["optimizer_global_step"]
        self.global_step, self.micro = sd["global_step"], sd["micro"]
        self._step_t.fill_(self.global_step)
        HF.notify_weights_updated()
