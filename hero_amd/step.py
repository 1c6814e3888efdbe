"""Training micro-step harness reproducing the reference loop (train_vcmr.py:202-262):

  loss = loss_st_ed + loss_neg_ctx + loss_neg_q -> mean -> backward            every micro-step
  on every `gradient_accumulation_steps`-th micro-step:
      gradient all-reduce (average over ranks)  ->  lr = lr0 * warmup_linear(step)  ->
      clip_grad_norm_(1.0)  ->  AdamW.step()  ->  zero_grad()

MI355X-first differences: no host synchronisation inside the step (the reference calls .item() four
times per micro-step), gradients live in one flat arena that is all-reduced in buckets while
backward is still running, and clipping + averaging + AdamW are one kernel pass per tensor.
"""
from types import SimpleNamespace

import torch

from .optim import build_optimizer, get_lr_sched
from .utils import distributed as D

TVR_OPTS = dict(  # config/train-tvr-8gpu.json
    learning_rate=1e-4, lr_mul=1.0, weight_decay=0.01, optim="adamw", betas=[0.9, 0.98],
    grad_norm=1.0, warmup_steps=500, num_train_steps=5000, gradient_accumulation_steps=2,
    train_batch_size=32, dropout=0.1, lw_neg_q=8.0, lw_neg_ctx=8.0, lw_st_ed=0.01, margin=0.1)


class TrainStep:
    def __init__(self, model, opts=None, task="tvr", bucket_bytes=64 << 20):
        self.model = model
        self.opts = SimpleNamespace(**{**TVR_OPTS, **(opts or {})})
        self.task = task
        self.optimizer = build_optimizer(model, self.opts)
        self.arena = D.GradArena(list(model.parameters()), bucket_bytes=bucket_bytes)
        self.micro = 0
        self.global_step = 0
        D.broadcast_tensors([p.data for p in model.parameters()], 0)     # train_vcmr.py:152

    def micro_step(self, batch):
        """One forward+backward; optimiser step on accumulation boundaries. Returns the loss
        tensor (device-resident, not synchronised)."""
        accum = self.opts.gradient_accumulation_steps
        boundary = (self.micro + 1) % accum == 0
        self.arena.set_sync(boundary)
        l_st_ed, l_ctx, l_q = self.model(batch, task=self.task, compute_loss=True)
        loss = (l_st_ed + l_ctx + l_q).mean()
        loss.backward()
        self.micro += 1
        if boundary:
            self.arena.finish()
            self.global_step += 1
            lr = get_lr_sched(self.global_step, self.opts)
            for g in self.optimizer.param_groups:
                g["lr"] = lr                       # train_vcmr.py:248-249 (all groups)
            untouched = [p for p in self.arena.params if p not in self.arena.touched]
            sumsq = self.optimizer.grad_sumsq(self.arena.flat) if self.opts.grad_norm != -1 else None
            self.optimizer.step(grad_sumsq=sumsq, max_grad_norm=float(self.opts.grad_norm),
                                grad_scale=1.0 / D.world_size(), skip=set(untouched))
            self.arena.zero()
        return loss.detach()
