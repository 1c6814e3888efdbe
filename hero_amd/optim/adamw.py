"""AdamW with the reference's update rule (optim/adamw.py:43-106) as ONE HIP kernel per tensor:
bias-corrected Adam step, then decoupled decay p -= lr * wd * p, eps = 1e-6 added to sqrt(v).
Gradient-norm clipping (train_vcmr.py:257-260) and the 1/world_size averaging can be folded into
the same pass (`step(grad_sumsq=..., max_grad_norm=..., grad_scale=...)`)."""
import ctypes as C

import torch
from torch.optim import Optimizer

from .. import _lib as L
from .. import functional as HF


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0,
                 correct_bias=True):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid AdamW hyper-parameters")
        if not correct_bias:
            raise NotImplementedError("correct_bias=False is not used by HERO")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias))

    def grad_sumsq(self):
        """Device scalar = sum of squared gradients over all parameters (for clipping)."""
        dev = self.param_groups[0]["params"][0].device if self.param_groups[0]["params"] else None
        for g in self.param_groups:
            for p in g["params"]:
                dev = p.device
                break
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    L.check(L.lib().hero_sumsq(L.ptr(gr), gr.numel(), L.ptr(out), L.stream()))
        return out

    @torch.no_grad()
    def step(self, closure=None, grad_sumsq=None, max_grad_norm=0.0, grad_scale=1.0, skip=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None or (skip is not None and p in skip):
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                a = L.AdamW(L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                            p.numel(), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                            st["step"], L.ptr(grad_sumsq), max_grad_norm, grad_scale, None)
                L.check(L.lib().hero_adamw(C.byref(a), L.stream()))
        HF.notify_weights_updated()
        return loss
