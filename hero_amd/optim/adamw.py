"""AdamW with the reference's update rule (optim/adamw.py:43-106): bias-corrected Adam step, then
decoupled decay p -= lr * wd * p, eps = 1e-6 added to sqrt(v) — as ONE multi-tensor HIP launch
per optimiser step (hero_adamw_multi).  Gradient-norm clipping (train_vcmr.py:257-260) and the
1/world_size averaging fold into the same pass (`step(grad_sumsq=..., max_grad_norm=...,
grad_scale=...)`).  Parameters whose .grad is None (or listed in `skip`) are left untouched, state
included, exactly like the reference's `if p.grad is None: continue`."""
import ctypes as C

import numpy as np
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
        if len(self.param_groups) > 8:
            raise ValueError("hero_amd AdamW supports up to 8 parameter groups")
        self._global_step = 0
        self._tables = {}         # signature -> (device tensors, n_chunks)
        self._pinned = []         # tables built or used while a hipGraph was capturing: the graph holds their addresses
        self._tsteps = None       # device-state mode: per-parameter step counts on the device (int32 [n params])
        self._slots = None        # parameter -> slot in _tsteps
        self.last_active = []     # (group index, parameter) of the last step(): what TrainStep hands to prebuild()

    def load_state_dict(self, state_dict):
        """A restore replaces the moment tensors the descriptor tables point at: drop the tables (the ones captured
        graphs use stay alive in `_pinned`, but such graphs must be re-captured - TrainStep refuses the restore)."""
        super().load_state_dict(state_dict)
        self._tables = {}
        self._tsteps = None       # re-seeded from the restored state['step'] at the next device-state step

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables, self._pinned = {}, []
        self._tsteps = self._slots = None

    # ---- gradient norm ---------------------------------------------------------------------------
    def grad_sumsq(self, flat=None):
        """Device scalar = sum of squared gradients (one launch over a flat arena if given)."""
        dev = next(p for g in self.param_groups for p in g["params"]).device
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        if getattr(self, "_sumsq_ws", None) is None or self._sumsq_ws.device != dev:
            self._sumsq_ws = torch.empty(2048, dtype=torch.float32, device=dev)
        ws = self._sumsq_ws
        if flat is not None:
            L.check(L.lib().hero_sumsq(L.ptr(flat), flat.numel(), L.ptr(out), L.ptr(ws), L.stream()))
            return out
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    L.check(L.lib().hero_sumsq(L.ptr(gr), gr.numel(), L.ptr(out), L.ptr(ws), L.stream()))
        return out

    # ---- descriptor table ------------------------------------------------------------------------
    def _device_steps(self, dev):
        """Per-parameter step counts on the device (seeded from the host-side state['step'])."""
        if self._tsteps is None:
            params = [p for g in self.param_groups for p in g["params"]]
            self._slots = {p: i for i, p in enumerate(params)}
            host = [int(self.state[p].get("step", 0)) if p in self.state else 0 for p in params]
            self._tsteps = torch.tensor(host, dtype=torch.int32, device=dev)
        return self._tsteps

    def sync_steps_from_device(self):
        """Bring state['step'] up to the device-side counters (hipGraph replays advance only those)."""
        if self._tsteps is None:
            return
        host = self._tsteps.cpu().tolist()
        for p, i in self._slots.items():
            if p in self.state and "step" in self.state[p]:
                self.state[p]["step"] = host[i]
        self._global_step = max([self._global_step] + host)

    def _build_table(self, active, slots=False):
        chunk = L.lib().hero_adamw_multi_chunk()
        descs = (L.TensorDesc * len(active))()
        ct, ci = [], []
        for i, (gi, p) in enumerate(active):
            st = self.state[p]
            g = p.grad
            if not (p.is_contiguous() and g.is_contiguous()):
                raise RuntimeError("hero_amd AdamW needs contiguous parameters and gradients")
            descs[i] = L.TensorDesc(L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                                    p.numel(), gi, self._slots[p] if slots else self._global_step - st["step"])
            n = -(-p.numel() // chunk)
            ct.extend([i] * n)
            ci.extend(range(n))
        dev = active[0][1].device
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        t_ct = torch.from_numpy(np.asarray(ct, dtype=np.int32)).to(dev)
        t_ci = torch.from_numpy(np.asarray(ci, dtype=np.int32)).to(dev)
        return raw, t_ct, t_ci, len(ct)

    @torch.no_grad()
    def step(self, closure=None, grad_sumsq=None, max_grad_norm=0.0, grad_scale=1.0, skip=None,
             step_tensor=None, lr_tensor=None):
        """step_tensor (int32[1]) / lr_tensor (float32[8]) on the device override the host-side step
        count and per-group learning rates — required when the step is captured in a hipGraph."""
        loss = closure() if closure is not None else None
        device_state = step_tensor is not None
        if not device_state:
            if self._tsteps is not None:                 # back from device-state mode: the device counters are the truth
                self.sync_steps_from_device()
                self._tsteps = None
            self._global_step += 1
        active = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None or (skip is not None and p in skip):
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not device_state:
                    st["step"] += 1
                active.append((gi, p))
        if not active:
            return loss
        tsteps = self._device_steps(active[0][1].device) if device_state else None
        self.last_active = active
        sig = self._table_for(active, device_state)
        raw, t_ct, t_ci, n_chunks = self._tables[sig]
        return self._launch(raw, t_ct, t_ci, n_chunks, len(active), tsteps, grad_sumsq, max_grad_norm, grad_scale, step_tensor, lr_tensor, loss)

    def prebuild(self, active, device_state=True):
        """Build (outside any stream capture) the descriptor table a later captured step() over the same parameters will
        need: a table holds the addresses of the parameters' straight compute copies, and a copy created since the last
        capture (another task's first forward) means a new table - whose upload is not capturable.  TrainStep calls this
        before it re-captures a task."""
        if active:
            self._device_steps(active[0][1].device)
            self._table_for(list(active), device_state)

    def _table_for(self, active, device_state):
        if device_state:
            # per-parameter counts live on the device and are advanced by the kernel for the ACTIVE parameters only:
            # graphs of other tasks never touch the counters of parameters they skip (the reference's state['step'],
            # optim/adamw.py:71-72)
            sig = ("dev",) + tuple((id(p), p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                                    self.state[p]["exp_avg_sq"].data_ptr(), self._slots[p]) for _, p in active)
        else:
            sig = tuple((id(p), p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                         self.state[p]["exp_avg_sq"].data_ptr(), self._global_step - self.state[p]["step"]) for _, p in active)
        capturing = torch.cuda.is_current_stream_capturing()
        if sig not in self._tables:
            if len(self._tables) >= 16 and not capturing:
                self._tables = {}                 # eager multi-task runs change the signature often; pinned tables survive
            self._tables[sig] = self._build_table(active, slots=device_state)
        if capturing and not any(t is self._tables[sig] for t in self._pinned):
            self._pinned.append(self._tables[sig])
        return sig

    def _launch(self, raw, t_ct, t_ci, n_chunks, n_active, tsteps, grad_sumsq, max_grad_norm, grad_scale, step_tensor, lr_tensor, loss):
        a = L.AdamWMulti()
        a.descs, a.chunk_tensor, a.chunk_index, a.n_chunks = raw.data_ptr(), t_ct.data_ptr(), t_ci.data_ptr(), n_chunks
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            a.groups[gi] = L.AdamWGroup(group["lr"], b1, b2, group["eps"], group["weight_decay"])
        a.step = max(self._global_step, 1)
        a.step_ptr, a.lr_ptr = L.ptr(step_tensor), L.ptr(lr_tensor)
        a.tensor_steps, a.n_tensors = L.ptr(tsteps), n_active
        a.grad_sumsq = L.ptr(grad_sumsq)
        a.max_grad_norm, a.grad_scale = max_grad_norm, grad_scale
        L.check(L.lib().hero_adamw_multi(C.byref(a), L.stream()))
        HF.notify_weights_updated()
        return loss
