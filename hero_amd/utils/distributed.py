"""Data-parallel collectives over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" in the CPU tests) replacing the reference's Horovod calls (utils/distributed.py,
model/pretrain.py:427-451).

One process per GPU.  Semantics kept from the reference:
  * gradient all-reduce is an AVERAGE over ranks (Horovod's allreduce_ default) followed by a
    division by `rescale_denom`; it happens once per optimiser step, before clipping;
  * the cross-GPU negative gather has NO collective in its backward: every rank computes the
    same global loss and gradients are averaged afterwards, so each rank keeps its own slice.
MI355X-first additions: GradArena keeps all gradients in ONE flat fp32 buffer that autograd
accumulates into in place, so buckets are all-reduced straight out of it (no flatten/unflatten
copies) and are launched from autograd hooks while the rest of backward is still running.
"""
import os

import torch
import torch.distributed as dist

from .. import functional as HF


def _on():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if _on() else 1


def rank():
    return dist.get_rank() if _on() else 0


# Which implementation carries the DEVICE collectives of this process (gradient buckets, parameter broadcast, the
# forward all-gather of the negatives).  "torch" (default): torch.distributed's process group - the path every multi-rank
# test covers (two ranks over gloo on CPU and on the GPU box, one rank over RCCL).  "abi": hero_comm_* of the C ABI
# (hero_amd/utils/comm.py; RCCL enqueued on ONE side stream this process owns).  One setting per process, chosen by
# set_exchange() BEFORE the first TrainStep - no environment switch; DESIGN.md section 6 says why "torch" is the default.
_EXCHANGE = ["torch"]


def set_exchange(kind):
    """Collective on every rank.  "abi" creates the RCCL communicator NOW (outside any stream capture: ncclCommInitRank
    and the unique-id broadcast are illegal under capture, and a rank-divergent first use would deadlock)."""
    if kind not in ("torch", "abi"):
        raise ValueError("set_exchange: 'torch' or 'abi'")
    if kind == "abi":
        from . import comm
        comm.communicator()
    _EXCHANGE[0] = kind


def exchange():
    return _EXCHANGE[0]


def _abi_on(t):
    return _EXCHANGE[0] == "abi" and t.is_cuda


def collectives_active():
    """True when gradient buckets really travel through torch.distributed: more than one rank, or an initialised
    1-rank group with HERO_DP_FORCE_COLLECTIVES=1 (how the RCCL path - init, comm-stream ordering, bf16 wire - is
    exercised on a 1-GPU box, tests/test_gpu_distributed.py)."""
    return _on() and (dist.get_world_size() > 1 or bool(os.environ.get("HERO_DP_FORCE_COLLECTIVES")))


# ---- gradient averaging (utils/distributed.py:19-46) -----------------------------------------
def all_reduce_and_rescale_tensors(tensors, rescale_denom):
    """Flatten -> ONE all-reduce -> average -> / rescale_denom -> unflatten (in place)."""
    tensors = list(tensors)
    if not tensors:
        return
    n = world_size()
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if n > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / (n * float(rescale_denom)))
    off = 0
    for t in tensors:
        k = t.numel()
        t.copy_(flat[off:off + k].view_as(t))
        off += k


def _broadcast(t, root):
    if _abi_on(t):
        from . import comm
        c = comm.communicator()
        c.fork()                                   # every collective of the communicator runs on ITS stream, in program order
        c.broadcast(t, root, stream=c.stream)
        c.join()
    else:
        dist.broadcast(t, src=root)


def broadcast_tensors(tensors, root_rank, buffer_size=10485760):
    """Bucketed broadcast of parameters from `root_rank` (utils/distributed.py:103-151)."""
    if world_size() == 1:
        return
    bucket, filled = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        _broadcast(flat, root_rank)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()

    for t in tensors:
        sz = t.numel() * t.element_size()
        if sz > buffer_size and t.is_contiguous():
            _broadcast(t, root_rank)
            continue
        if filled + sz > buffer_size:
            flush()
            bucket, filled = [], 0
        bucket.append(t)
        filled += sz
    flush()


class GradArena(HF.GradSink):
    """Flat fp32 gradient arena with bucketed, backward-overlapped all-reduce.

    `p.grad` of every parameter is a view into ONE buffer.  The HIP backward kernels accumulate
    into it directly (this object is installed as hero_amd.functional's gradient sink); the few
    parameters that go through stock autograd (task-head ops) accumulate in place through
    AccumulateGrad.  Buckets are cut in reverse parameter order (~ the order gradients become
    final); a bucket's all-reduce is issued asynchronously the moment its last gradient is final,
    while the rest of backward is still running; `finish()` issues the leftovers and waits.
    A parameter used several times in one forward (the cross-modal encoder runs on subtitles AND
    on queries) is final when its last pending use has been back-propagated (`use`/`done`).
    The 1/world_size factor is NOT applied here: hand `grad_scale = 1/world_size` to the fused
    optimiser or call `scale_()`.
    """

    def __init__(self, params, bucket_bytes=64 << 20, overlap=True, install=True, groups=(),
                 static_usage=False, compress=None, backend=None):
        """groups: tuples of parameters whose gradients must be CONTIGUOUS in the arena, in the given
        order (e.g. an attention block's query/key/value weights: the backward then writes
        d[Wq;Wk;Wv] with one GEMM instead of three, see hero_amd.functional._qkv_bwd)."""
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slices = {}
        self.buckets = []            # [start, end, n_params]
        self.bucket_of = {}
        group_of = {}
        for g in groups:
            g = tuple(g)
            if all(q.requires_grad and q.numel() % 4 == 0 for q in g):
                for q in g:
                    group_of[q] = g
        off, start, count = 0, 0, 0
        for p in reversed(self.params):
            if p in self.slices:
                continue                     # already placed with its group
            for q in group_of.get(p, (p,)):
                n = q.numel()
                pad = (-off) % 4             # keep every slice 16-byte aligned for the kernels
                off += pad
                self.slices[q] = (off, off + n)
                self.bucket_of[q] = len(self.buckets)
                off += n
                count += 1
            if (off - start) * 4 >= bucket_bytes:
                self.buckets.append([start, off, count])
                start, count = off, 0
        if count:
            self.buckets.append([start, off, count])
        if off > total:
            self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for p in self.params:
            s, e = self.slices[p]
            p.grad = self.flat[s:e].view_as(p)
        self.sync = True
        self.overlap = overlap
        # compress="bf16": a bucket travels as bf16 (the reference all-reduces fp16 gradients: 242 MB instead of
        # 484 MB per optimiser step, utils/distributed.py:27-38 under amp O2) - cast into a per-bucket wire buffer
        # when the bucket is issued, summed by RCCL in bf16, widened back into the fp32 arena in finish()
        self.compress = compress
        if compress not in (None, "bf16"):
            raise ValueError("GradArena: compress must be None or 'bf16'")
        self._wire = {}
        # backend: "torch" = torch.distributed's process group; "abi" = hero_comm_* of the C ABI (RCCL enqueued on the side
        # stream this process owns, hero_amd/utils/comm.py); default: the process-wide set_exchange() choice
        if backend is None:
            backend = "abi" if _EXCHANGE[0] == "abi" and dev.type == "cuda" else "torch"
        if backend not in ("torch", "abi"):
            raise ValueError("GradArena: backend must be 'torch' or 'abi'")
        if backend == "abi" and dev.type != "cuda":
            raise ValueError("GradArena: backend 'abi' (hero_comm_* = RCCL) needs a CUDA arena")
        self.backend = backend
        self._comm = None
        self._abi_open = False
        if backend == "abi":               # the communicator exists before any backward pass (or capture) needs it
            self._abi()
        # static_usage: the caller guarantees that every optimiser step touches the same parameters
        # (one task, fixed graph).  Buckets then wait only for the parameters that received a gradient
        # in the previous step, so a bucket that also holds never-used parameters (pooler, lm_head,
        # ...) is still all-reduced as soon as its used gradients are final instead of in finish().
        self.static_usage = static_usage
        self.mute = False            # measurement switch (bench.py): the same steps WITHOUT the gradient exchange
        self._expect = None
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._final = set()
        self._uses = {}
        self._handles = []
        self.touched = set()
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._autograd_hook)
        if install:
            HF.set_grad_sink(self)

    # ---- GradSink interface (HIP kernels) --------------------------------------------------------
    def dst(self, p):
        sl = self.slices.get(p)
        if sl is None:
            return super().dst(p)
        if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + sl[0] * 4:
            p.grad = self.flat[sl[0]:sl[1]].view_as(p)
        return p.grad

    def dst_group(self, params):
        """One [sum(rows), cols] view over the gradients of `params` if they lie back to back in the
        arena (see `groups`), else None."""
        sl = [self.slices.get(p) for p in params]
        if any(x is None for x in sl) or any(sl[i][1] != sl[i + 1][0] for i in range(len(sl) - 1)):
            return None
        for p in params:
            self.dst(p)
        tail = params[0].shape[1:]
        if any(p.shape[1:] != tail for p in params):
            return None
        return self.flat[sl[0][0]:sl[-1][1]].view((-1,) + tuple(tail))

    def use(self, p):
        self._uses[p] = self._uses.get(p, 0) + 1

    def done(self, p):
        n = self._uses.get(p, 1) - 1
        self._uses[p] = n
        if n <= 0:
            self._on_final(p)

    # ---- stock autograd path -----------------------------------------------------------------------
    def _autograd_hook(self, p):
        sl = self.slices[p]
        if p.grad.data_ptr() != self.flat.data_ptr() + sl[0] * 4:
            # something replaced .grad (e.g. zero_grad(set_to_none=True)); fold it back
            self.flat[sl[0]:sl[1]].add_(p.grad.reshape(-1))
            p.grad = self.flat[sl[0]:sl[1]].view_as(p)
        if self._uses.get(p, 0) > 0:
            # a parameter of the HIP backward (registered by `use`): AccumulateGrad - and this hook - also runs when the
            # node returns no gradient for it.  Its finality is `done`'s call, which may come LATER than the node
            # (weight gradients are queued and launched in groups, functional.k_wgrad): do not count it here.
            return
        self._on_final(p)

    def _on_final(self, p):
        self.touched.add(p)
        if p in self._final or p not in self.bucket_of:
            return
        self._final.add(p)
        if not (self.sync and self.overlap) or not collectives_active() or self.mute:
            return
        b = self.bucket_of[p]
        if self._expect is not None and p not in self._expect:
            # a parameter that no earlier step touched (e.g. the st/ed head after steps that dropped it,
            # model/pretrain.py:74-75): its bucket was not counting on it.  Still correct if the bucket
            # has not been issued yet - hold it back for finish() and expect the parameter from now on.
            if self._launched[b]:
                raise RuntimeError("GradArena(static_usage=True): a parameter of shape %s received its first "
                                   "gradient after its bucket's all-reduce was issued; construct the arena "
                                   "with static_usage=False for steps that change the set of used "
                                   "parameters" % (tuple(p.shape),))
            self._expect = self._expect | {p}
            self._pending[b] = 1 << 30
            return
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        s, e, _ = self.buckets[b]
        if self.backend == "abi":
            # fork the communicator's stream off the current one (the bucket's gradients are final in stream order),
            # all-reduce there while backward continues here; finish() joins
            c = self._abi()
            buf = self.flat[s:e]
            if self.compress == "bf16":
                buf = self._wire.get(b)
                if buf is None:
                    buf = self._wire[b] = torch.empty(e - s, dtype=torch.bfloat16, device=self.flat.device)
                # the cast stays on THIS stream: a bandwidth kernel running beside the persistent GEMMs of backward takes
                # CUs from their one-workgroup-per-CU launch and the late workgroups become a second round (measured:
                # 7.50 vs 7.17 ms per captured micro-step with the cast on the side stream)
                buf.copy_(self.flat[s:e])
            c.fork()
            c.allreduce_buckets([buf], stream=c.stream)
            self._abi_open = True
            self._handles.append((None, b if self.compress == "bf16" else None))
        elif self.compress == "bf16":
            w = self._wire.get(b)
            if w is None:
                w = self._wire[b] = torch.empty(e - s, dtype=torch.bfloat16, device=self.flat.device)
            w.copy_(self.flat[s:e])
            self._handles.append((dist.all_reduce(w, op=dist.ReduceOp.SUM, async_op=True), b))
        else:
            self._handles.append((dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True), None))

    def _abi(self):
        if self._comm is None:
            from . import comm
            self._comm = comm.communicator()
        return self._comm

    def set_sync(self, flag):
        """Call at the start of every micro-step; False on gradient-accumulation micro-steps that do
        not end in an optimiser step."""
        self.sync = flag
        self._final.clear()          # finality is per backward pass (accumulation adds to the same slots)
        self._uses.clear()

    def wants_overlap(self):
        return bool(self.sync) and collectives_active()

    def finish(self):
        """Issue all-reduces for buckets the hooks did not complete, then wait for everything."""
        if collectives_active() and self.sync and not self.mute:
            for b in range(len(self.buckets)):
                self._launch(b)
            if self._abi_open:
                self._comm.join()
                self._abi_open = False
            for h, b in self._handles:
                if h is not None:
                    h.wait()
                if b is not None:                       # widen the summed wire buffer back into the arena
                    s, e, _ = self.buckets[b]
                    self.flat[s:e].copy_(self._wire[b])
        self._handles = []
        if self.static_usage and self.touched:
            # union over all steps so far: a parameter that is used only on some steps (drop_svmr_prob,
            # task mixes) keeps its bucket waiting for finish() on the steps that skip it
            self._expect = frozenset(self.touched) | (self._expect or frozenset())
        if self._expect is not None:
            self._pending = [0] * len(self.buckets)
            for p in self._expect:
                self._pending[self.bucket_of[p]] += 1
        else:
            self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._final.clear()
        self._uses.clear()

    def scale_(self, factor):
        self.flat.mul_(factor)

    def wire_bytes(self):
        """Bytes one optimiser step puts on the wire per rank (before the collective algorithm's own factor)."""
        per = 2 if self.compress == "bf16" else 4
        return sum((e - s) * per for s, e, _ in self.buckets)

    def probe_allreduce_ms(self, reps=3):
        """Time the bucketed all-reduce of one optimiser step ALONE (no backward to hide behind), on scratch buffers of
        the wire dtype and bucket sizes: what the exchange costs when nothing overlaps it.  Collective on every rank."""
        if not collectives_active():
            return 0.0
        dt = torch.bfloat16 if self.compress == "bf16" else torch.float32
        bufs = [torch.zeros(e - s, dtype=dt, device=self.flat.device) for s, e, _ in self.buckets]
        times = []
        if self.backend == "abi":
            c = self._abi()
            for _ in range(reps + 1):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                c.fork()
                c.allreduce_buckets(bufs, stream=c.stream)     # one group on the communicator's stream
                c.join()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            return sorted(times[1:])[len(times[1:]) // 2]
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hs = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in bufs]
            for h in hs:
                h.wait()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        return sorted(times[1:])[len(times[1:]) // 2]

    def zero(self):
        self.flat.zero_()
        self.touched.clear()
        for p in self.params:                       # re-attach after a foreign zero_grad()
            if p.grad is None:
                s, e = self.slices[p]
                p.grad = self.flat[s:e].view_as(p)


# ---- cross-GPU negatives (model/pretrain.py:383-401, 427-451) ---------------------------------
class _AllGatherRows(torch.autograd.Function):
    """Variable-dim-0 all-gather; backward = this rank's slice of the gradient, no collective."""

    @staticmethod
    def forward(ctx, tensor, dims):
        r, n = rank(), world_size()
        ctx.offset, ctx.dim = sum(dims[:r]), tensor.shape[0]
        if _abi_on(tensor):
            from . import comm                      # no padding, no cat: the ranks' rows land back to back
            c = comm.communicator()
            send = tensor.contiguous()
            c.fork()                                # on the communicator's one stream, like the gradient buckets; `send` and the
            out = c.allgather_var(send, dims, stream=c.stream)      # result are allocated on the CURRENT stream, which the
            c.join()                                # join orders behind the collective - no record_stream needed
            return out
        mx = max(dims)
        buf = tensor.new_zeros((mx,) + tuple(tensor.shape[1:]))
        buf[:tensor.shape[0]] = tensor
        out = [torch.empty_like(buf) for _ in range(n)]
        dist.all_gather(out, buf.contiguous())
        return torch.cat([o[:d] for o, d in zip(out, dims)], dim=0)

    @staticmethod
    def backward(ctx, grad):
        return grad.narrow(0, ctx.offset, ctx.dim), None


_HOST_GROUP = [None, None]


def _host_group():
    """A gloo process group for small host-side metadata (None = the default group when that one
    already is gloo)."""
    pg = dist.distributed_c10d._get_default_group()
    if _HOST_GROUP[0] is not pg:
        _HOST_GROUP[0] = pg
        if dist.get_backend() == "gloo":
            _HOST_GROUP[1] = None
        else:
            try:
                import os
                if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
                    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # single node: no hostname lookups
                _HOST_GROUP[1] = dist.new_group(backend="gloo")
            except Exception:                     # no usable host interface: sizes travel through RCCL
                _HOST_GROUP[1] = "device"
    return _HOST_GROUP[1]


# Every rank feeds batches of the same padded shape (fixed-shape training such as the bench's D2 batches, or a captured
# hipGraph): the per-forward size exchange below is skipped and every rank's sizes are taken to be this rank's.
UNIFORM_SHAPES = [False]


class uniform_shapes:
    """`with uniform_shapes(True):` every rank feeds batches of the same padded shape inside - gather_negatives skips its
    host-side size exchange.  Scoped (a TrainStep sets it around its own forwards): a second TrainStep in the process
    cannot flip the first one's behaviour."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev, UNIFORM_SHAPES[0] = UNIFORM_SHAPES[0], self.on

    def __exit__(self, *exc):
        UNIFORM_SHAPES[0] = self.prev


def gather_negatives(query, context, context_mask, return_own=False):
    """All-gather (queries, L2-normalised contexts, masks) across ranks, padding contexts to the
    global max clip length.  ONE small integer all-gather carries every size needed.
    return_own: also return (first video, number of videos) of this rank inside the gathered set."""
    n = world_size()
    if UNIFORM_SHAPES[0]:
        metas = [[query.shape[0], context.shape[0], context.shape[1]]] * n
    else:
        # the sizes are host values (tensor shapes): exchange them on the host (gloo side group) so the
        # forward pass has no device synchronisation in it
        hg = _host_group()
        meta = torch.tensor([query.shape[0], context.shape[0], context.shape[1]], dtype=torch.int64,
                            device=query.device if hg == "device" else "cpu")
        metas = [torch.empty_like(meta) for _ in range(n)]
        dist.all_gather(metas, meta, group=None if hg == "device" else hg)
        metas = torch.stack(metas).tolist()
    nq = [m[0] for m in metas]
    nv = [m[1] for m in metas]
    max_len = max(m[2] for m in metas)
    pad = max_len - context.shape[1]
    if pad:
        context = torch.cat([context, context.new_zeros(context.shape[0], pad, context.shape[2])], 1)
        context_mask = torch.cat([context_mask, context_mask.new_zeros(context_mask.shape[0], pad)], 1)
    q = _AllGatherRows.apply(query.contiguous(), nq)
    c = _AllGatherRows.apply(context.contiguous(), nv)
    m = _AllGatherRows.apply(context_mask.contiguous(), nv)
    if return_own:
        return q, c, m, (sum(nv[:rank()]), nv[rank()])
    return q, c, m


# ---- pickled-object helpers (utils/distributed.py:182-212) ------------------------------------
def all_gather_list(data):
    if world_size() == 1:
        return [data]
    out = [None] * world_size()
    dist.all_gather_object(out, data)
    return out


def any_broadcast(data, root_rank):
    if world_size() == 1:
        return data
    box = [data if rank() == root_rank else None]
    dist.broadcast_object_list(box, src=root_rank)
    return box[0]


__all__ = ["all_reduce_and_rescale_tensors", "broadcast_tensors", "GradArena", "gather_negatives",
           "all_gather_list", "any_broadcast", "world_size", "rank", "collectives_active"]
