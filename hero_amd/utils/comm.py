"""Host side of `hero_comm_*` (include/hero_hip.h): the gradient exchange over RCCL behind the C ABI.

Opt-in (`hero_amd.utils.distributed.set_exchange("abi")`, `bench.py --exchange abi`); the default exchange goes through torch.distributed's
process group (hero_amd/utils/distributed.py).  What the ABI path changes: every collective is enqueued on a HIP stream
this module owns - no process-group stream, no watchdog thread, no Work objects - so a bucket's all-reduce is an
ordinary node of the stream (or of the hipGraph being captured on it).  The 128-byte RCCL unique id travels from rank 0
over the torch.distributed group that exists anyway (rendezvous is its job); with one rank no group is needed at all.

Replaces: utils/distributed.py:19-46 (allreduce_), :103-151 (broadcast_), model/pretrain.py:427-451 (allgather).
"""
import ctypes as C

import torch
import torch.distributed as dist

from .. import _lib as L


class Communicator:
    """One RCCL communicator of this process (= this GPU) and the side stream its collectives run on."""

    def __init__(self, device=None):
        lib = L.lib()
        if not lib.hero_comm_available():
            raise RuntimeError("hero_amd.utils.comm: librccl.so could not be opened; the ABI exchange has no fallback "
                               "(set_exchange('torch') uses torch.distributed's process group)")
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if on else 0
        self.world = dist.get_world_size() if on else 1
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = (C.c_char * 128)()
        if self.rank == 0:
            L.check(lib.hero_comm_unique_id(uid))
        if self.world > 1:
            dev = self.device if dist.get_backend() == "nccl" else "cpu"
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).to(dev)
            dist.broadcast(t, src=0)
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(lib.hero_comm_init(uid, self.rank, self.world, C.byref(self._h)))
        self.stream = torch.cuda.Stream(device=self.device)
        self._done = torch.cuda.Event()

    def close(self):
        if self._h:
            torch.cuda.synchronize(self.device)
            L.check(L.lib().hero_comm_destroy(self._h))
            self._h = C.c_void_p()

    # every method takes the stream to enqueue on (None = the current torch stream)
    def _s(self, stream):
        return C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)

    def allreduce_buckets(self, tensors, stream=None):
        """In-place SUM over ranks of up to 64 fp32 / bf16 device tensors, as one RCCL group."""
        lib = L.lib()
        for i in range(0, len(tensors), 64):
            part = tensors[i:i + 64]
            arr = (L.CommBucket * len(part))()
            for a, t in zip(arr, part):
                a.buf, a.count, a.dtype = L.ptr(t), t.numel(), L.dt(t)
            L.check(lib.hero_comm_allreduce_buckets(self._h, arr, len(part), self._s(stream)))

    def broadcast(self, t, root=0, stream=None):
        L.check(L.lib().hero_comm_broadcast(self._h, L.ptr(t), t.numel() * t.element_size(), root, self._s(stream)))

    def allgather(self, send, stream=None):
        """[world, *send.shape] with slice r = rank r's `send` (equal shapes on every rank)."""
        out = send.new_empty((self.world,) + tuple(send.shape))
        L.check(L.lib().hero_comm_allgather(self._h, L.ptr(send), L.ptr(out), send.numel() * send.element_size(),
                                            self._s(stream)))
        return out

    def allgather_var(self, send, dims, stream=None):
        """cat of the ranks' `send` tensors along dim 0; dims[r] = rank r's dim-0 size (host ints, equal on every rank)."""
        row = send.element_size()
        for d in send.shape[1:]:
            row *= int(d)
        out = send.new_empty((sum(dims),) + tuple(send.shape[1:]))
        if out.numel() == 0:
            return out
        arr = (C.c_size_t * self.world)(*[int(d) * row for d in dims])
        L.check(L.lib().hero_comm_allgather_var(self._h, L.ptr(send) if send.numel() else None, L.ptr(out), arr, self._s(stream)))
        return out

    # fork / join of the side stream against the current one (captured as graph dependencies under stream capture)
    def fork(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def join(self):
        self._done.record(self.stream)
        torch.cuda.current_stream(self.device).wait_event(self._done)


_COMM = [None]


def communicator():
    """The process-wide communicator (created collectively by the first call on every rank)."""
    if _COMM[0] is None:
        _COMM[0] = Communicator()
    return _COMM[0]


def shutdown():
    if _COMM[0] is not None:
        _COMM[0].close()
        _COMM[0] = None
