"""Host -> device hand-over of batches, one step ahead (loader half).

`MetaLoader` is the reference's multi-task scheduler (data/loader.py:19-59), `PrefetchLoader` the reference's class of the
same name (data/loader.py:89-144): it wraps any iterable of host
batches (a DataLoader with `collate_fn=hero_amd.collate.vcmr_collate` and `pin_memory=True`), moves batch k + 1 to the
device on a side stream while batch k is being consumed, and yields device batches - same protocol (`wait_stream`
before the hand-over, `record_stream` on every tensor), so a reference training loop takes it unchanged.

`StaticBatchFeeder` is what a hipGraph-captured step needs on top of that: the captured micro-step reads FIXED
buffers, so the feeder owns (a) the static batch the graphs are captured on, (b) two staging sets that the copy stream
fills alternately while the current replay runs, and (c) a small captured `commit` graph per staging set that moves
staging -> static on the compute stream and rebuilds everything derived from the payload on the device:
    staging c_v_feats (33 MB of fp32 frame features for the TVR batch: the only large H2D transfer) -> static
    f_v_feats         gathered from c_v_feats by hero_collate_gather_feats (not transferred: half the PCIe bytes)
    index / mask tensors, frame map   hero_amd.collate.DeviceCollate.rebuild from ~1 KB of lengths
    memoised derived tensors          hero_amd.functional.refresh_memo
Only batches of the SAME padded shape can share a captured step (ragged batches of varying shape run eagerly through
PrefetchLoader)."""

import torch

from . import functional as HF
from .collate import DeviceCollate

# tensors of a vcmr batch that carry payload (everything else is derived from lengths on the device)
PAYLOAD = ("f_sub_input_ids", "c_v_feats", "query_input_ids", "query_attn_masks", "targets", "q_vidx")
DERIVED = ("f_v_feats", "f_attn_masks", "f_gather_index", "c_attn_masks")


def move_to_device(batch, device, non_blocking=True):
    """data/loader.py:49-66 (move_to_cuda): tensors are moved, lists / tuples / dicts are walked RECURSIVELY (the reference
    wraps `PrefetchLoader(MetaLoader(...))`, whose items are `(task, batch_dict)` tuples); anything else passes through."""
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=non_blocking)
    if isinstance(batch, dict):
        return {k: move_to_device(v, device, non_blocking) for k, v in batch.items()}
    if isinstance(batch, tuple) and hasattr(batch, "_fields"):          # namedtuple
        return type(batch)(*(move_to_device(v, device, non_blocking) for v in batch))
    if isinstance(batch, (list, tuple)):
        return type(batch)(move_to_device(v, device, non_blocking) for v in batch)
    return batch


def _record_stream(batch, stream):
    if torch.is_tensor(batch):
        if batch.is_cuda:
            batch.record_stream(stream)
    elif isinstance(batch, dict):
        for v in batch.values():
            _record_stream(v, stream)
    elif isinstance(batch, (list, tuple)):
        for v in batch:
            _record_stream(v, stream)


class MetaLoader:
    """Scheduler of a multi-task run (the reference's class of this name, data/loader.py:19-59; pretrain.py's loop
    iterates it): yields `(task, batch)` forever.

    loaders        {task: DataLoader} or {task: (any iterable of batches, ratio)} - a task's chance is proportional to its
                   integer ratio (1 for a bare DataLoader); anything else raises ValueError, as in the reference
    accum_steps    a new task is drawn (Python's `random`) only every accum_steps micro-steps: an accumulation window
                   stays on one task
    distributed    rank 0's draw is broadcast, so every rank trains the same task (the reference's Horovod
                   `any_broadcast`; here `hero_amd.utils.distributed.any_broadcast` over the process group)
    A loader that runs out is restarted.  Attributes name2loader / name2iter / sampling_pools / step are the reference's."""

    def __init__(self, loaders, accum_steps=1, distributed=False):
        assert isinstance(loaders, dict)
        self.name2loader, self.sampling_pools = {}, []
        for task, entry in loaders.items():
            if isinstance(entry, tuple):
                loader, ratio = entry
            elif isinstance(entry, torch.utils.data.DataLoader):
                loader, ratio = entry, 1
            else:
                raise ValueError()
            self.name2loader[task] = loader
            self.sampling_pools += [task] * ratio
        self.name2iter = {task: iter(loader) for task, loader in self.name2loader.items()}
        self.accum_steps, self.distributed, self.step = accum_steps, distributed, 0

    def _draw(self):
        import random
        from .utils.distributed import any_broadcast
        task = random.choice(self.sampling_pools)
        return any_broadcast(task, 0) if self.distributed else task

    def _batch_of(self, task):
        try:
            return next(self.name2iter[task])
        except StopIteration:
            self.name2iter[task] = iter(self.name2loader[task])
            return next(self.name2iter[task])

    def __iter__(self):
        task = self.sampling_pools[0]
        while True:
            if self.step % self.accum_steps == 0:
                task = self._draw()
            self.step += 1
            yield task, self._batch_of(task)


class PrefetchLoader:
    """overlap compute and host -> device transfer (data/loader.py:89-144)"""

    def __init__(self, loader, device=None):
        self.loader = loader
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.stream = torch.cuda.Stream(self.device)

    def __iter__(self):
        it = iter(self.loader)
        self.preload(it)
        batch = self.next(it)
        while batch is not None:
            yield batch
            batch = self.next(it)

    def __len__(self):
        return len(self.loader)

    def preload(self, it):
        try:
            self.batch = next(it)
        except StopIteration:
            self.batch = None
            return
        with torch.cuda.stream(self.stream):
            self.batch = move_to_device(self.batch, self.device)

    def next(self, it):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        batch = self.batch
        if batch is not None:
            _record_stream(batch, torch.cuda.current_stream(self.device))
        self.preload(it)
        return batch

    def __getattr__(self, name):
        return getattr(self.loader, name)


def pin_batch(batch):
    """Page-lock the payload tensors of a host batch (what DataLoader(pin_memory=True) does in its pin thread)."""
    out = dict(batch)
    for k in PAYLOAD:
        if torch.is_tensor(out.get(k)) and not out[k].is_pinned():
            out[k] = out[k].pin_memory()
    out["lengths"] = {k: torch.as_tensor(v, dtype=torch.int32).pin_memory() for k, v in batch["lengths"].items()}
    return out


class StaticBatchFeeder:
    def __init__(self, host_batch, device, capture_commit=True):
        """host_batch: a vcmr_collate batch (with `lengths`) that fixes the padded shapes."""
        self.device = torch.device(device)
        D = host_batch["c_v_feats"].shape[2]
        self.dc = DeviceCollate.for_batch(host_batch, self.device, vfeat_dim=D if D % 4 == 0 else None)
        self.shapes = {k: tuple(host_batch[k].shape) for k in PAYLOAD if torch.is_tensor(host_batch.get(k))}
        dev = lambda t: torch.empty(t.shape, dtype=t.dtype, device=self.device)      # noqa: E731
        self.static = {k: v.to(self.device) for k, v in host_batch.items()
                       if torch.is_tensor(v) and k not in DERIVED and k not in ("c_pos_ids", "f_sub_input_attn_masks")}
        if self.dc.f_v_feats is None:
            self.static["f_v_feats"] = host_batch["f_v_feats"].to(self.device)
        self.static.update(self.dc.batch_entries())
        self.static["_static_buffers"] = True        # TrainStep: padded formulation only (the pack plan is host-derived)
        # TWO staging sets, used alternately: the copy stream fills one while the commit of the previous batch reads the
        # other, so the only ordering the copy stream needs - "the commit that read this set two batches ago is done" -
        # is checked on the HOST (Event.synchronize, already satisfied in steady state).  A stream-side
        # wait_event on an event of the compute stream cost 0.28 ms per step (tools/lab/feedprobe.py): the runtime turns
        # the recorded event into a barrier packet with a completion signal in front of the step's graph launch.
        self.stage = [{k: dev(host_batch[k]) for k in self.shapes} for _ in range(2)]
        self.stage_len = [torch.zeros_like(self.dc._in_flat) for _ in range(2)]           # the six length arrays, one buffer
        self._pin_len = [torch.zeros(self.dc._in_flat.shape, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self._landed = [torch.cuda.Event(), torch.cuda.Event()]
        self._consumed = [torch.cuda.Event(), torch.cuda.Event()]
        self.skip_h2d = False     # lab switch (tools/lab/feedprobe.py): prefetch() does everything but the transfers
        self._fill = 0            # staging set the next prefetch writes
        self._ready = []          # staging sets that hold a prefetched, not yet committed batch (FIFO)
        self._graph = [None, None]
        self._orders = []         # (batch key, device order tensor, skip index): segment orders computed on the HOST (capture())
        self.stage_order = [{}, {}]
        self._order_ok = [False, False]
        self._versioned = None
        self._want_graph = capture_commit
        self.prefetch(host_batch)
        self.commit()

    def prefetch(self, host_batch):
        """Start the H2D copies of the NEXT batch on the copy stream (returns immediately).  The payload tensors
        should be pinned (pin_batch / DataLoader(pin_memory=True)), otherwise the copies are synchronous."""
        for k, shp in self.shapes.items():
            if tuple(host_batch[k].shape) != shp:
                raise ValueError("StaticBatchFeeder: %s has shape %s, the captured step was built for %s (run batches of "
                                 "other shapes eagerly through PrefetchLoader)" % (k, tuple(host_batch[k].shape), shp))
        if host_batch["f_attn_masks"].shape[1] != self.dc.Lf:
            raise ValueError("StaticBatchFeeder: f_attn_masks is %d wide, the captured step %d" % (host_batch["f_attn_masks"].shape[1], self.dc.Lf))
        if self.dc.f_v_feats is None:
            raise ValueError("StaticBatchFeeder needs vfeat_dim % 4 == 0 (f_v_feats is gathered on the device)")
        if len(self._ready) >= 2:
            raise RuntimeError("StaticBatchFeeder: two batches are already prefetched; commit() one first")
        s = self._fill
        self._fill ^= 1
        self._ready.append(s)
        self._consumed[s].synchronize()                                  # host-side; never recorded -> returns at once
        if self.skip_h2d:                                                # lab switch (tools/lab/feedprobe.py): everything but the transfers
            self._landed[s].record(self.copy_stream)
            return
        with torch.cuda.stream(self.copy_stream):
            for k in self.shapes:
                self.stage[s][k].copy_(host_batch[k], non_blocking=True)
            flat = self._pin_len[s]                           # one pinned image of the input buffer, one H2D copy
            flat.zero_()
            for k, (o, n) in self.dc._in_off.items():
                src = torch.as_tensor(host_batch["lengths"][k], dtype=torch.int32).reshape(-1)
                if src.numel() > n or (k in ("sub_nfrm", "sub_ntok", "vid_nfrm") and src.numel() != n):
                    raise ValueError("StaticBatchFeeder: %s has %d entries (capacity %d)" % (k, src.numel(), n))
                flat[o:o + src.numel()] = src
            self.stage_len[s].copy_(flat, non_blocking=True)
            # Segment orders of the id tensors (embedding-gradient scatter): a stable argsort of <= 10 k ids is ~0.3 ms of numpy
            # on a host that has ~6 ms of slack per step; on the device it was a one-workgroup radix sort inside every commit
            # (69 + 20 us of the commit graph's 239: tools/lab/feedprobe.py).
            for k, _, skip in self._orders:
                o = torch.from_numpy(HF.host_segment_order(host_batch[k].numpy(), skip))
                self._pin_order[s][k].copy_(o)
                self.stage_order[s][k].copy_(self._pin_order[s][k], non_blocking=True)
            self._order_ok[s] = bool(self._orders)
            self._landed[s].record(self.copy_stream)

    def _commit_body(self, s):
        for k in self.shapes:
            self.static[k].copy_(self.stage[s][k])
        self.dc.load_lengths(self.stage_len[s], src_device=True)
        self.dc.rebuild(c_v_feats=self.static["c_v_feats"])
        for k, out, _ in self._orders:                       # host-sorted orders: copied in, not re-sorted
            out.copy_(self.stage_order[s][k])
        HF.refresh_memo([t for t in self.static.values() if torch.is_tensor(t)], skip_outputs=[o for _, o, _ in self._orders])

    def commit(self):
        """Make the oldest prefetched batch the current one: on the compute stream, staging -> static buffers, then
        every derived tensor is rebuilt on the device.  Returns the static batch (always the same object)."""
        if not self._ready:
            raise RuntimeError("StaticBatchFeeder.commit: nothing prefetched")
        s = self._ready.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._landed[s])
        if self._graph[s] is not None:
            if self._orders and not self._order_ok[s]:
                raise RuntimeError("StaticBatchFeeder: this batch was prefetched before capture(); prefetch after capture()")
            self._graph[s].replay()
            # a replay does not move the version counters the eager-mode caches (functional.memo) key on
            torch._C._increment_version(self._versioned)       # takes an ITERABLE of tensors (a bare tensor is iterated row by row)
        else:
            self._commit_body(s)
        self._consumed[s].record(cur)
        return self.static

    def capture(self):
        """Capture the commit sequences (one per staging set) in hipGraphs of their own (call after the training step
        has been captured / warmed up, so that every memoised tensor exists): one graph launch per batch instead of
        ~40 small ones."""
        if not self._want_graph or self._graph[0] is not None:
            return
        if self._ready:
            raise RuntimeError("StaticBatchFeeder.capture: commit() the prefetched batch first (orders are computed on the host from now on)")
        torch.cuda.synchronize(self.device)
        # the step has run on self.static: its segment orders exist - take over the ones that depend on host ids only
        self._orders = HF.host_sortable_orders(self.static)
        self.stage_order = [{k: torch.empty_like(o) for k, o, _ in self._orders} for _ in range(2)]
        self._pin_order = [{k: torch.empty(o.shape, dtype=o.dtype).pin_memory() for k, o, _ in self._orders} for _ in range(2)]
        for s_ in range(2):                                  # valid contents for the capture run itself
            for k, o, _ in self._orders:
                self.stage_order[s_][k].copy_(o)
        for s in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._commit_body(s)
            self._graph[s] = g
        self._versioned = [t for t in self.static.values() if torch.is_tensor(t)] + [t for t in self.dc.frame_map]
        torch.cuda.synchronize(self.device)
