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
Only batches of the SAME padded shape can share a captured step.  `BucketedBatchFeeder` (round 6) is what real - ragged - TVR
batches need on top of that (every batch of data/data.py:406-471 has its own shape): a handful of BUCKET shapes, every host
batch padded up to the smallest bucket that holds it (masks keep the true lengths), one StaticBatchFeeder - and one pair
of captured step graphs (hero_amd.step.TrainStep keys its graphs by the batch's `_bucket`) - per bucket, and the six
cross-modal layers running PACKED through a static pack plan of the bucket's row capacity.  Batches no bucket holds run
eagerly through PrefetchLoader."""

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
    def __init__(self, host_batch, device, capture_commit=True, packed_rows=None, min_rows=0, frm_capacity=None):
        """host_batch: a vcmr_collate batch (with `lengths`) that fixes the padded shapes.
        packed_rows: also own a STATIC PACK PLAN (model/layers.py BertEncoder.register_static_plan) of that row capacity for
        the two sequence groups of the fused query pass (subtitle rows, query rows): the six cross-modal layers then run on
        the VALID positions only (+ the pad rows up to the capacity) also under hipGraph replay - every batch fed must have
        between min_rows and packed_rows valid positions.  The plan of the next batch is built on the host from its masks in
        prefetch() and copied in by commit().  frm_capacity: room for the frame lists (`lengths['sub_frm']`)."""
        self.device = torch.device(device)
        self._frm_capacity = frm_capacity
        D = host_batch["c_v_feats"].shape[2]
        self.dc = DeviceCollate.for_batch(host_batch, self.device, vfeat_dim=D if D % 4 == 0 else None, entry_capacity=frm_capacity)
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
        self._memo_keys = []
        self._want_graph = capture_commit
        self.plan = None
        if packed_rows:
            from .model.layers import BertEncoder
            groups = (tuple(host_batch["f_attn_masks"].shape), tuple(host_batch["query_attn_masks"].shape))
            self.plan = BertEncoder.static_plan_layout(groups, int(packed_rows), int(min_rows))
            n = self.plan["size"]
            self.plan_flat = torch.zeros(n, dtype=torch.int32, device=self.device)
            self.stage_plan = [torch.zeros(n, dtype=torch.int32, device=self.device) for _ in range(2)]
            self._pin_plan = [torch.zeros(n, dtype=torch.int32).pin_memory() for _ in range(2)]
            BertEncoder.register_static_plan([self.static["f_attn_masks"], self.static["query_attn_masks"]], self.plan_flat, self.plan)
            self.static["_static_plan"] = True       # TrainStep: packed through the registered plan, never through a host-derived one
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
            if self.plan is not None:                # the pack plan of THIS batch, from its host masks (numpy, ~0.2 ms)
                from .model.layers import BertEncoder
                BertEncoder.fill_static_plan(self._pin_plan[s].numpy(), self.plan,
                                             [host_batch["f_attn_masks"].numpy(), host_batch["query_attn_masks"].numpy()])
                self.stage_plan[s].copy_(self._pin_plan[s], non_blocking=True)
            self._landed[s].record(self.copy_stream)

    def _commit_body(self, s):
        for k in self.shapes:
            self.static[k].copy_(self.stage[s][k])
        self.dc.load_lengths(self.stage_len[s], src_device=True)
        self.dc.rebuild(c_v_feats=self.static["c_v_feats"])
        for k, out, _ in self._orders:                       # host-sorted orders: copied in, not re-sorted
            out.copy_(self.stage_order[s][k])
        if self.plan is not None:
            self.plan_flat.copy_(self.stage_plan[s])
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
            # ... but the entries the commit graph has just refreshed in place ARE current: re-key them, so that whoever looks
            # them up next (an eager step, the capture of another bucket's step graphs) finds them instead of rebuilding
            self._memo_keys = HF.restamp_memo(self._memo_keys)
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
        self._memo_keys = HF.last_refreshed_keys()           # the memo entries the commit graphs keep current
        self._versioned = [t for t in self.static.values() if torch.is_tensor(t)] + [t for t in self.dc.frame_map]
        torch.cuda.synchronize(self.device)


# ---- ragged batches under hipGraph replay: a handful of bucket shapes ------------------------------------------------------
def batch_dims(b):
    """The shape of a vcmr_collate batch as the captured step sees it + its packed row count."""
    T, max_vl = b["f_v_feats"].shape[:2]
    Bv, NF = b["c_v_feats"].shape[:2]
    nq, Lq = b["query_input_ids"].shape
    return dict(T=int(T), max_vl=int(max_vl), max_sl=int(b["f_sub_input_ids"].shape[1]), Lf=int(b["f_attn_masks"].shape[1]),
                B=int(Bv), NF=int(NF), nq=int(nq), Lq=int(Lq), frm=int(len(b["lengths"]["sub_frm"])),
                rows=int(b["f_attn_masks"].sum()) + int(b["query_attn_masks"].sum()))


def pad_batch(b, d):
    """A vcmr_collate host batch padded up to the bucket shape `d` (keys of batch_dims; `rows` is not a padded dimension):
    every tensor keeps its contents, the new rows / columns are padding in the reference's own conventions (token id 1, mask
    0, zero features - data/data.py:406-471), the extra subtitle rows (no frames, no tokens: fully masked) are appended to
    the last video, position ids are re-made for the new widths.  The model's results under the masks are those of the
    reference on the same padded batch - i.e. of these videos collated together with longer ones."""
    import numpy as np
    have = batch_dims(b)
    for k in ("T", "max_vl", "max_sl", "Lf", "NF", "Lq", "frm"):
        if have[k] > d[k]:
            raise ValueError("pad_batch: %s = %d does not fit the bucket's %d" % (k, have[k], d[k]))
    if have["B"] != d["B"] or have["nq"] != d["nq"]:
        raise ValueError("pad_batch: %d videos / %d queries, the bucket has %d / %d" % (have["B"], have["nq"], d["B"], d["nq"]))
    if d["Lf"] > d["max_vl"] + d["max_sl"]:
        raise ValueError("pad_batch: Lf %d > max_vl + max_sl = %d" % (d["Lf"], d["max_vl"] + d["max_sl"]))
    if all(have[k] == d[k] for k in ("T", "max_vl", "max_sl", "Lf", "NF", "Lq")):
        return b
    F = torch.nn.functional
    T, dT = have["T"], d["T"] - have["T"]
    out = dict(b)
    out["f_sub_input_ids"] = F.pad(b["f_sub_input_ids"], (0, d["max_sl"] - have["max_sl"], 0, dT), value=1)
    out["f_sub_pos_ids"] = torch.arange(d["max_sl"], dtype=torch.long).clamp_(max=511).unsqueeze(0)
    out["f_v_feats"] = F.pad(b["f_v_feats"], (0, 0, 0, d["max_vl"] - have["max_vl"], 0, dT))
    out["f_v_pos_ids"] = torch.arange(d["max_vl"], dtype=torch.long).unsqueeze(0)
    out["f_attn_masks"] = F.pad(b["f_attn_masks"], (0, d["Lf"] - have["Lf"], 0, dT))
    nfrm = np.concatenate([np.asarray(b["lengths"]["sub_nfrm"]), np.zeros(dT, np.int32)]).astype(np.int32)
    ntok = np.concatenate([np.asarray(b["lengths"]["sub_ntok"]), np.zeros(dT, np.int32)]).astype(np.int32)
    eff = np.maximum(nfrm, 1)
    gi = torch.arange(d["Lf"], dtype=torch.long).repeat(d["T"], 1)              # get_gather_index for the new max_vl
    for r in range(d["T"]):
        gi[r, eff[r]:eff[r] + ntok[r]] = torch.arange(d["max_vl"], d["max_vl"] + int(ntok[r]))
    out["f_gather_index"] = gi
    if "f_sub_input_attn_masks" in b:
        out["f_sub_input_attn_masks"] = F.pad(b["f_sub_input_attn_masks"], (0, d["max_sl"] - have["max_sl"], 0, dT))
    out["c_v_feats"] = F.pad(b["c_v_feats"], (0, 0, 0, d["NF"] - have["NF"]))
    out["c_attn_masks"] = F.pad(b["c_attn_masks"], (0, d["NF"] - have["NF"]))
    if "c_pos_ids" in b:
        out["c_pos_ids"] = torch.arange(d["NF"], dtype=torch.long).repeat(have["B"], 1)
    out["query_input_ids"] = F.pad(b["query_input_ids"], (0, d["Lq"] - have["Lq"]), value=1)
    out["query_pos_ids"] = torch.arange(d["Lq"], dtype=torch.long).unsqueeze(0)
    out["query_attn_masks"] = F.pad(b["query_attn_masks"], (0, d["Lq"] - have["Lq"]))
    ln = {k: np.asarray(v, dtype=np.int32) for k, v in b["lengths"].items()}
    ln["sub_nfrm"], ln["sub_ntok"] = nfrm, ntok
    ln["sub_frm_off"] = np.concatenate([ln["sub_frm_off"], np.full(dT, ln["sub_frm_off"][-1], np.int32)])
    ln["vid_sub_off"] = ln["vid_sub_off"].copy()
    ln["vid_sub_off"][-1] = d["T"]                            # the padding rows ride with the last video (no frames: they add nothing)
    out["lengths"] = ln
    if "num_subs" in b:                                       # host lists of the reference batch, for eager / oracle runs of the padded batch
        out["num_subs"] = list(b["num_subs"][:-1]) + [b["num_subs"][-1] + dT]
        last = list(b["sub_idx2frame_idx"][-1]) + [(b["num_subs"][-1] + i, []) for i in range(dT)]
        out["sub_idx2frame_idx"] = list(b["sub_idx2frame_idx"][:-1]) + [last]
    return out


def bucket_of(dims, buckets):
    """Index of the first bucket (cost order) that holds a batch of these batch_dims, or -1."""
    for i, b in enumerate(buckets):
        if (all(dims[k] <= b[k] for k in ("T", "max_vl", "max_sl", "Lf", "NF", "Lq", "frm")) and dims["B"] == b["B"]
                and dims["nq"] == b["nq"] and b["min_rows"] <= dims["rows"] <= b["rows"]):
            return i
    return -1


def pad_to_bucket(host_batch, buckets):
    """host batch -> (bucket index, batch padded to that bucket and tagged `_bucket`); (-1, the batch itself) when none holds it."""
    i = bucket_of(batch_dims(host_batch), buckets)
    if i < 0:
        return -1, host_batch
    out = pad_batch(host_batch, buckets[i])
    if out is host_batch:
        out = dict(host_batch)
    out["_bucket"] = i
    return i, out


class BucketPadder:
    """Picklable `collate_fn` for DataLoader workers: `BucketPadder(buckets, vcmr_collate)` collates the items like the
    reference (data/vcmr.py:140-159) and pads the batch to its bucket, so that the main process only pins and copies."""

    def __init__(self, buckets, collate=None):
        self.buckets, self.collate = [dict(b) for b in buckets], collate

    def __call__(self, items):
        batch = self.collate(items) if self.collate is not None else items
        return pad_to_bucket(batch, self.buckets)[1]


class BucketedBatchFeeder:
    """Feeds a hipGraph-replayed training step with RAGGED batches (VERDICT r5 "missing" #2; data/data.py:406-471 gives every
    batch its own T / max_vl / max_sl / frame count; data/loader.py:89-144 is what feeds them in the reference).

    buckets: a few shape dicts (batch_dims keys; `rows` = packed row capacity of the bucket, `min_rows` its lower bound),
    ordered by cost.  A host batch goes to the first bucket that holds it (pad_batch), each bucket has its own
    StaticBatchFeeder - static buffers, two staging sets, a captured commit graph, a static pack plan - created the first
    time a batch lands in it, and its static batch carries `_bucket` so that TrainStep keeps one pair of captured step graphs
    per bucket.  Use:
        feeder.prefetch(host_batch)                      # pads (or takes a batch already padded by feeder.pad in a worker)
        while ...:
            static = feeder.commit()                     # None -> no bucket held that batch: feeder.take_eager() is it
            loss = trainer.micro_step(static)
            feeder.prefetch(next_host_batch)
    """

    def __init__(self, buckets, device):
        if not buckets:
            raise ValueError("BucketedBatchFeeder: no buckets")
        self.buckets = [dict(b) for b in buckets]
        self.device = torch.device(device)
        self.feeders = [None] * len(buckets)
        self._stepped = [0] * len(buckets)       # commits handed out per bucket (the caller steps on each before the next prefetch)
        self._fifo = []                           # bucket index (or -1: eager) of the prefetched batches
        self._eager = []
        self.served = [0] * len(buckets)
        self.eager_served = 0

    # -- bucket geometry ----------------------------------------------------------------------------------------------------
    @staticmethod
    def derive_buckets(dims_list, n_buckets=3, row_quantum=512, slack=1.0):
        """Bucket shapes from a SAMPLE of batch_dims: every padded dimension is the sample's maximum (x `slack`, T to a
        multiple of 8, widths to a multiple of 4) - the maxima over ~500 subtitles / 32 videos barely move from batch to
        batch - and the PACKED ROW COUNT, which is what the cost of the six cross-modal layers follows, is cut into n_buckets
        capacities (multiples of `row_quantum` rows = whole GEMM row tiles) between the sample's extremes."""
        import math
        up = lambda v, q: int(math.ceil(v * slack / q) * q)      # noqa: E731
        mx = lambda k: max(d[k] for d in dims_list)              # noqa: E731
        base = dict(T=up(mx("T"), 8), max_vl=up(mx("max_vl"), 1), max_sl=up(mx("max_sl"), 4), NF=up(mx("NF"), 4), Lq=up(mx("Lq"), 1),
                    B=dims_list[0]["B"], nq=dims_list[0]["nq"], frm=up(mx("frm") * 1.25, 64))
        base["Lf"] = min(up(mx("Lf"), 4), base["max_vl"] + base["max_sl"])
        lo, hi = min(d["rows"] for d in dims_list), up(mx("rows"), row_quantum)
        caps = sorted({int(math.ceil((lo + (hi - lo) * (i + 1) / n_buckets) / row_quantum) * row_quantum) for i in range(n_buckets)})
        out = [dict(base, rows=c) for c in caps]
        for i, b in enumerate(out):        # a batch goes to the smallest capacity that holds it; the first bucket also takes smaller ones
            b["min_rows"] = out[i - 1]["rows"] + 1 if i else max(0, int(lo * 0.75))
        return out

    def bucket_of(self, dims):
        return bucket_of(dims, self.buckets)

    def pad(self, host_batch):
        """host batch -> (bucket index, padded batch): see pad_to_bucket / BucketPadder (the DataLoader-worker form)."""
        return pad_to_bucket(host_batch, self.buckets)

    # -- the feed loop ------------------------------------------------------------------------------------------------------
    def prefetch(self, host_batch):
        i = host_batch.get("_bucket")
        if i is None:
            i, host_batch = self.pad(host_batch)
        if i < 0:
            self._fifo.append(-1)
            self._eager.append(move_to_device({k: v for k, v in host_batch.items() if k != "lengths"}, self.device))
            return -1
        hb = pin_batch(host_batch)
        f = self.feeders[i]
        if f is None:
            b = self.buckets[i]
            f = self.feeders[i] = StaticBatchFeeder(hb, self.device, packed_rows=b["rows"], min_rows=b["min_rows"], frm_capacity=b["frm"])
            f.static["_bucket"] = i
            # the constructor committed this batch already: queue it again so that commit() hands it out in order
            f.prefetch(hb)
        else:
            if f._graph[0] is None and self._stepped[i] > 0 and not f._ready:
                f.capture()               # the step has run on this bucket's static batch: its derived tensors exist
            f.prefetch(hb)
        self._fifo.append(i)
        return i

    def commit(self):
        if not self._fifo:
            raise RuntimeError("BucketedBatchFeeder.commit: nothing prefetched")
        i = self._fifo.pop(0)
        if i < 0:
            self.eager_served += 1
            self._last_eager = self._eager.pop(0)
            return None
        self._stepped[i] += 1
        self.served[i] += 1
        return self.feeders[i].commit()

    def take_eager(self):
        """The batch of the commit() that returned None (no bucket held it): a plain device batch for an eager micro-step."""
        return self._last_eager

    @property
    def graphs(self):
        return sum(1 for f in self.feeders if f is not None)
