"""The batch boundary: HERO's collate functions (host half) and their device half.

Host half - same names, inputs and outputs as the reference, so a reference DataLoader can take these as
`collate_fn` unchanged (pinned by tests/golden/case_collate.npz, which the reference's own code produced):
    video_item      VideoFeatSubTokDataset.__getitem__   data/data.py:345-403
    video_collate   video_collate + get_gather_index     data/data.py:406-471, 504-512
    query_collate   query_collate                        data/vcmr.py:120-137
    vcmr_collate    vcmr_collate                         data/vcmr.py:140-159
They are written over LENGTH arrays (one allocation per output, no pad_sequence / per-row python tensors) and also
return the handful of int32 length arrays (`batch["lengths"]`) from which the device half rebuilds every index tensor.

Device half - the reference builds every index tensor of a batch on the host in the DataLoader workers and
`collect_frame_outputs` (model/model.py:156-187) walks python lists per forward.  `DeviceCollate` takes the length
arrays and derives, with kernels, in place, into buffers of fixed capacity:

    f_gather_index, f_attn_masks  [T, out_size]          int64   out_size = max_i(frames_i + tokens_i), data.py:433
    c_attn_masks                  [B, NF]                int64
    f_v_feats                     [T, max_vl, D]         gathered from c_v_feats (optional: halves the PCIe bytes)
    frame_map = (offsets [B*NF + 1], entries [capacity], inverse [T * out_size])   int32, the CSR that
                hero_csr_gather_sum consumes (== hero_amd.model.model.build_frame_map, bit for bit)

Because the outputs keep their addresses, a hipGraph captured on one batch replays on the next batch of
the same SHAPE after `update()` + `hero_amd.functional.refresh_memo()` (tests/test_gpu_collate.py).
The packed (variable-length) formulation of ragged batches changes the GEMM row counts with the batch
and therefore stays an eager-mode feature.
"""
import numpy as np
import torch

from . import _lib as L


PAD_ID = 1          # RoBERTa <pad>, hard-coded in the reference's collates (data/data.py:423, data/vcmr.py:122)
MAX_POS = 511       # position ids are clamped here (data/data.py:429)


def video_item(v_feat, sub2frames, sub_tokens, sep=2, sub_ctx_len=0):
    """One video -> the 7-tuple of VideoFeatSubTokDataset.__getitem__ (data/data.py:345-403).
    v_feat [n_frames, D] float; sub2frames [(sub_idx, [frame_idx])]; sub_tokens[sub_idx] = token id list."""
    n_total, nf = len(sub2frames), v_feat.shape[0]
    ids, feats, masks = [], [], []
    for sub_idx, frames in sub2frames:
        toks = [sep]
        for j in range(sub_idx - sub_ctx_len, sub_idx + 1):      # the subtitle and its `sub_ctx_len` predecessors
            if 0 <= j < n_total:
                toks.extend(sub_tokens[j])
        keep = [f for f in frames if 0 <= f < nf]                 # frames past the (clipped) video are dropped
        if keep:
            feats.append(v_feat[torch.as_tensor(keep)])
            masks.append(torch.ones(len(keep) + len(toks), dtype=torch.long))
        else:                                                     # no frame: ONE zero frame slot, masked (data.py:380-382)
            feats.append(v_feat.new_zeros(1, v_feat.shape[1]))
            m = torch.ones(1 + len(toks), dtype=torch.long)
            m[0] = 0
            masks.append(m)
        ids.append(torch.tensor(toks, dtype=torch.long))
    return (ids, feats, masks, v_feat, torch.ones(nf, dtype=torch.long), n_total, sub2frames)


def st_ed_label(ts, max_idx, frame_interval=1.5):
    """[start s, end s] -> (first, last) frame index, VcmrDataset.get_st_ed_label (data/vcmr.py:103-117)."""
    import math
    st = min(math.floor(ts[0] / frame_interval), max_idx)
    return st, min(max(math.ceil(ts[1] / frame_interval) - 1, st + 1), max_idx)


def vcmr_item(video, vid, queries, cls_=0, frame_interval=1.5):
    """video: a video_item tuple; queries: [(token id list, [start s, end s])] -> VcmrDataset.__getitem__'s
    (video, vid, ((ids, mask, vid, target), ...)) (data/vcmr.py:73-98)."""
    last = video[3].shape[0] - 1
    out = []
    for toks, ts in queries:
        ids = torch.tensor([cls_] + list(toks), dtype=torch.long)
        out.append((ids, torch.ones_like(ids), vid, torch.tensor(st_ed_label(ts, last, frame_interval), dtype=torch.long)))
    return (video, vid, tuple(out))


def _pad_rows(rows, width, fill, dtype):
    out = torch.full((len(rows), width), fill, dtype=dtype)
    for r, t in enumerate(rows):
        out[r, :t.shape[0]] = t
    return out


def _pad_feats(rows, width):
    out = rows[0].new_zeros(len(rows), width, rows[0].shape[-1])
    for r, t in enumerate(rows):
        out[r, :t.shape[0]] = t
    return out


def video_collate(inputs):
    """list of video_item tuples -> the reference's batch dict (data/data.py:406-471), same keys / shapes / dtypes,
    plus `lengths` (int32 arrays, DeviceCollate's input).  Widths: f_attn_masks / f_gather_index have
    out_size = max_i(len(mask_i)) = max_i(frame slots_i + tokens_i) columns - NOT max_vl + max_sl."""
    ids = [t for it in inputs for t in it[0]]
    feats = [t for it in inputs for t in it[1]]
    masks = [t for it in inputs for t in it[2]]
    clips = [it[3] for it in inputs]
    num_subs = [it[5] for it in inputs]
    sub2frm = [it[6] for it in inputs]
    ntok = np.array([t.shape[0] for t in ids], dtype=np.int32)
    vlen = np.array([t.shape[0] for t in feats], dtype=np.int32)              # frame SLOTS (>= 1)
    max_sl, max_vl = int(ntok.max()), int(vlen.max())
    out_size = max(int(m.shape[0]) for m in masks)
    T = len(ids)
    gidx = torch.arange(out_size, dtype=torch.long).repeat(T, 1)              # get_gather_index, data.py:504-512
    for r in range(T):
        gidx[r, vlen[r]:vlen[r] + ntok[r]] = torch.arange(max_vl, max_vl + int(ntok[r]))
    nfr = np.array([c.shape[0] for c in clips], dtype=np.int32)
    NF = int(nfr.max())
    c_feats = _pad_feats(clips, NF)
    batch = {
        "f_sub_input_ids": _pad_rows(ids, max_sl, PAD_ID, torch.long),
        "f_sub_pos_ids": torch.arange(max_sl, dtype=torch.long).clamp_(max=MAX_POS).unsqueeze(0),
        "f_v_feats": _pad_feats(feats, max_vl),
        "f_v_pos_ids": torch.arange(max_vl, dtype=torch.long).unsqueeze(0),
        "f_attn_masks": _pad_rows(masks, out_size, 0, torch.long),
        "f_gather_index": gidx,
        "f_sub_input_attn_masks": (torch.arange(max_sl).unsqueeze(0) < torch.from_numpy(ntok).unsqueeze(1)).long(),
        "c_v_feats": c_feats,
        "c_pos_ids": torch.arange(NF, dtype=torch.long).repeat(len(clips), 1),
        "c_attn_masks": _pad_rows([it[4] for it in inputs], NF, 0, torch.long),
        "num_subs": num_subs,
        "sub_idx2frame_idx": sub2frm,
    }
    # frames per row as the MASK sees them (0 for the zero-slot rows): what DeviceCollate consumes
    nfrm_eff = np.array([int(m[0]) * int(v) for m, v in zip(masks, vlen)], dtype=np.int32)
    batch["lengths"] = _lengths(num_subs, sub2frm, nfrm_eff, ntok, nfr)
    return batch


def query_collate(query_input_ids, query_attn_mask, targets):
    """data/vcmr.py:120-137."""
    Lq = max(t.shape[0] for t in query_input_ids)
    return {"query_input_ids": _pad_rows(query_input_ids, Lq, PAD_ID, torch.long),
            "query_pos_ids": torch.arange(Lq, dtype=torch.long).unsqueeze(0),
            "query_attn_masks": _pad_rows(query_attn_mask, Lq, 0, torch.long),
            "targets": torch.stack(list(targets))}


def vcmr_collate(inputs):
    """list of (video_item tuple, vid, ((query ids, query mask, vid, target), ...)) -> batch (data/vcmr.py:140-159)."""
    vids = [it[1] for it in inputs]
    qs = [q for it in inputs for q in it[2]]
    batch = query_collate([q[0] for q in qs], [q[1] for q in qs], [q[3] for q in qs])
    batch.update(video_collate([it[0] for it in inputs]))
    batch["vids"] = vids
    where = {v: i for i, v in enumerate(vids)}
    batch["q_vidx"] = torch.tensor([where[q[2]] for q in qs], dtype=torch.long)
    return batch


def _lengths(num_subs, sub2frm, sub_nfrm, sub_ntok, n_frames):
    """Flat int32 description of a batch (host, ~1 KB).  sub_frm lists the frames as collect_frame_outputs will use
    them (the UNFILTERED lists of sub_idx2frame_idx, model/model.py:174-184); sub_nfrm the mask's frame counts."""
    frm, off = [], [0]
    for v, n in enumerate(num_subs):
        rows = sorted(sub2frm[v], key=lambda t: t[0])
        assert [sid for sid, _ in rows] == list(range(n)), "subtitle ids must be their row offsets inside the video"
        for _, frames in rows:
            frm.extend(frames)
            off.append(len(frm))
    i32 = lambda a: np.asarray(a, dtype=np.int32)      # noqa: E731
    return {"sub_nfrm": i32(sub_nfrm), "sub_ntok": i32(sub_ntok), "sub_frm_off": i32(off), "sub_frm": i32(frm if frm else [0]),
            "vid_sub_off": i32(np.concatenate([[0], np.cumsum(num_subs)])), "vid_nfrm": i32(n_frames)}


def lengths_from_lists(num_subs, sub_idx2frame_idx, sub_ntok, n_frames, sub_nfrm=None):
    """Length arrays from the host lists of an existing batch dict.  sub_ntok: tokens (incl. SEP) per subtitle
    row; sub_nfrm: frames per row as f_attn_masks sees them (default: the list lengths)."""
    if sub_nfrm is None:
        sub_nfrm = [len(fr) for v, n in enumerate(num_subs) for _, fr in sorted(sub_idx2frame_idx[v], key=lambda t: t[0])]
    return _lengths(num_subs, sub_idx2frame_idx, sub_nfrm, sub_ntok, n_frames)


class DeviceCollate:
    def __init__(self, T, max_vl, max_sl, B, NF, device, entry_capacity=None, out_size=None, vfeat_dim=None):
        """out_size: width of f_attn_masks / f_gather_index (the reference's: max over rows of frame slots +
        tokens; default max_vl + max_sl, its upper bound).  vfeat_dim: also own f_v_feats [T, max_vl, vfeat_dim]
        and fill it from c_v_feats in update() (hero_collate_gather_feats)."""
        self.T, self.max_vl, self.max_sl, self.B, self.NF = T, max_vl, max_sl, B, NF
        self.Lf = out_size or (max_vl + max_sl)
        if not 0 < self.Lf <= max_vl + max_sl:
            raise ValueError("DeviceCollate: out_size %d outside (0, max_vl + max_sl = %d]" % (self.Lf, max_vl + max_sl))
        self.device = torch.device(device)
        cap = entry_capacity or T * max_vl
        z = lambda *shape, dt=torch.int32: torch.zeros(*shape, dtype=dt, device=self.device)      # noqa: E731
        self.f_gather_index = z(T, self.Lf, dt=torch.int64)
        self.f_attn_masks = z(T, self.Lf, dt=torch.int64)
        self.c_attn_masks = z(B, NF, dt=torch.int64)
        self.f_v_feats = z(T, max_vl, vfeat_dim, dt=torch.float32) if vfeat_dim else None
        self.offsets = z(B * NF + 1)
        self.entries = z(max(cap, 1))
        self.inverse = z(T * self.Lf)
        self._counts = z(B * NF)
        self._row_vid = z(T)
        # the six length arrays live in ONE int32 buffer (slots padded to 4 elements): a feeder refreshes them with one copy
        sizes = {"sub_nfrm": T, "sub_ntok": T, "sub_frm_off": T + 1, "sub_frm": max(cap, 1), "vid_sub_off": B + 1, "vid_nfrm": B}
        self._in_off, off = {}, 0
        for k, n in sizes.items():
            self._in_off[k] = (off, n)
            off += (n + 3) & ~3
        self._in_flat = z(off)
        self._in = {k: self._in_flat[o:o + n] for k, (o, n) in self._in_off.items()}

    @classmethod
    def for_batch(cls, batch, device, entry_capacity=None, **kw):
        """Buffers sized for a (host) batch of video_collate.  entry_capacity: at least that much room for the frame lists."""
        T, max_vl = batch["f_v_feats"].shape[:2]
        B, NF = batch["c_attn_masks"].shape
        cap = max(int(batch["lengths"]["sub_frm"].shape[0]), T * max_vl) if "lengths" in batch else None
        if entry_capacity:
            cap = max(cap or 0, int(entry_capacity))
        return cls(T, max_vl, batch["f_sub_input_ids"].shape[1], B, NF, device, out_size=batch["f_attn_masks"].shape[1],
                   entry_capacity=cap, **kw)

    @property
    def frame_map(self):
        return self.offsets, self.entries, self.inverse

    def load_lengths(self, lengths, src_device=None):
        """Copy the length arrays into the device-side input slots (a few hundred bytes, stream-ordered).
        lengths: host arrays (video_collate's batch["lengths"]) or, with src_device=True, a dict of int32 DEVICE
        tensors of exactly the slot sizes (a staging copy made on another stream: StaticBatchFeeder)."""
        if src_device and torch.is_tensor(lengths):          # a flat staging copy of the whole input buffer (same layout)
            if lengths.shape != self._in_flat.shape:
                raise ValueError("DeviceCollate: flat length buffer of %d entries, expected %d" % (lengths.numel(), self._in_flat.numel()))
            self._in_flat.copy_(lengths, non_blocking=True)
            return self
        for k, dst in self._in.items():
            src = lengths[k] if src_device else torch.as_tensor(lengths[k], dtype=torch.int32)
            if src.numel() > dst.numel() or (k in ("sub_nfrm", "sub_ntok", "vid_nfrm") and src.numel() != dst.numel()):
                raise ValueError("DeviceCollate: %s has %d entries, the buffers were sized for %d" % (k, src.numel(), dst.numel()))
            dst[:src.numel()].copy_(src, non_blocking=True)
        return self

    def rebuild(self, c_v_feats=None):
        """Rebuild every index tensor in place from the loaded lengths: kernels only, no host data, no synchronisation
        (capturable in a hipGraph).  c_v_feats (device, [B, NF, vfeat_dim] fp32): also rebuild f_v_feats from it."""
        i, s, lib = self._in, L.stream(), L.lib()
        L.check(lib.hero_collate_subs(L.ptr(i["sub_nfrm"]), L.ptr(i["sub_ntok"]), L.ptr(self.f_gather_index),
                                      L.ptr(self.f_attn_masks), self.T, self.max_vl, self.Lf, s))
        if c_v_feats is not None:
            if self.f_v_feats is None or tuple(c_v_feats.shape) != (self.B, self.NF, self.f_v_feats.shape[2]) or c_v_feats.dtype != torch.float32:
                raise ValueError("DeviceCollate: c_v_feats must be fp32 [B, NF, vfeat_dim] and the collate built with vfeat_dim")
            L.check(lib.hero_collate_gather_feats(L.ptr(c_v_feats.contiguous()), L.ptr(self.f_v_feats), L.ptr(i["vid_sub_off"]),
                                                  L.ptr(i["vid_nfrm"]), L.ptr(i["sub_frm_off"]), L.ptr(i["sub_frm"]),
                                                  L.ptr(self._row_vid), self.T, self.max_vl, self.B, self.NF, self.f_v_feats.shape[2], s))
            torch._C._increment_version([self.f_v_feats])
        L.check(lib.hero_collate_clip_mask(L.ptr(i["vid_nfrm"]), L.ptr(self.c_attn_masks), self.B, self.NF, s))
        L.check(lib.hero_collate_frame_map(L.ptr(i["vid_sub_off"]), L.ptr(i["sub_frm_off"]), L.ptr(i["sub_frm"]), None,
                                           L.ptr(self._counts), None, None, self.B, self.NF, self.Lf, 0, s))
        self.offsets[:1].zero_()
        torch.cumsum(self._counts, 0, out=self.offsets[1:])          # exclusive scan of the counts (device op)
        self.inverse.fill_(-1)
        L.check(lib.hero_collate_frame_map(L.ptr(i["vid_sub_off"]), L.ptr(i["sub_frm_off"]), L.ptr(i["sub_frm"]),
                                           L.ptr(self.offsets), None, L.ptr(self.entries), L.ptr(self.inverse),
                                           self.B, self.NF, self.Lf, 1, s))
        # raw kernels wrote them: caches keyed on (address, version) must miss (the call takes an ITERABLE of tensors)
        torch._C._increment_version([self.f_gather_index, self.f_attn_masks, self.c_attn_masks])
        return self

    def update(self, lengths, c_v_feats=None):
        """load_lengths + rebuild: lengths are host arrays (video_collate's batch["lengths"] / lengths_from_lists)."""
        return self.load_lengths(lengths).rebuild(c_v_feats)

    def batch_entries(self):
        """The keys of the reference batch dict this object owns (+ `frame_map`, which replaces the host
        lists num_subs / sub_idx2frame_idx inside HierarchicalVlModel.collect_frame_outputs)."""
        out = {"f_gather_index": self.f_gather_index, "f_attn_masks": self.f_attn_masks,
               "c_attn_masks": self.c_attn_masks, "frame_map": self.frame_map}
        if self.f_v_feats is not None:
            out["f_v_feats"] = self.f_v_feats
        return out
