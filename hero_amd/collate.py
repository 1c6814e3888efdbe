"""Collate on the device.

The reference builds every index tensor of a batch on the host: `video_collate` / `get_gather_index`
(data/data.py:406-512) in the DataLoader workers, and `collect_frame_outputs` (model/model.py:156-187)
walks python lists per forward.  `DeviceCollate` takes the handful of LENGTH arrays that describe a
batch and derives, with kernels, in place, into buffers of fixed capacity:

    f_gather_index, f_attn_masks  [T, max_vl + max_sl]   int64
    c_attn_masks                  [B, NF]                int64
    frame_map = (offsets [B*NF + 1], entries [capacity], inverse [T * Lf])   int32, the CSR that
                hero_csr_gather_sum consumes (== hero_amd.model.model.build_frame_map, bit for bit)

Because the outputs keep their addresses, a hipGraph captured on one batch replays on the next batch of
the same SHAPE after `update()` + `hero_amd.functional.refresh_memo()` (tests/test_gpu_collate.py).
The packed (variable-length) formulation of ragged batches changes the GEMM row counts with the batch
and therefore stays an eager-mode feature.
"""
import numpy as np
import torch

from . import _lib as L


def lengths_from_lists(num_subs, sub_idx2frame_idx, sub_ntok, n_frames):
    """The loader-side description of a batch as flat int32 arrays (host, tiny): what a data loader
    would hand over instead of the index tensors.  sub_ntok: tokens (incl. SEP) per subtitle row."""
    nfrm, frm, off = [], [], [0]
    for v, n in enumerate(num_subs):
        rows = sorted(sub_idx2frame_idx[v], key=lambda t: t[0])
        assert [sid for sid, _ in rows] == list(range(n)), "subtitle ids must be their row offsets inside the video"
        for _, frames in rows:
            nfrm.append(len(frames))
            frm.extend(frames)
            off.append(len(frm))
    i32 = lambda a: np.asarray(a, dtype=np.int32)      # noqa: E731
    return {"sub_nfrm": i32(nfrm), "sub_ntok": i32(sub_ntok), "sub_frm_off": i32(off), "sub_frm": i32(frm if frm else [0]),
            "vid_sub_off": i32(np.concatenate([[0], np.cumsum(num_subs)])), "vid_nfrm": i32(n_frames)}


class DeviceCollate:
    def __init__(self, T, max_vl, max_sl, B, NF, device, entry_capacity=None):
        self.T, self.max_vl, self.max_sl, self.B, self.NF = T, max_vl, max_sl, B, NF
        self.Lf = max_vl + max_sl
        self.device = torch.device(device)
        cap = entry_capacity or T * max_vl
        z = lambda *shape, dt=torch.int32: torch.zeros(*shape, dtype=dt, device=self.device)      # noqa: E731
        self.f_gather_index = z(T, self.Lf, dt=torch.int64)
        self.f_attn_masks = z(T, self.Lf, dt=torch.int64)
        self.c_attn_masks = z(B, NF, dt=torch.int64)
        self.offsets = z(B * NF + 1)
        self.entries = z(max(cap, 1))
        self.inverse = z(T * self.Lf)
        self._counts = z(B * NF)
        self._in = {"sub_nfrm": z(T), "sub_ntok": z(T), "sub_frm_off": z(T + 1), "sub_frm": z(max(cap, 1)),
                    "vid_sub_off": z(B + 1), "vid_nfrm": z(B)}

    @property
    def frame_map(self):
        return self.offsets, self.entries, self.inverse

    def update(self, lengths):
        """lengths: dict of int32 arrays / tensors (see lengths_from_lists).  Copies them to the device
        (a few hundred bytes) and rebuilds every index tensor in place, stream-ordered, no synchronisation."""
        for k, dst in self._in.items():
            src = torch.as_tensor(lengths[k], dtype=torch.int32)
            if src.numel() > dst.numel() or (k in ("sub_nfrm", "sub_ntok", "vid_nfrm") and src.numel() != dst.numel()):
                raise ValueError("DeviceCollate: %s has %d entries, the buffers were sized for %d" % (k, src.numel(), dst.numel()))
            dst[:src.numel()].copy_(src, non_blocking=True)
        i, s, lib = self._in, L.stream(), L.lib()
        L.check(lib.hero_collate_subs(L.ptr(i["sub_nfrm"]), L.ptr(i["sub_ntok"]), L.ptr(self.f_gather_index),
                                      L.ptr(self.f_attn_masks), self.T, self.max_vl, self.max_sl, s))
        L.check(lib.hero_collate_clip_mask(L.ptr(i["vid_nfrm"]), L.ptr(self.c_attn_masks), self.B, self.NF, s))
        L.check(lib.hero_collate_frame_map(L.ptr(i["vid_sub_off"]), L.ptr(i["sub_frm_off"]), L.ptr(i["sub_frm"]), None,
                                           L.ptr(self._counts), None, None, self.B, self.NF, self.Lf, 0, s))
        self.offsets[0] = 0
        torch.cumsum(self._counts, 0, out=self.offsets[1:])          # exclusive scan of the counts (device op)
        self.inverse.fill_(-1)
        L.check(lib.hero_collate_frame_map(L.ptr(i["vid_sub_off"]), L.ptr(i["sub_frm_off"]), L.ptr(i["sub_frm"]),
                                           L.ptr(self.offsets), None, L.ptr(self.entries), L.ptr(self.inverse),
                                           self.B, self.NF, self.Lf, 1, s))
        for t in (self.f_gather_index, self.f_attn_masks, self.c_attn_masks):
            torch._C._increment_version(t)     # raw kernels wrote them: caches keyed on (address, version) must miss
        return self

    def batch_entries(self):
        """The keys of the reference batch dict this object owns (+ `frame_map`, which replaces the host
        lists num_subs / sub_idx2frame_idx inside HierarchicalVlModel.collect_frame_outputs)."""
        return {"f_gather_index": self.f_gather_index, "f_attn_masks": self.f_attn_masks,
                "c_attn_masks": self.c_attn_masks, "frame_map": self.frame_map}
