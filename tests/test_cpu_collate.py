"""The batch boundary against the reference's OWN data code.

tests/golden/case_collate.npz was produced by importing the reference's VideoFeatSubTokDataset.__getitem__,
video_collate, get_gather_index, VcmrDataset.__getitem__ and vcmr_collate (tests/golden/make_golden_collate.py).
hero_amd.collate's host half must reproduce every tensor and list of those batches bit for bit, including
  * f_attn_masks / f_gather_index NARROWER than max_vl + max_sl (data/data.py:433-436),
  * a subtitle without frames (one zero slot, mask bit 0), frames no subtitle covers, a video clipped by max_clip_len,
  * position ids clamped at 511,
and the oracle on the narrow batch must give the reference model's outputs."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN

Z = np.load(os.path.join(GOLDEN, "case_collate.npz"))
CASES = json.loads(str(Z["__cases__"]))


def ref_batch(case):
    out = {}
    for k in Z.files:
        if k.startswith(case + ".out."):
            a = Z[k]
            out[k[len(case) + 5:]] = json.loads(str(a)) if a.dtype.kind == "U" else torch.from_numpy(a)
    return out


def rebuild(case):
    """The raw per-video inputs of a case -> hero_amd.collate.vcmr_collate."""
    from hero_amd import collate as C
    desc = json.loads(str(Z[case + ".desc"]))
    want = ref_batch(case)
    items = []
    by_vid = {v["vid"]: (i, v) for i, v in enumerate(desc["videos"])}
    for qid in desc["query_order"]:                     # sampled_by_q: one item per query, in query order
        vi, v = by_vid["v" + qid[1:3]]
        feat = torch.from_numpy(Z["%s.feat.%s" % (case, v["vid"])])[:desc["max_clip_len"]]
        # sub2frames as SubTokLmdb.compute_sub2frames left them = what the reference hands through the batch
        s2f = [(sid, list(fr)) for sid, fr in want["sub_idx2frame_idx"][vi]]
        video = C.video_item(feat, s2f, v["sub_tokens"], sep=2)
        q = v["queries"][int(qid.split("_")[1])]
        items.append(C.vcmr_item(video, v["vid"], [(q["tokens"], q["ts"])], cls_=0, frame_interval=desc["frame_interval"]))
    return C.vcmr_collate(items), want


@pytest.mark.parametrize("case", CASES)
def test_host_collate_equals_reference_collate(case):
    got, want = rebuild(case)
    for k, w in want.items():
        g = got[k]
        if torch.is_tensor(w):
            assert g.dtype == w.dtype and g.shape == w.shape, (k, g.shape, w.shape, g.dtype, w.dtype)
            assert torch.equal(g, w), k
        else:
            assert json.loads(json.dumps(g)) == w, k
    assert set(got) - set(want) == {"lengths"}


def test_fixture_has_the_narrow_and_clamped_cases():
    n = ref_batch("narrow")
    assert n["f_attn_masks"].shape[1] < n["f_v_feats"].shape[1] + n["f_sub_input_ids"].shape[1]
    assert (n["f_attn_masks"][:, 0] == 0).any()                       # a zero-frame subtitle
    assert int(ref_batch("clamp")["f_sub_pos_ids"].max()) == 511 and ref_batch("clamp")["f_sub_pos_ids"].shape[1] > 512
    narrower = sum(ref_batch(c)["f_attn_masks"].shape[1] < ref_batch(c)["f_v_feats"].shape[1] + ref_batch(c)["f_sub_input_ids"].shape[1]
                   for c in CASES)
    assert narrower >= 4


def test_oracle_on_the_narrow_reference_batch():
    from oracle import hero_oracle as O
    P, cfgj, vfeat, max_frm = O.load_npz_model(os.path.join(GOLDEN, "tiny_model.npz"))
    cfg = O.cfg_from_json(cfgj)
    b = ref_batch("narrow")
    f_seq = O.f_encoder_repr(b, P, cfg)
    m = b["f_attn_masks"].bool()
    want = torch.from_numpy(Z["narrow.model.f_seq"])
    assert (f_seq[m] - want[m]).abs().max() < 1e-5
    rep = O.forward_repr(b, P, cfg)
    want = torch.from_numpy(Z["narrow.model.repr"])
    cm = b["c_attn_masks"].bool()
    assert (rep[cm] - want[cm]).abs().max() < 1e-5
    losses = O.vsm_losses(b, P, cfg)
    for got, key in zip(losses, ("loss_st_ed", "loss_neg_ctx", "loss_neg_q")):
        np.testing.assert_allclose(got.detach().numpy(), Z["narrow.model." + key], rtol=1e-4, atol=1e-6)


def test_lengths_describe_the_batch():
    """The int32 length arrays of video_collate regenerate its masks (host restatement of collate.hip)."""
    got, _ = rebuild("narrow")
    ln = got["lengths"]
    W = got["f_attn_masks"].shape[1]
    for r in range(got["f_attn_masks"].shape[0]):
        nf, nt = int(ln["sub_nfrm"][r]), int(ln["sub_ntok"][r])
        row = [1 if (p < nf + nt if nf else 1 <= p < 1 + nt) else 0 for p in range(W)]
        assert got["f_attn_masks"][r].tolist() == row
    assert ln["vid_nfrm"].tolist() == got["c_attn_masks"].sum(1).tolist()


def test_oracle_lse_hard_negative_branch_against_reference():
    """ranking_loss_type = 'lse' with hard-negative weighting (model/pretrain.py:203-292, 340-362): the oracle's
    losses AND gradients on the narrow reference batch equal the reference's."""
    from oracle import hero_oracle as O
    P, cfgj, vfeat, max_frm = O.load_npz_model(os.path.join(GOLDEN, "tiny_model.npz"))
    cfg = O.cfg_from_json(cfgj)
    b = ref_batch("narrow")
    Pq = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("pad")) for k, v in P.items()}
    losses = O.vsm_losses(b, Pq, cfg, hard=(1, 10.0), ranking="lse")
    for got, key in zip(losses, ("loss_st_ed", "loss_neg_ctx", "loss_neg_q")):
        np.testing.assert_allclose(got.detach().numpy(), Z["narrow.lse." + key], rtol=1e-5, atol=1e-6)
    sum(losses).backward()
    names = [k[len("narrow.lse.grad."):] for k in Z.files if k.startswith("narrow.lse.grad.")]
    assert len(names) == 4
    for n in names:
        want = torch.from_numpy(Z["narrow.lse.grad." + n])
        assert (Pq[n].grad - want).abs().max() <= 1e-5 * want.abs().max() + 1e-7, n


def test_reference_gather_index_names_a_source_row_first_from_its_valid_position():
    """Round 5: the embedding interleave's backward is ONE gather through the first-occurrence map of f_gather_index
    (hero_inverse_first, functional.GatherRowsFn(valid=f_attn_masks)) - exact only because, in every index tensor the REFERENCE's
    get_gather_index produces (data/data.py:504-512; ten batches of case_collate.npz incl. the narrow ones, a zero-frame subtitle
    and the 511 clamp), a valid position is always the FIRST reference to its source row: the repeats sit in the padded
    identity tail, whose gradients are exactly zero."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "case_collate.npz"), allow_pickle=True)
    keys = [k for k in z.files if k.endswith("f_gather_index")]
    assert len(keys) >= 10
    checked = repeats = 0
    for k in keys:
        gi, m = z[k], z[k.replace("f_gather_index", "f_attn_masks")]
        assert gi.shape == m.shape
        for row_g, row_m in zip(gi.tolist(), m.tolist()):
            first_pos = {}
            for j, v in enumerate(row_g):
                first_pos.setdefault(v, j)
            repeats += len(row_g) - len(first_pos)
            for j, (v, ok) in enumerate(zip(row_g, row_m)):
                if ok:
                    assert first_pos[v] == j, (k, j, v)
                    checked += 1
    assert checked > 2000 and repeats > 0           # the fixtures do contain repeated references (all of them padded)


def test_host_segment_order_is_a_stable_sort_with_dropped_rows_last():
    """hero_amd.functional.host_segment_order (what StaticBatchFeeder.prefetch computes instead of the device's
    hero_segment_sort): rows by (id, row), ids < 0 or == skip behind every real destination, stable."""
    import numpy as np
    import torch
    from hero_amd import functional as HF
    g = torch.Generator().manual_seed(3)
    for n, vocab, skip in ((9600, 50272, 1), (33, 5, 0), (100, 7, -1)):
        ids = torch.randint(0, vocab, (n,), generator=g)
        ids[::9] = -1
        key = torch.where((ids < 0) | (ids == skip), torch.full_like(ids, 1 << 40), ids)
        want = torch.sort(key, stable=True).indices.to(torch.int32)
        got = torch.from_numpy(HF.host_segment_order(ids.numpy(), skip))
        assert torch.equal(got, want)


# ---- round 6: bucket padding and the static pack plan (host halves of hero_amd.loader.BucketedBatchFeeder) -------------------
def _ragged_host_batches(n, seed0=40):
    from hero_amd import synth
    out = []
    for s in range(n):
        gen = torch.Generator().manual_seed(seed0 + s)
        ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))     # noqa: E731
        subs, n_frames = [], []
        for v in range(2):
            nf, cur, f0 = ri(18, 32), [], 0
            for s_ in range(ri(4, 8)):
                fr = list(range(f0, min(f0 + ri(0, 4), nf)))
                f0 += len(fr)
                cur.append((fr, ri(2, 9)))
            subs.append(cur)
            n_frames.append(nf)
        b = synth.video_batch(subs, n_frames, 96, 128, gen)                 # the tiny golden model: vfeat 96, vocabulary 160
        b.update(synth.query_batch(2, [ri(4, 12), ri(4, 12)], 128, gen))
        b["targets"] = torch.tensor([[1, 3], [2, ri(3, 9)]])
        b["q_vidx"] = torch.arange(2)
        out.append(b)
    return out


def test_bucket_padding_is_a_legal_reference_batch():
    """loader.pad_batch / derive_buckets / BucketPadder on the host: a padded batch has the bucket's shapes, the original
    contents, the reference's padding conventions (data/data.py:406-471: id 1, mask 0, zero features, gather index of the
    new max_vl), lengths DeviceCollate can consume - and the ORACLE gives the original batch's ranking losses on it (padded
    positions are masked out of every score) and a start / end loss within the convolution's edge effect."""
    import pickle
    from hero_amd.collate import vcmr_collate
    from hero_amd.loader import BucketPadder, BucketedBatchFeeder, batch_dims, bucket_of, pad_batch, pad_to_bucket
    from oracle import hero_oracle as O
    P, cfgj, _, _ = O.load_npz_model(os.path.join(GOLDEN, "tiny_model.npz"))
    cfg = O.cfg_from_json(cfgj)
    host = _ragged_host_batches(6)
    dims = [batch_dims(h) for h in host]
    buckets = BucketedBatchFeeder.derive_buckets(dims, n_buckets=3, row_quantum=16)
    assert 2 <= len(buckets) <= 3 and all(b["rows"] % 16 == 0 for b in buckets)
    assert all(a["rows"] < b["rows"] and b["min_rows"] == a["rows"] + 1 for a, b in zip(buckets, buckets[1:]))
    pickle.loads(pickle.dumps(BucketPadder(buckets, vcmr_collate)))                 # a DataLoader worker can carry it
    for h, d in zip(host, dims):
        i = bucket_of(d, buckets)
        assert i >= 0 and buckets[i]["min_rows"] <= d["rows"] <= buckets[i]["rows"]
        assert i == 0 or d["rows"] > buckets[i - 1]["rows"]                           # the smallest capacity that holds it
        j, p = pad_to_bucket(h, buckets)
        assert j == i == p["_bucket"]
        b = buckets[i]
        pd = batch_dims(p)
        assert all(pd[k] == b[k] for k in ("T", "max_vl", "max_sl", "Lf", "NF", "Lq")) and pd["rows"] == d["rows"]
        T, Lf = d["T"], d["Lf"]
        assert torch.equal(p["f_sub_input_ids"][:T, :d["max_sl"]], h["f_sub_input_ids"]) and int((p["f_sub_input_ids"][T:] != 1).sum()) == 0
        assert torch.equal(p["f_attn_masks"][:T, :Lf], h["f_attn_masks"]) and int(p["f_attn_masks"][T:].sum()) == 0
        assert torch.equal(p["c_v_feats"][:, :d["NF"]], h["c_v_feats"]) and float(p["c_v_feats"][:, d["NF"]:].abs().sum()) == 0
        assert torch.equal(p["query_attn_masks"][:, :d["Lq"]], h["query_attn_masks"])
        # the gather index is get_gather_index (data/data.py:504-512) for the bucket's max_vl: frames first, then the tokens
        for r in range(T):
            n_valid = int(h["f_attn_masks"][r].sum())
            src_h = h["f_gather_index"][r][h["f_attn_masks"][r] == 1]
            src_p = p["f_gather_index"][r][p["f_attn_masks"][r] == 1]
            assert n_valid == len(src_p)
            assert torch.equal(torch.where(src_h >= d["max_vl"], src_h - d["max_vl"] + b["max_vl"], src_h), src_p)
        assert len(p["lengths"]["sub_nfrm"]) == b["T"] and int(p["lengths"]["vid_sub_off"][-1]) == b["T"]
        assert p["num_subs"][-1] == h["num_subs"][-1] + b["T"] - T
        want = [float(x) for x in O.vsm_losses(h, P, cfg)]
        got = [float(x) for x in O.vsm_losses(p, P, cfg)]
        np.testing.assert_allclose(got[1:], want[1:], rtol=1e-5, atol=1e-6)
        assert abs(got[0] - want[0]) < 0.05 * abs(want[0])
    assert pad_batch(p, buckets[p["_bucket"]]) is p                                   # already at the bucket's shape: untouched
    big = dict(host[0])
    big["query_input_ids"] = torch.nn.functional.pad(host[0]["query_input_ids"], (0, 40), value=1)
    big["query_attn_masks"] = torch.nn.functional.pad(host[0]["query_attn_masks"], (0, 40))
    assert pad_to_bucket(big, buckets)[0] == -1                                       # no bucket holds it: the caller runs it eagerly


def test_static_pack_plan_is_the_dynamic_plan_plus_pad_rows():
    """BertEncoder.fill_static_plan (numpy, what StaticBatchFeeder.prefetch runs per batch) builds the maps BertEncoder._pack_plan
    derives from device masks - valid positions in row-major order, the groups back to back - and appends the pad rows as pad
    sequences of <= 32 rows up to the plan's fixed capacity; the sequence count, every buffer size and the attention length class
    depend on the LAYOUT only."""
    from hero_amd.model.layers import BertEncoder as BE
    host = _ragged_host_batches(4)
    from hero_amd.loader import BucketedBatchFeeder, batch_dims, pad_to_bucket
    buckets = BucketedBatchFeeder.derive_buckets([batch_dims(h) for h in host], n_buckets=1, row_quantum=16)
    lay = None
    for h in host:
        _, p = pad_to_bucket(h, buckets)
        masks = [p["f_attn_masks"], p["query_attn_masks"]]
        groups = tuple(tuple(m.shape) for m in masks)
        new = BE.static_plan_layout(groups, buckets[0]["rows"], buckets[0]["min_rows"])
        assert lay is None or {k: v for k, v in new.items()} == lay                   # one layout for every batch of the bucket
        lay = new
        flat = np.full(lay["size"], 12345, dtype=np.int32)
        valid = BE.fill_static_plan(flat, lay, [m.numpy() for m in masks])
        assert valid == batch_dims(h)["rows"]
        sec = lambda n: flat[lay[n][0]:lay[n][0] + lay[n][1]]      # noqa: E731
        fl = torch.cat([(m != 0).reshape(-1) for m in masks])
        gather = torch.nonzero(fl).reshape(-1).to(torch.int32).numpy()
        assert np.array_equal(sec("gather")[:valid], gather) and (sec("gather")[valid:] == -1).all()
        inv = np.full(fl.numel(), -1, np.int32)
        inv[gather] = np.arange(valid)
        assert np.array_equal(sec("inverse"), inv)
        n0 = masks[0].numel()
        assert np.array_equal(sec("back0")[:valid], np.where(gather < n0, gather, -1)) and (sec("back0")[valid:] == -1).all()
        assert np.array_equal(sec("back1")[:valid], np.where(gather >= n0, gather - n0, -1))
        off = sec("off")
        counts = torch.cat([(m != 0).sum(1) for m in masks]).numpy()
        assert off[0] == 0 and np.array_equal(np.diff(off[:lay["n_real"] + 1]), counts)
        pad = np.diff(off[lay["n_real"]:])
        assert (pad >= 0).all() and pad.max() <= BE.PAD_CHUNK and int(off[-1]) == lay["rows_cap"] and len(off) == lay["n_seq"] + 1
    with pytest.raises(ValueError):
        BE.fill_static_plan(flat, dict(lay, rows_cap=valid - 1), [m.numpy() for m in masks])      # more valid rows than the capacity


def test_refresh_memo_redoes_chained_entries_in_dependency_order():
    """functional.refresh_memo (what a feeder's commit runs after new data landed in the static batch): entries derived from
    OTHER entries are redone after them whatever the order of the memo table - round 6 found a (query, video) pair mask
    (a builder entry) whose fp32 cast (an entry of its own) stayed one batch behind; restamp_memo moves re-keyed entries to
    the end of the table, so table order is not dependency order either."""
    from hero_amd import functional as HF
    HF.reset_caches()
    a = torch.arange(6, dtype=torch.int64)
    other = torch.ones(3, dtype=torch.int64)
    m1 = HF.memo("t1", (a,), lambda: a * 2)
    m2 = HF.memo("t2", (m1,), lambda: m1 + 1)
    m3 = HF.memo("t3", (m2, a), lambda: m2 * a)
    mo = HF.memo("to", (other,), lambda: other * 7)
    k1 = [k for k in HF._MEMO if k[0] == "t1"]
    HF.restamp_memo(k1)                                      # t1 now sits BEHIND its dependants in the table
    assert [k[0] for k in HF._MEMO][-1] == "t1"
    a.copy_(torch.tensor([5, 4, 3, 2, 1, 0]))
    other.fill_(2)
    HF.refresh_memo(sources=[a])
    assert torch.equal(m1, a * 2) and torch.equal(m2, a * 2 + 1) and torch.equal(m3, (a * 2 + 1) * a)
    assert torch.equal(mo, torch.full((3,), 7))              # not derived from `a`: left alone
    assert sorted(k[0] for k in HF.last_refreshed_keys()) == ["t1", "t3"]        # the DIRECT dependants (what a feeder restamps)
    HF.refresh_memo()
    assert torch.equal(mo, torch.full((3,), 14))
    HF.reset_caches()
