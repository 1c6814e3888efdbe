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
