#!/usr/bin/env python3
"""Golden vectors for the BATCH boundary, produced by the reference's own data code.

Container-only (needs /root/reference).  The reference's dataset / collate path for the TVR fine-tuning step
    SubTokLmdb.compute_sub2frames            data/data.py:217-250
    VideoFeatSubTokDataset.__getitem__       data/data.py:345-403
    video_collate / get_gather_index         data/data.py:406-471, 504-512
    VcmrDataset.__getitem__ / vcmr_collate   data/vcmr.py:73-159
is IMPORTED and RUN here on in-memory stand-ins for the two LMDB readers (the on-disk stores are the only thing
replaced: a dict of token lists for `SubTokLmdb.db`, a dict of feature tensors behind `VideoFeatLmdb.__getitem__`).
Packages the image lacks are stubbed before the import: lmdb, lz4.frame, msgpack_numpy, toolz / cytoolz (two
one-line itertools equivalents), horovod.torch (size 1), apex FusedLayerNorm (-> nn.LayerNorm).

Writes tests/golden/case_collate.npz:
  * per case `<c>`: the raw per-video description (`<c>.desc`, JSON: token ids per subtitle, frame lists, frame
    counts, queries, time stamps) + the feature tensors, and EVERY tensor / list of the reference batch (`<c>.out.*`);
  * for the `narrow` case (f_attn_masks narrower than max_vl + max_sl, a zero-frame subtitle, frames no subtitle
    covers, a subtitle cut by max_clip_len) also the reference MODEL's outputs on that exact batch with the tiny
    weights of tiny_model.npz: f_seq, pre_temporal, repr, the 'txt' stream, and the three VCMR losses.

Run:  python tests/golden/make_golden_collate.py
"""
import itertools
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("HERO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_stubs  # noqa: E402  (apex + horovod stubs)


def install_data_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("lmdb", open=None)
    lz4 = mod("lz4")
    lz4.frame = mod("lz4.frame", compress=lambda b: b, decompress=lambda b: b)
    mod("msgpack_numpy", patch=lambda: None)
    tz = mod("toolz")
    tz.sandbox = mod("toolz.sandbox", unzip=lambda seq: zip(*seq))
    mod("cytoolz", concat=itertools.chain.from_iterable)
    hvd = sys.modules["horovod.torch"]
    hvd.local_size = lambda: 1


def build_case(D, spec, vfeat, max_clip_len, frame_interval=1.5, seed=0, vocab=160):
    """spec: list of videos = dict(nframe_db, subs=[(sub_idx, [frames], n_words)], unmatched=[...],
    queries=[(n_words, (t0, t1))]).  Returns (reference batch, description of the raw inputs)."""
    from data.data import SubTokLmdb, VideoFeatLmdb, VideoFeatSubTokDataset, QueryTokLmdb
    from data.vcmr import VcmrDataset, vcmr_collate
    g = torch.Generator().manual_seed(seed)

    class FeatDb(VideoFeatLmdb):                       # the LMDB reader replaced by a dict of tensors
        def __init__(self, feats, max_clip_len):
            self.feats, self.max_clip_len, self.frame_interval = feats, max_clip_len, frame_interval
            self.name2nframe = {k: v.shape[0] for k, v in feats.items()}

        def __getitem__(self, name):                   # data/data.py:107-119 without the lmdb read
            n = min(self.name2nframe[name], self.max_clip_len)
            return self.feats[name][:n].float()

        def __del__(self):
            pass

    class SubDb(SubTokLmdb):                           # keeps the real compute_sub2frames
        def __init__(self, db, max_clip_len):
            self.db, self.max_clip_len = db, max_clip_len
            self.sep, self.cls_ = 2, 0
            self.id2len = {k: v["nframe"] for k, v in db.items()}
            self.vid2dur, self.vid2idx = {}, {}
            self.vid_sub2frame, self.vid2vonly_frames = self.compute_sub2frames()

        def __getitem__(self, k):
            return self.db[k]

        def __del__(self):
            pass

    class QDb(QueryTokLmdb):
        def __init__(self, db, q2v):
            self.db, self.query2video, self.cls_ = db, q2v, 0
            self.video2query, self.query_data = {}, {}
            self.id2len = {k: len(v["input_ids"]) for k, v in db.items()}

        def __getitem__(self, k):
            return self.db[k]

        def __del__(self):
            pass

    feats, subdb, qdb, q2v, desc = {}, {}, {}, {}, []
    for v, s in enumerate(spec):
        vid = "v%02d" % v
        feats[vid] = torch.randn(s["nframe_db"], vfeat, generator=g)
        toks = {}
        for sub_idx, frames, nw in s["subs"]:
            toks[sub_idx] = torch.randint(3, vocab, (nw,), generator=g).tolist()
        n_sub_total = max(toks) + 1
        subdb[vid] = {"input_ids": [toks.get(i, []) for i in range(n_sub_total)],
                      "unique_sub2frames": [(si, list(fr)) for si, fr, _ in s["subs"]],
                      "unmatched_frames": list(s.get("unmatched", [])), "nframe": s["nframe_db"]}
        for qi, (nw, ts) in enumerate(s["queries"]):
            qid = "q%02d_%d" % (v, qi)
            qdb[qid] = {"input_ids": torch.randint(3, vocab, (nw,), generator=g).tolist(), "target": list(ts)}
            q2v[qid] = vid
        desc.append({"vid": vid, "nframe_db": s["nframe_db"], "subs": subdb[vid]["unique_sub2frames"],
                     "sub_tokens": subdb[vid]["input_ids"],
                     "queries": [{"tokens": qdb["q%02d_%d" % (v, qi)]["input_ids"], "ts": list(ts)}
                                 for qi, (nw, ts) in enumerate(s["queries"])]})
    img_db = FeatDb(feats, max_clip_len)
    txt_db = SubDb(subdb, max_clip_len)
    video_db = VideoFeatSubTokDataset(txt_db, img_db, max_txt_len=-1, sub_ctx_len=0)
    query_db = QDb(qdb, q2v)
    ds = VcmrDataset(sorted(feats), video_db, query_db, sampled_by_q=True)
    batch = vcmr_collate([ds[i] for i in range(len(ds))])
    raw = {"videos": desc, "max_clip_len": max_clip_len, "frame_interval": frame_interval,
           "query_order": ds.qids}
    return batch, raw, feats


def pack(prefix, batch, raw, feats):
    out = {prefix + ".desc": np.array(json.dumps(raw))}
    for k, v in feats.items():
        out["%s.feat.%s" % (prefix, k)] = v.numpy()
    for k, v in batch.items():
        out["%s.out.%s" % (prefix, k)] = v.numpy() if torch.is_tensor(v) else np.array(json.dumps(v))
    return out


def random_spec(g, n_videos, max_clip_len):
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))     # noqa: E731
    spec = []
    for _ in range(n_videos):
        nf = ri(6, 40)
        subs, f0 = [], 0
        for si in range(ri(2, 9)):
            k = ri(0, 6)
            if ri(0, 4) == 0:
                f0 += ri(1, 3)                     # frames no subtitle covers
            fr = list(range(f0, f0 + k))           # may run past nf / max_clip_len: the dataset filters them
            f0 += k
            subs.append((si, fr, ri(1, 18)))
        spec.append(dict(nframe_db=nf, subs=subs, unmatched=[], queries=[(ri(3, 12), (ri(0, 20) * 1.0, ri(21, 60) * 1.0))]))
    return spec


def main():
    install_stubs()
    install_data_stubs()
    sys.path.insert(0, REF)
    D = None
    out = {}

    # ---- narrow: out_size = max_i(frames_i + tokens_i) < max_vl + max_sl (data/data.py:433-436) -------------
    narrow = [
        dict(nframe_db=12, unmatched=[9, 10],
             subs=[(0, [0, 1, 2, 3, 4], 2),         # 5 frames + 3 tokens (SEP + 2) = 8 columns: the widest row
                   (1, [], 4),                      # zero-frame subtitle: mask [0, 1, 1, 1, 1, 1] (data.py:380-382)
                   (2, [5], 5),                     # 1 + 6 = 7
                   (3, [6, 7, 8], 3)],
             queries=[(6, (1.6, 7.4))]),
        dict(nframe_db=20,                          # cut to max_clip_len = 14 by the feature reader
             subs=[(0, [0, 1], 4), (1, [2, 3, 4], 1), (2, [12, 13, 14, 15], 3),   # frames 14, 15 dropped
                   (3, [16, 17], 2)],               # never reached: compute_sub2frames stops at the cut
             queries=[(4, (0.0, 3.1))]),
        dict(nframe_db=5, subs=[(0, [1, 2, 3], 3)], queries=[(9, (2.0, 30.0))]),
    ]
    batch, raw, feats = build_case(D, narrow, vfeat=96, max_clip_len=14, seed=11)
    W = batch["f_attn_masks"].shape[1]
    assert W < batch["f_v_feats"].shape[1] + batch["f_sub_input_ids"].shape[1], "case must be narrow"
    out.update(pack("narrow", batch, raw, feats))

    # the reference MODEL on that batch (tiny weights of tiny_model.npz)
    from model.vcmr import HeroForVcmr              # noqa: reference import
    z = np.load(os.path.join(HERE, "tiny_model.npz"))
    sd = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("__")}
    model = HeroForVcmr.from_pretrained(
        os.path.join(HERE, "tiny_config.json"), state_dict=sd, vfeat_dim=int(z["__vfeat__"]),
        max_frm_seq_len=int(z["__max_frm__"]), lw_neg_ctx=8.0, lw_neg_q=8.0, lw_st_ed=0.01,
        ranking_loss_type="hinge", use_hard_negative=False, hard_pool_size=20, margin=0.1, use_all_neg=True,
        drop_svmr_prob=0.0)
    model.eval()
    import collections
    with torch.no_grad():
        fr = model.v_encoder.f_encoder(batch, "repr")[0]
        pre = model.v_encoder.forward_repr(collections.defaultdict(lambda: None, batch), encode_clip=False)
        rep = model.v_encoder(batch, "repr")
        txt = model.v_encoder.f_encoder({"input_ids": batch["query_input_ids"], "pos_ids": batch["query_pos_ids"],
                                         "attn_masks": batch["query_attn_masks"]}, "txt")[0]
    model.train()                                   # the training branch of the loss (reduction = mean), dropout off
    for m_ in model.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
    with torch.no_grad():
        l_st, l_ctx, l_q = model(batch, task="tvr", compute_loss=True)
    out.update({"narrow.model.f_seq": fr.numpy(), "narrow.model.pre_temporal": pre.numpy(),
                "narrow.model.repr": rep.numpy(), "narrow.model.txt": txt.numpy(),
                "narrow.model.loss_st_ed": l_st.numpy(), "narrow.model.loss_neg_ctx": l_ctx.numpy(),
                "narrow.model.loss_neg_q": l_q.numpy()})
    # the smooth ranking loss with hard-negative weighting (ranking_loss_type = "lse", model/pretrain.py:203-292, 340-362):
    # losses and gradients of the same batch - pins the oracle's lse / hard-negative branch
    lse = HeroForVcmr.from_pretrained(
        os.path.join(HERE, "tiny_config.json"), state_dict=sd, vfeat_dim=int(z["__vfeat__"]),
        max_frm_seq_len=int(z["__max_frm__"]), lw_neg_ctx=8.0, lw_neg_q=8.0, lw_st_ed=0.01,
        ranking_loss_type="lse", use_hard_negative=True, hard_pool_size=1, hard_neg_weight=10, margin=0.1,
        use_all_neg=True, drop_svmr_prob=0.0)
    lse.train()
    for m_ in lse.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
    a_, b_, c_ = lse(batch, task="tvr", compute_loss=True)
    (a_ + b_ + c_).mean().backward()
    lp = dict(lse.named_parameters())
    out.update({"narrow.lse.loss_st_ed": a_.detach().numpy(), "narrow.lse.loss_neg_ctx": b_.detach().numpy(),
                "narrow.lse.loss_neg_q": c_.detach().numpy()})
    for n_ in ("video_query_linear.weight", "q_feat_attn.query_input_proj.net.1.weight",
               "v_encoder.c_encoder.encoder.layer.0.output.dense.weight",
               "v_encoder.f_encoder.encoder.layer.1.attention.self.key.weight"):
        out["narrow.lse.grad." + n_] = lp[n_].grad.numpy().copy()
    print("narrow lse + hard negatives:", float(a_), float(b_), float(c_))
    print("narrow: f_attn_masks", tuple(batch["f_attn_masks"].shape), "max_vl + max_sl =",
          batch["f_v_feats"].shape[1] + batch["f_sub_input_ids"].shape[1], "losses", l_st.tolist(), l_ctx.tolist(), l_q.tolist())

    # ---- clamp: a 600-token subtitle -> f_sub_pos_ids clamped at 511 (data/data.py:427-429); collate only -------
    clamp = [dict(nframe_db=4, subs=[(0, [0, 1], 599), (1, [2], 3)], queries=[(5, (0.0, 2.0))]),
             dict(nframe_db=3, subs=[(0, [0, 1, 2], 7)], queries=[(520, (1.0, 4.0))])]
    batch, raw, feats = build_case(D, clamp, vfeat=4, max_clip_len=100, seed=12)
    assert int(batch["f_sub_pos_ids"].max()) == 511 and batch["f_sub_pos_ids"].shape[1] == 600
    out.update(pack("clamp", batch, raw, feats))

    # ---- random ragged batches (collate only, small features) --------------------------------------------------
    g = torch.Generator().manual_seed(5)
    n_narrow = 0
    for i in range(8):
        batch, raw, feats = build_case(D, random_spec(g, 2 + i % 4, 24), vfeat=6, max_clip_len=24, seed=20 + i)
        n_narrow += batch["f_attn_masks"].shape[1] < batch["f_v_feats"].shape[1] + batch["f_sub_input_ids"].shape[1]
        out.update(pack("rand%d" % i, batch, raw, feats))
    print("random cases narrower than max_vl + max_sl:", n_narrow, "of 8")
    out["__cases__"] = np.array(json.dumps(["narrow", "clamp"] + ["rand%d" % i for i in range(8)]))
    np.savez_compressed(os.path.join(HERE, "case_collate.npz"), **out)
    print("wrote case_collate.npz", os.path.getsize(os.path.join(HERE, "case_collate.npz")), "bytes")


if __name__ == "__main__":
    main()
