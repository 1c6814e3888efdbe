"""Synthetic batches with the exact keys / shapes / dtypes of the reference's collate functions
(data/data.py:406-471 video_collate, data/vcmr.py:120-145 query_collate), SURVEY.md §8(d).

Nothing here reads a dataset: features ~ N(0,1), token ids uniform in [3, vocab) with SEP (2) first
for subtitles and CLS (0) first for queries, pad id 1, masks = length prefixes.
"""
import torch

# name -> (videos, frames/video, subs/video, frames/sub, tokens/sub, query tokens)
SHAPES = {
    "D1": dict(videos=2, frames=32, subs=8, fps=4, toks=8, qtoks=12),       # config 1 (plumbing)
    "D2": dict(videos=32, frames=60, subs=15, fps=4, toks=20, qtoks=15),    # TVR finetune, per GPU
    "D4": dict(videos=512, frames=256, subs=64, fps=4, toks=20, qtoks=15),  # long-video stress
}


def video_batch(subs, n_frames, vfeat_dim, vocab, gen, max_frames=None):
    """subs: per video a list of (frame index list, n_tokens incl. SEP).  Random features / tokens pushed through
    the collate of hero_amd.collate (== the reference's, tests/test_cpu_collate.py).  max_frames: pad the clip
    stream to a fixed number of frames (same-shape batches for hipGraph replay)."""
    from .collate import video_collate, video_item
    items = []
    for vs, nf in zip(subs, n_frames):
        v_feat = torch.randn(nf, vfeat_dim, generator=gen)
        toks = [torch.randint(3, vocab, (nt - 1,), generator=gen).tolist() for _, nt in vs]
        items.append(video_item(v_feat, [(sid, list(fr)) for sid, (fr, _) in enumerate(vs)], toks, sep=2))
    batch = video_collate(items)
    NF = batch["c_v_feats"].shape[1]
    if max_frames and max_frames > NF:
        pad = max_frames - NF
        batch["c_v_feats"] = torch.nn.functional.pad(batch["c_v_feats"], (0, 0, 0, pad))
        batch["c_attn_masks"] = torch.nn.functional.pad(batch["c_attn_masks"], (0, pad))
        batch["c_pos_ids"] = torch.arange(max_frames, dtype=torch.long).repeat(len(subs), 1)
    return batch


def query_batch(n, lens, vocab, gen):
    Lq = max(lens)
    ids = torch.ones(n, Lq, dtype=torch.long)
    m = torch.zeros(n, Lq, dtype=torch.long)
    for i, l in enumerate(lens):
        ids[i, 0] = 0
        if l > 1:
            ids[i, 1:l] = torch.randint(3, vocab, (l - 1,), generator=gen)
        m[i, :l] = 1
    return {"query_input_ids": ids, "query_pos_ids": torch.arange(Lq).unsqueeze(0),
            "query_attn_masks": m}


def to_device(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def make_batch(name="D2", vfeat_dim=4352, vocab=50272, seed=1, device="cpu", ragged=False,
               videos=None):
    """Canonical (or ragged) TVR-shaped VCMR batch: one query per video (sampled_by_q=True)."""
    sh = dict(SHAPES[name])
    if videos is not None:
        sh["videos"] = videos
    gen = torch.Generator().manual_seed(seed)
    B = sh["videos"]
    if not ragged:
        subs = [[(list(range(s * sh["fps"], (s + 1) * sh["fps"])), sh["toks"])
                 for s in range(sh["subs"])] for _ in range(B)]
        n_frames = [sh["frames"]] * B
        qlens = [sh["qtoks"]] * B
    else:
        subs, n_frames, qlens = [], [], []
        ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))  # noqa: E731
        for _ in range(B):
            nf = ri(30, 100)
            cur, f0 = [], 0
            for _s in range(ri(8, 25)):
                k = ri(0, 8)
                fr = [f for f in range(f0, min(f0 + k, nf))]
                f0 += len(fr)
                cur.append((fr, ri(4, 40)))
            subs.append(cur)
            n_frames.append(nf)
            qlens.append(ri(5, 25))
    batch = video_batch(subs, n_frames, vfeat_dim, vocab, gen)
    batch.update(query_batch(B, qlens, vocab, gen))
    st = torch.tensor([int(torch.randint(0, max(nf - 1, 1), (1,), generator=gen)) for nf in n_frames])
    ed = torch.minimum(st + 1 + torch.randint(0, 4, (B,), generator=gen),
                       torch.tensor(n_frames) - 1)
    batch["targets"] = torch.stack([st, ed], dim=1)
    batch["q_vidx"] = torch.arange(B)
    return to_device(batch, device)


# ---- pre-training tasks (BASELINE configs[3], config/pretrain-tv-16gpu.json) ---------------------------
MASK_ID = 4      # any in-vocabulary id stands in for <mask>; the token it replaces is the label


def make_pretrain_batches(name="D2", vfeat_dim=4352, vocab=50272, seed=1, device="cpu", mask_prob=0.15,
                          queries_per_video=5, videos=None):
    """One synthetic batch per pre-training task on the same videos, with the batch keys of the reference's
    collates: 'mlm' (data/mlm.py:134-176), 'mfm-nce' (data/mfm.py:77-97), 'fom' (data/fom.py:50-93), 'vsm'
    (data/vsm.py:105-145: `queries_per_video` queries for every video)."""
    sh = dict(SHAPES[name])
    if videos is not None:
        sh["videos"] = videos
    gen = torch.Generator().manual_seed(seed)
    B = sh["videos"]
    subs = [[(list(range(s * sh["fps"], (s + 1) * sh["fps"])), sh["toks"]) for s in range(sh["subs"])]
            for _ in range(B)]
    n_frames = [sh["frames"]] * B
    vb = video_batch(subs, n_frames, vfeat_dim, vocab, gen)
    T, max_vl = vb["f_v_feats"].shape[:2]
    Lf = vb["f_attn_masks"].shape[1]
    out = {}

    # MLM: 15 % of the sub-tokens (never the leading SEP) are replaced by <mask>; targets live in the
    # interleaved (frames first, then tokens) layout of f_attn_masks
    ids = vb["f_sub_input_ids"].clone()
    tok = torch.zeros_like(ids, dtype=torch.bool)
    r = 0
    for vs in subs:
        for fr, nt in vs:
            tok[r, 1:nt] = True
            r += 1
    pick = tok & (torch.rand(ids.shape, generator=gen) < mask_prob)
    if not pick.any():
        pick[0, 1] = True
    tgt = torch.zeros(T, Lf, dtype=torch.bool)
    r = 0
    for vs in subs:
        for fr, nt in vs:
            nf = max(len(fr), 1)
            tgt[r, nf:nf + ids.shape[1]][: Lf - nf] = pick[r][: Lf - nf]
            r += 1
    labels = ids[pick]                                      # row-major order == nonzero order of tgt
    ids = ids.masked_fill(pick, MASK_ID)
    out["mlm"] = {"input_ids": ids, "position_ids": vb["f_sub_pos_ids"], "v_feat": vb["f_v_feats"],
                  "f_pos_ids": vb["f_v_pos_ids"], "attn_masks": vb["f_attn_masks"],
                  "gather_index": vb["f_gather_index"], "txt_mask_tgt": tgt, "txt_labels": labels}

    # MFM: 15 % of the frames, masked in the clip stream and in the subtitle that carries them
    cm = (torch.rand(B, max(n_frames), generator=gen) < mask_prob) & vb["c_attn_masks"].bool()
    cm[0, 0] = True
    fm = torch.zeros(T, max_vl, dtype=torch.bool)
    r = 0
    for b, vs in enumerate(subs):
        for fr, nt in vs:
            for k, f in enumerate(fr):
                fm[r, k] = cm[b, f]
            r += 1
    mfm = dict(vb)
    mfm["feat_targets"] = vb["c_v_feats"][cm].clone()
    mfm["c_v_feats"] = vb["c_v_feats"].masked_fill(cm.unsqueeze(-1), 0)
    mfm["f_v_feats"] = vb["f_v_feats"].masked_fill(fm.unsqueeze(-1), 0)
    mfm["c_v_masks"], mfm["f_v_masks"] = cm, fm
    out["mfm-nce"] = mfm

    # FOM: 15 % of the frames of every video are permuted among themselves
    Lc = max(n_frames)
    orders = torch.arange(Lc).unsqueeze(0).repeat(B, 1)
    targets = torch.full((B, Lc), -1, dtype=torch.long)
    for b, nf in enumerate(n_frames):
        k = max(2, int(round(nf * mask_prob)))
        pos = torch.randperm(nf, generator=gen)[:k]
        perm = pos[torch.randperm(k, generator=gen)]
        orders[b, pos] = perm
        targets[b, perm] = pos
    fom = dict(vb)
    fom["shuffled_orders"], fom["targets"] = orders, targets
    out["fom"] = fom

    # VSM: several queries per video (query m belongs to video m // queries_per_video)
    nq = B * queries_per_video
    vsm = dict(vb)
    vsm.update(query_batch(nq, [sh["qtoks"]] * nq, vocab, gen))
    st = torch.randint(0, sh["frames"] - 2, (nq,), generator=gen)
    vsm["targets"] = torch.stack([st, torch.clamp(st + 1 + torch.randint(0, 4, (nq,), generator=gen),
                                                 max=sh["frames"] - 1)], dim=1)
    vsm["q_vidx"] = torch.arange(nq) // queries_per_video
    out["vsm"] = vsm
    return {k: to_device(v, device) for k, v in out.items()}
