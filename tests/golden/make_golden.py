#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the *reference* HERO code.

Container-only: needs /root/reference (read-only checkout of linjieli222/HERO).
Nothing here is shipped to or executed on the GPU box; only the `.npz` files it
writes are.  The reference depends on two packages that are absent from this
image; both are replaced by in-memory stubs before the import:

  * ``apex.normalization.fused_layer_norm.FusedLayerNorm`` -> ``torch.nn.LayerNorm``
    (apex's own CPU path is ``F.layer_norm``; same math, biased variance).
  * ``horovod.torch`` -> single-process identities (size 1, rank 0).

Fixtures (tiny HERO config: hidden 128, 2 heads of 64, ff 256, f/c layers 2/1,
vocab 160, vfeat 96 — head size 64 is kept because the HIP attention kernel is
specialised for HERO's head size):

  tiny_model.npz    state dict of HeroForVcmr (fp32) + the config JSON text
  case_regular.npz  2 videos, all subtitles have frames, no padding
  case_ragged.npz   3 videos: subtitle without frames, frames without subtitle,
                    padded frames/tokens, unequal subtitle counts
  case_mfm.npz      regular batch + f_v_masks/c_v_masks (mask embedding path)
  case_train.npz    HeroForVcmr training forward (dropout 0): three losses,
                    gradients of selected parameters, AdamW result after 2 steps

Run:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("HERO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

TINY = {
    "f_config": {
        "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu",
        "hidden_dropout_prob": 0.1, "hidden_size": 128,
        "initializer_range": 0.02, "intermediate_size": 256,
        "max_position_embeddings": 66, "num_attention_heads": 2,
        "num_hidden_layers": 2, "type_vocab_size": 2, "vocab_size": 160},
    "c_config": {
        "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu",
        "hidden_dropout_prob": 0.1, "hidden_size": 128,
        "initializer_range": 0.02, "intermediate_size": 256,
        "max_position_embeddings": 66, "num_attention_heads": 2,
        "num_hidden_layers": 1, "type_vocab_size": 2},
    "q_config": {
        "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu",
        "hidden_dropout_prob": 0.1, "hidden_size": 128,
        "initializer_range": 0.02, "intermediate_size": 256,
        "num_attention_heads": 2, "max_position_embeddings": 66,
        "num_hidden_layers": 0, "type_vocab_size": 1, "vocab_size": 160},
}
VFEAT = 96
MAX_FRM = 16


def install_stubs():
    apex = types.ModuleType("apex")
    norm = types.ModuleType("apex.normalization")
    fln = types.ModuleType("apex.normalization.fused_layer_norm")
    fln.FusedLayerNorm = torch.nn.LayerNorm
    amp = types.ModuleType("apex.amp")
    apex.normalization, norm.fused_layer_norm, apex.amp = norm, fln, amp
    sys.modules.update({"apex": apex, "apex.normalization": norm,
                        "apex.normalization.fused_layer_norm": fln,
                        "apex.amp": amp})
    hvd_pkg = types.ModuleType("horovod")
    hvd = types.ModuleType("horovod.torch")
    hvd.size = lambda: 1
    hvd.rank = lambda: 0
    hvd.local_rank = lambda: 0
    hvd.allgather = lambda t, name=None: t
    hvd.allgather_async = lambda t, name=None: t
    hvd.synchronize = lambda h: h
    hvd_pkg.torch = hvd
    sys.modules.update({"horovod": hvd_pkg, "horovod.torch": hvd})


def synth_video_batch(gen, subs, n_frames, max_frames=None, vocab=160):
    """Build a batch dict the way the reference's collate does.

    ``subs``: per video, list of (frame_idx list, n_tokens incl. SEP).
    ``n_frames``: per video frame count.  Follows data/data.py:355-471.
    """
    T = sum(len(s) for s in subs)
    v_lens, t_lens = [], []
    for vs in subs:
        for fr, nt in vs:
            v_lens.append(max(len(fr), 1))
            t_lens.append(nt)
    max_vl, max_sl = max(v_lens), max(t_lens)
    max_f = max_frames or max(n_frames)
    B = len(subs)
    c_v_feats = torch.zeros(B, max_f, VFEAT)
    c_attn = torch.zeros(B, max_f, dtype=torch.long)
    for b, nf in enumerate(n_frames):
        c_v_feats[b, :nf] = torch.randn(nf, VFEAT, generator=gen)
        c_attn[b, :nf] = 1
    ids = torch.ones(T, max_sl, dtype=torch.long)          # pad id 1
    f_v = torch.zeros(T, max_vl, VFEAT)
    f_attn = torch.zeros(T, max_vl + max_sl, dtype=torch.long)
    gidx = torch.arange(max_vl + max_sl).unsqueeze(0).repeat(T, 1)
    sub2frm, num_subs = [], []
    row = 0
    for b, vs in enumerate(subs):
        cur = []
        for sid, (fr, nt) in enumerate(vs):
            ids[row, 0] = 2
            ids[row, 1:nt] = torch.randint(3, vocab, (nt - 1,), generator=gen)
            if len(fr):
                f_v[row, :len(fr)] = c_v_feats[b, fr]
                f_attn[row, :len(fr) + nt] = 1
                nf = len(fr)
            else:                                            # data/data.py:380-382
                f_attn[row, 1:1 + nt] = 1
                nf = 1
            gidx[row, nf:nf + nt] = torch.arange(max_vl, max_vl + nt)
            cur.append((sid, list(fr)))
            row += 1
        sub2frm.append(cur)
        num_subs.append(len(vs))
    return {
        "f_sub_input_ids": ids,
        "f_sub_pos_ids": torch.arange(max_sl).unsqueeze(0),
        "f_v_feats": f_v,
        "f_v_pos_ids": torch.arange(max_vl).unsqueeze(0),
        "f_attn_masks": f_attn,
        "f_gather_index": gidx,
        "c_v_feats": c_v_feats,
        "c_attn_masks": c_attn,
        "num_subs": num_subs,
        "sub_idx2frame_idx": sub2frm,
    }


def synth_queries(gen, n, lens, vocab=160):
    L = max(lens)
    ids = torch.ones(n, L, dtype=torch.long)
    m = torch.zeros(n, L, dtype=torch.long)
    for i, l in enumerate(lens):
        ids[i, 0] = 0
        ids[i, 1:l] = torch.randint(3, vocab, (l - 1,), generator=gen)
        m[i, :l] = 1
    return ids, torch.arange(L).unsqueeze(0), m


def pack_batch(batch):
    """Tensors -> numpy; host lists -> JSON strings (npz-friendly)."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out["in." + k] = v.numpy()
        else:
            out["in." + k] = np.array(json.dumps(v))
    return out


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from model.vcmr import HeroForVcmr            # noqa: reference import
    from optim.adamw import AdamW                  # noqa: reference import

    cfg_path = os.path.join(HERE, "tiny_config.json")
    with open(cfg_path, "w") as f:
        json.dump(TINY, f, indent=1)

    torch.manual_seed(0)
    model = HeroForVcmr.from_pretrained(
        cfg_path, state_dict={}, vfeat_dim=VFEAT, max_frm_seq_len=MAX_FRM,
        lw_neg_ctx=8.0, lw_neg_q=8.0, lw_st_ed=0.01,
        ranking_loss_type="hinge", use_hard_negative=False,
        hard_pool_size=20, margin=0.1, use_all_neg=True, drop_svmr_prob=0.0)
    # LN affine / biases are 1/0 after init; perturb so that every parameter
    # matters in the comparison.
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    np.savez_compressed(
        os.path.join(HERE, "tiny_model.npz"),
        __config__=np.array(json.dumps(TINY)),
        __vfeat__=np.array(VFEAT), __max_frm__=np.array(MAX_FRM),
        **{k: v.numpy() for k, v in sd.items()})

    model.eval()
    gen = torch.Generator().manual_seed(1)

    def run_repr(batch, name, extra=None):
        with torch.no_grad():
            fr = model.v_encoder.f_encoder(batch, "repr")[0]
            pre = model.v_encoder.forward_repr(
                __import__("collections").defaultdict(lambda: None, batch),
                encode_clip=False)
            out = model.v_encoder(batch, "repr")
        d = pack_batch(batch)
        d["out.f_seq"] = fr.numpy()
        d["out.pre_temporal"] = pre.numpy()
        d["out.repr"] = out.numpy()
        if extra:
            d.update(extra)
        np.savez_compressed(os.path.join(HERE, name), **d)
        print(name, "f_seq", tuple(fr.shape), "repr", tuple(out.shape),
              "abs-mean", float(out.abs().mean()))

    # --- regular -----------------------------------------------------------
    reg = synth_video_batch(
        gen,
        subs=[[([0, 1, 2], 6), ([3, 4, 5], 6), ([6, 7, 8], 6)],
              [([0, 1, 2], 6), ([3, 4, 5], 6), ([6, 7, 8], 6)]],
        n_frames=[9, 9])
    qi, qp, qm = synth_queries(gen, 2, [5, 5])
    with torch.no_grad():
        txt = model.v_encoder.f_encoder(
            {"input_ids": qi, "pos_ids": qp, "attn_masks": qm}, "txt")[0]
    run_repr(reg, "case_regular.npz", {
        "in.query_input_ids": qi.numpy(), "in.query_pos_ids": qp.numpy(),
        "in.query_attn_masks": qm.numpy(), "out.txt": txt.numpy()})

    # --- ragged ------------------------------------------------------------
    rag = synth_video_batch(
        gen,
        subs=[[([0, 1], 4), ([], 7), ([2, 3, 4, 5, 6], 3), ([9], 9)],
              [([1, 2, 3], 5)],
              [([0], 2), ([1, 2], 8), ([5, 6, 7, 8], 6)]],
        n_frames=[12, 5, 9], max_frames=13)
    qi, qp, qm = synth_queries(gen, 3, [4, 9, 6])
    with torch.no_grad():
        txt = model.v_encoder.f_encoder(
            {"input_ids": qi, "pos_ids": qp, "attn_masks": qm}, "txt")[0]
    run_repr(rag, "case_ragged.npz", {
        "in.query_input_ids": qi.numpy(), "in.query_pos_ids": qp.numpy(),
        "in.query_attn_masks": qm.numpy(), "out.txt": txt.numpy()})

    # --- mfm-masked (mask embedding on both streams) --------------------------
    mfm = synth_video_batch(
        gen,
        subs=[[([0, 1, 2], 5), ([3, 4], 6)], [([0, 1], 4), ([2, 3, 4], 7)]],
        n_frames=[6, 5], max_frames=6)
    cm = torch.zeros(2, 6, dtype=torch.bool)
    cm[0, 1] = cm[0, 4] = cm[1, 2] = True
    fm = torch.zeros(4, 3, dtype=torch.bool)
    fm[0, 1] = fm[1, 1] = fm[3, 0] = True          # same frames, per-subtitle view
    mfm["c_v_masks"], mfm["f_v_masks"] = cm, fm
    mfm["c_v_feats"] = mfm["c_v_feats"].masked_fill(cm.unsqueeze(-1), 0)
    mfm["f_v_feats"] = mfm["f_v_feats"].masked_fill(fm.unsqueeze(-1), 0)
    # reference's forward_mfm adds mask_embedding to c_v_feats before repr
    # (model/model.py:244-247); reproduce that call order here.
    with torch.no_grad():
        cvm = mfm["c_v_feats"] + model.v_encoder.mask_embedding(cm.long())
    mb = dict(mfm)
    mb["c_v_feats"] = cvm
    run_repr(mb, "case_mfm.npz", {"in.c_v_feats_unmasked": mfm["c_v_feats"].numpy()})

    # --- training forward + grads + 2 AdamW steps ----------------------------
    model.train()
    for m_ in model.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
    import random
    random.seed(0)
    tr = synth_video_batch(
        gen,
        subs=[[([0, 1, 2], 6), ([3, 4], 5), ([], 4), ([6, 7], 7)],
              [([0, 1], 4), ([2, 3, 4], 7)],
              [([1, 2, 3, 4], 5), ([5], 3), ([7, 8], 6)],
              [([0, 1, 2, 3], 8)]],
        n_frames=[9, 6, 10, 4])
    qi, qp, qm = synth_queries(gen, 4, [6, 4, 8, 5])
    tr.update({"query_input_ids": qi, "query_pos_ids": qp,
               "query_attn_masks": qm,
               "targets": torch.tensor([[1, 3], [0, 2], [4, 8], [1, 2]]),
               "q_vidx": torch.arange(4)})
    names = [
        "v_encoder.f_encoder.encoder.layer.0.attention.self.query.weight",
        "v_encoder.f_encoder.encoder.layer.0.attention.self.value.bias",
        "v_encoder.f_encoder.encoder.layer.1.output.dense.weight",
        "v_encoder.f_encoder.encoder.layer.1.output.LayerNorm.weight",
        "v_encoder.f_encoder.encoder.layer.0.intermediate.dense.bias",
        "v_encoder.f_encoder.embeddings.word_embeddings.weight",
        "v_encoder.f_encoder.embeddings.position_embeddings.weight",
        "v_encoder.f_encoder.embeddings.token_type_embeddings.weight",
        "v_encoder.f_encoder.embeddings.LayerNorm.bias",
        "v_encoder.f_encoder.img_embeddings.img_linear.weight",
        "v_encoder.f_encoder.img_embeddings.img_LayerNorm.weight",
        "v_encoder.f_encoder.img_embeddings.position_embeddings.weight",
        "v_encoder.frame_transform.LayerNorm.bias",
        "v_encoder.frame_transform.net.1.weight",
        "v_encoder.c_encoder.embeddings.position_embeddings.weight",
        "v_encoder.c_encoder.encoder.layer.0.attention.self.key.weight",
        "v_encoder.c_encoder.encoder.layer.0.output.dense.bias",
        "q_feat_attn.query_self_attention.self.query.weight",
        "q_feat_attn.query_input_proj.net.1.weight",
        "video_query_linear.weight",
        "video_st_predictor.weight",
    ]
    params = dict(model.named_parameters())
    from optim.misc import build_optimizer           # noqa: reference import
    opts = types.SimpleNamespace(lr_mul=1.0, learning_rate=1e-3,
                                 weight_decay=0.01, optim="adamw",
                                 betas=[0.9, 0.98])
    optim = build_optimizer(model, opts)
    d = pack_batch(tr)
    for step in range(2):
        optim.zero_grad()
        l_st, l_ctx, l_q = model(tr, task="tvr", compute_loss=True)
        loss = (l_st + l_ctx + l_q).mean()
        loss.backward()
        gsq = sum(float(p.grad.double().pow(2).sum())
                  for p in model.parameters() if p.grad is not None)
        if step == 0:
            d["out.loss_st_ed"] = l_st.detach().numpy()
            d["out.loss_neg_ctx"] = l_ctx.detach().numpy()
            d["out.loss_neg_q"] = l_q.detach().numpy()
            d["out.grad_norm"] = np.array(gsq ** 0.5)
            for n_ in names:
                d["grad." + n_] = params[n_].grad.detach().numpy().copy()
            d["out.no_grad_params"] = np.array(json.dumps(
                [n_ for n_, p in model.named_parameters() if p.grad is None]))
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optim.step()
        d["out.loss_step%d" % step] = loss.detach().numpy()
    for n_ in names:
        d["after2." + n_] = params[n_].detach().numpy().copy()
    d["out.param_l2_after2"] = np.array(
        sum(float(p.double().pow(2).sum()) for p in model.parameters()) ** 0.5)
    np.savez_compressed(os.path.join(HERE, "case_train.npz"), **d)
    print("case_train.npz losses", float(d["out.loss_st_ed"]),
          float(d["out.loss_neg_ctx"]), float(d["out.loss_neg_q"]),
          "gnorm", float(d["out.grad_norm"]))


if __name__ == "__main__":
    main()
