#!/usr/bin/env python3
"""Golden vectors for the pre-training heads (BASELINE.json configs[3]: MLM, MFM-NCE / MFFR, FOM) by
importing the *reference* HERO code with the tiny weights of tests/golden/tiny_model.npz.

Container-only (needs /root/reference), exactly like make_golden.py, whose stubs and synthetic
collate it reuses.  Writes tests/golden/case_pretrain.npz:

  mlm.*   f_encoder(batch, 'mlm')        per-masked-token losses + prediction scores
          (model/encoder.py:355-374)
  mfm.*   v_encoder(batch, 'mfm-nce'/'mffr')  losses (model/model.py:239-289)
  fom.*   v_encoder(batch, 'fom')        loss + logits (model/model.py:306-336)
  vsm.*   model(batch, 'vsm') with TWO queries per video (data/vsm.py:21,105-145: query_per_video, q_vidx) - the
          cross branch of get_pred_from_mod_query + the [row, q_vidx] selection (model/pretrain.py:72-110, 188-201)
          and the per > 1 ranking loss (model/pretrain.py:203-292): the three weighted losses
  grad.<task>.<param>  gradients of mean(loss) for a few parameters of each task

Run:  python tests/golden/make_golden_pretrain.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G            # noqa: E402  (stubs + synthetic collate)


def main():
    G.install_stubs()
    sys.path.insert(0, G.REF)
    from model.vcmr import HeroForVcmr            # noqa: reference import

    z = np.load(os.path.join(HERE, "tiny_model.npz"), allow_pickle=False)
    sd = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("__")}
    model = HeroForVcmr.from_pretrained(
        os.path.join(HERE, "tiny_config.json"), state_dict=sd, vfeat_dim=G.VFEAT, max_frm_seq_len=G.MAX_FRM,
        lw_neg_ctx=8.0, lw_neg_q=8.0, lw_st_ed=0.01, ranking_loss_type="hinge", use_hard_negative=False,
        hard_pool_size=20, margin=0.1, use_all_neg=True, drop_svmr_prob=0.0)
    model.train()
    for m_ in model.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
    enc = model.v_encoder
    params = dict(model.named_parameters())
    gen = torch.Generator().manual_seed(11)
    d = {}

    def grads(task, loss, names):
        model.zero_grad()
        loss.mean().backward()
        for n_ in names:
            d["grad.%s.%s" % (task, n_)] = params[n_].grad.detach().numpy().copy()

    vb = G.synth_video_batch(
        gen,
        subs=[[([0, 1, 2], 6), ([3, 4], 5), ([], 4), ([6, 7], 7)],
              [([0, 1], 4), ([2, 3, 4], 7)],
              [([1, 2, 3, 4], 5), ([5], 3), ([7, 8], 6)]],
        n_frames=[9, 6, 10])
    d.update(G.pack_batch(vb))

    # ---- MLM (data/mlm.py:134-176 batch keys) ------------------------------------------------
    T, Lf = vb["f_attn_masks"].shape
    tgt = torch.zeros(T, Lf, dtype=torch.bool)
    picks = [(0, 4), (0, 7), (1, 3), (2, 2), (3, 5), (4, 3), (5, 6), (6, 5), (8, 4)]
    for r, c in picks:
        assert vb["f_attn_masks"][r, c] == 1
        tgt[r, c] = True
    labels = torch.randint(3, 160, (len(picks),), generator=gen)
    mlm = {"input_ids": vb["f_sub_input_ids"], "position_ids": vb["f_sub_pos_ids"], "v_feat": vb["f_v_feats"],
           "f_pos_ids": vb["f_v_pos_ids"], "attn_masks": vb["f_attn_masks"], "gather_index": vb["f_gather_index"],
           "txt_mask_tgt": tgt, "txt_labels": labels}
    d["in.txt_mask_tgt"], d["in.txt_labels"] = tgt.numpy(), labels.numpy()
    scores = enc(mlm, "mlm", compute_loss=False)
    loss = enc(mlm, "mlm", compute_loss=True)
    d["mlm.scores"], d["mlm.loss"] = scores.detach().numpy(), loss.detach().numpy()
    grads("mlm", loss, ["v_encoder.f_encoder.lm_head.dense.weight", "v_encoder.f_encoder.lm_head.bias",
                        "v_encoder.f_encoder.lm_head.LayerNorm.weight",
                        "v_encoder.f_encoder.embeddings.word_embeddings.weight",
                        "v_encoder.f_encoder.encoder.layer.1.attention.self.key.weight"])

    # ---- MFM (data/mfm.py:77-97) -----------------------------------------------------------------
    cm = torch.zeros(3, 10, dtype=torch.bool)
    cm[0, 1] = cm[0, 6] = cm[1, 3] = cm[2, 2] = cm[2, 8] = True
    fm = torch.zeros(T, vb["f_v_feats"].shape[1], dtype=torch.bool)
    fm[0, 1] = fm[3, 0] = fm[5, 1] = fm[6, 1] = fm[8, 1] = True       # the same frames, per subtitle
    feat_targets = vb["c_v_feats"][cm].clone()
    d["in.c_v_masks"], d["in.f_v_masks"], d["in.feat_targets"] = cm.numpy(), fm.numpy(), feat_targets.numpy()
    for task in ("mfm-nce", "mffr"):
        b = dict(vb)
        b["c_v_feats"] = vb["c_v_feats"].clone()               # forward_mfm mutates it
        b["f_v_feats"] = vb["f_v_feats"].masked_fill(fm.unsqueeze(-1), 0)
        b.update({"c_v_masks": cm, "f_v_masks": fm, "feat_targets": feat_targets})
        loss = enc(b, task, compute_loss=True)
        d["mfm.%s.loss" % task] = loss.detach().numpy()
        grads(task, loss, ["v_encoder.feat_regress.net.0.weight", "v_encoder.feat_regress.net.3.bias",
                           "v_encoder.mask_embedding.weight", "v_encoder.f_encoder.img_embeddings.mask_embedding.weight",
                           "v_encoder.c_encoder.encoder.layer.0.intermediate.dense.weight"])

    # ---- FOM (data/fom.py:50-93) --------------------------------------------------------------------
    B, Lc = vb["c_attn_masks"].shape
    orders = torch.arange(Lc).unsqueeze(0).repeat(B, 1)
    targets = torch.full((B, Lc), -1, dtype=torch.long)
    for b_, nf in enumerate([9, 6, 10]):
        k = 3
        pos = torch.randperm(nf, generator=gen)[:k]
        perm = pos[torch.randperm(k, generator=gen)]
        orders[b_, pos] = perm                                   # frame at pos[i] moves to perm[i]
        targets[b_, perm] = pos
    fb = dict(vb)
    fb.update({"shuffled_orders": orders, "targets": targets})
    d["in.shuffled_orders"], d["in.fom_targets"] = orders.numpy(), targets.numpy()
    logits = enc(fb, "fom", compute_loss=False)
    loss = enc(fb, "fom", compute_loss=True)
    d["fom.logits"], d["fom.loss"] = logits.detach().numpy(), loss.detach().numpy()
    grads("fom", loss, ["v_encoder.fom_output.linear_1.weight", "v_encoder.fom_output.linear_2.bias",
                        "v_encoder.fom_output.LayerNorm.weight", "v_encoder.frame_transform.net.1.weight",
                        "v_encoder.c_encoder.embeddings.position_embeddings.weight"])
    # ---- VSM, several queries per video (data/vsm.py:105-145) - appended LAST: the draws above are unchanged ----------
    per = 2
    qi, qp, qm = G.synth_queries(gen, 3 * per, [5, 7, 4, 6, 3, 7])
    tg = torch.tensor([[1, 3], [0, 2], [2, 4], [-1, 5], [3, 6], [7, 9]])          # one ignored start (padding_value -1)
    vsm = dict(vb)
    vsm.update({"query_input_ids": qi, "query_pos_ids": qp, "query_attn_masks": qm, "targets": tg,
                "q_vidx": torch.arange(3 * per) // per})
    for k in ("query_input_ids", "query_pos_ids", "query_attn_masks", "targets", "q_vidx"):
        d["in.vsm." + k] = vsm[k].numpy()
    model.zero_grad()
    losses = model(vsm, task="tvr", compute_loss=True)            # HeroForVcmr routes it to HeroForPretraining.forward(task="vsm")
    d["vsm.losses"] = torch.stack([l.reshape(()) for l in losses]).detach().numpy()
    sum(l.sum() for l in losses).backward()
    for n_ in ["video_query_linear.weight", "video_st_predictor.weight", "q_feat_attn.modular_vector_mapping.weight",
               "v_encoder.c_encoder.encoder.layer.0.output.dense.weight", "v_encoder.f_encoder.encoder.layer.0.attention.self.query.weight"]:
        d["grad.vsm.%s" % n_] = params[n_].grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "case_pretrain.npz"), **d)
    print("vsm (2 queries per video)", d["vsm.losses"])
    print("case_pretrain.npz  mlm", float(d["mlm.loss"].mean()), " mfm-nce", float(d["mfm.mfm-nce.loss"].mean()),
          " mffr", float(d["mfm.mffr.loss"].mean()), " fom", float(d["fom.loss"]))


if __name__ == "__main__":
    main()
