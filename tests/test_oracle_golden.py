"""Pin the CPU oracle to vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU-only; runs in seconds."""
import os

import pytest
import torch

from oracle import hero_oracle as O

TOL = dict(rtol=1e-5, atol=2e-6)


@pytest.fixture(scope="module")
def tiny(golden_dir):
    P, cfgj, vfeat, max_frm = O.load_npz_model(os.path.join(golden_dir, "tiny_model.npz"))
    return P, O.cfg_from_json(cfgj)


@pytest.mark.parametrize("case", ["regular", "ragged", "mfm"])
def test_repr_matches_reference(tiny, golden_dir, case):
    P, cfg = tiny
    batch, outs = O.load_npz_case(os.path.join(golden_dir, "case_%s.npz" % case))
    taps = {}
    with torch.no_grad():
        out = O.forward_repr(batch, P, cfg, taps=taps)
    torch.testing.assert_close(taps["f_seq"], outs["out.f_seq"], **TOL)
    torch.testing.assert_close(taps["pre_temporal"], outs["out.pre_temporal"], **TOL)
    torch.testing.assert_close(out, outs["out.repr"], **TOL)


@pytest.mark.parametrize("case", ["regular", "ragged"])
def test_txt_matches_reference(tiny, golden_dir, case):
    P, cfg = tiny
    batch, outs = O.load_npz_case(os.path.join(golden_dir, "case_%s.npz" % case))
    with torch.no_grad():
        t = O.f_encoder_txt(batch["query_input_ids"], batch["query_pos_ids"],
                            batch["query_attn_masks"], P, cfg)
    torch.testing.assert_close(t, outs["out.txt"], **TOL)


def test_train_losses_grads_and_adamw(tiny, golden_dir):
    P0, cfg = tiny
    batch, outs = O.load_npz_case(os.path.join(golden_dir, "case_train.npz"))
    P = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pad"))
         for k, v in P0.items()}
    state = {}
    for step in range(2):
        for p in P.values():
            p.grad = None
        l_st, l_ctx, l_q = O.vsm_losses(batch, P, cfg)
        loss = l_st + l_ctx + l_q
        loss.backward()
        if step == 0:
            torch.testing.assert_close(l_st.detach().reshape(()), outs["out.loss_st_ed"].reshape(()), **TOL)
            torch.testing.assert_close(l_ctx.detach(), outs["out.loss_neg_ctx"], **TOL)
            torch.testing.assert_close(l_q.detach(), outs["out.loss_neg_q"], **TOL)
            for k, g in outs.items():
                if k.startswith("grad."):
                    torch.testing.assert_close(P[k[5:]].grad, g, rtol=1e-4, atol=1e-6)
            no_grad = set(outs["out.no_grad_params"])
            mine = {k for k, p in P.items() if p.requires_grad and p.grad is None}
            # lm_head.decoder.weight is tied to the word embedding in the
            # reference (one parameter); the flat dict holds it twice.
            mine.discard("v_encoder.f_encoder.lm_head.decoder.weight")
            assert mine == no_grad
        torch.testing.assert_close(loss.detach(), outs["out.loss_step%d" % step].reshape(()), **TOL)
        G = {k: p.grad for k, p in P.items() if p.requires_grad and p.grad is not None}
        gn = torch.sqrt(sum(g.double().pow(2).sum() for g in G.values())).float()
        if step == 0:
            torch.testing.assert_close(gn.double(), outs["out.grad_norm"].double().reshape(()),
                                       rtol=1e-5, atol=0)
        clip = min(1.0, 1.0 / (float(gn) + 1e-6))
        G = {k: g * clip for k, g in G.items()}
        with torch.no_grad():
            O.adamw_step({k: p for k, p in P.items() if p.requires_grad}, G, state,
                         lr=1e-3, step=step + 1)
    for k, v in outs.items():
        if k.startswith("after2."):
            torch.testing.assert_close(P[k[7:]].detach(), v, rtol=1e-5, atol=1e-6)


# ---- pre-training heads (BASELINE.json configs[3]) pinned by tests/golden/case_pretrain.npz --------
def _task_batches(batch):
    vb = {k: v for k, v in batch.items()}
    mlm = {"input_ids": vb["f_sub_input_ids"], "position_ids": vb["f_sub_pos_ids"], "v_feat": vb["f_v_feats"],
           "f_pos_ids": vb["f_v_pos_ids"], "attn_masks": vb["f_attn_masks"], "gather_index": vb["f_gather_index"],
           "txt_mask_tgt": vb["txt_mask_tgt"], "txt_labels": vb["txt_labels"]}
    mfm = dict(vb)
    mfm["f_v_feats"] = vb["f_v_feats"].masked_fill(vb["f_v_masks"].unsqueeze(-1), 0)
    fom = dict(vb)
    fom["targets"] = vb["fom_targets"]
    for k in ("f_v_masks", "c_v_masks"):
        fom.pop(k)
    return mlm, mfm, fom


def _vsm_batch(batch):
    """The several-queries-per-video VSM batch of case_pretrain.npz (data/vsm.py:105-145 keys)."""
    vsm = {k: v for k, v in batch.items() if not k.startswith("vsm.") and k not in ("f_v_masks", "c_v_masks")}
    for k in ("query_input_ids", "query_pos_ids", "query_attn_masks", "targets", "q_vidx"):
        vsm[k] = batch["vsm." + k]
    return vsm


def _leaf_params(P0):
    return {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pad")) for k, v in P0.items()}


def _check_grads(P, outs, task):
    tied = ("v_encoder.f_encoder.embeddings.word_embeddings.weight", "v_encoder.f_encoder.lm_head.decoder.weight")
    n = 0
    for k, g in outs.items():
        if not k.startswith("grad.%s." % task):
            continue
        name = k[len("grad.%s." % task):]
        got = P[name].grad
        if name == tied[0] and P[tied[1]].grad is not None:        # one tied parameter in the reference
            got = got + P[tied[1]].grad
        torch.testing.assert_close(got, g, rtol=1e-4, atol=1e-6)
        n += 1
    assert n >= 5


def test_pretrain_heads_match_reference(tiny, golden_dir):
    P0, cfg = tiny
    batch, outs = O.load_npz_case(os.path.join(golden_dir, "case_pretrain.npz"))
    mlm, mfm, fom = _task_batches(batch)
    P = _leaf_params(P0)
    sc = O.mlm_scores(mlm, P, cfg)
    torch.testing.assert_close(sc.detach(), outs["mlm.scores"], **TOL)
    loss = O.mlm_loss(mlm, P, cfg)
    torch.testing.assert_close(loss.detach(), outs["mlm.loss"], **TOL)
    loss.mean().backward()
    _check_grads(P, outs, "mlm")
    for task, kind in (("mfm-nce", "nce"), ("mffr", "regression")):
        P = _leaf_params(P0)
        loss = O.mfm_loss(mfm, P, cfg, loss=kind)
        torch.testing.assert_close(loss.detach(), outs["mfm.%s.loss" % task], rtol=1e-5, atol=1e-5)
        loss.mean().backward()
        _check_grads(P, outs, task)
    P = _leaf_params(P0)
    logits = O.fom_logits(fom, P, cfg)
    torch.testing.assert_close(logits.detach(), outs["fom.logits"], **TOL)
    loss = O.fom_loss(fom, P, cfg)
    torch.testing.assert_close(loss.detach(), outs["fom.loss"], **TOL)
    loss.backward()
    _check_grads(P, outs, "fom")


def test_vsm_with_several_queries_per_video_matches_reference(tiny, golden_dir):
    """Round 6: configs[3]'s VSM batches carry query_per_video queries for every video (data/vsm.py:21,105-145), which takes
    the reference through the CROSS branch of get_pred_from_mod_query and the [row, q_vidx] selection
    (model/pretrain.py:93-99, 188-201) and through the per > 1 ranking loss.  The reference's three weighted losses and
    five gradients on a 3-video x 2-query batch (one ignored start target) pin the oracle's restatement of it."""
    P0, cfg = tiny
    batch, outs = O.load_npz_case(os.path.join(golden_dir, "case_pretrain.npz"))
    vsm = _vsm_batch(batch)
    assert vsm["query_input_ids"].shape[0] == 2 * vsm["c_v_feats"].shape[0]
    P = _leaf_params(P0)
    losses = O.vsm_losses(vsm, P, cfg)
    torch.testing.assert_close(torch.stack([l.reshape(()) for l in losses]).detach(), outs["vsm.losses"], **TOL)
    sum(losses).backward()
    _check_grads(P, outs, "vsm")
