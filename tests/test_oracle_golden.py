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
