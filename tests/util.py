"""Shared helpers for the parity tests (fixtures -> hero_amd models / device batches)."""
import os

import torch

from oracle import hero_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tiny(device, cls=None, **kw):
    """hero_amd model with the reference-generated tiny weights (tests/golden/tiny_model.npz)."""
    from hero_amd.model import HeroForVcmr
    P, cfgj, vfeat, max_frm = O.load_npz_model(os.path.join(GOLDEN, "tiny_model.npz"))
    args = dict(lw_neg_ctx=8.0, lw_neg_q=8.0, lw_st_ed=0.01, ranking_loss_type="hinge",
                use_hard_negative=False, hard_pool_size=20, margin=0.1, use_all_neg=True,
                drop_svmr_prob=0.0)
    args.update(kw)
    model = (cls or HeroForVcmr).from_pretrained(os.path.join(GOLDEN, "tiny_config.json"),
                                                 {k: v.clone() for k, v in P.items()},
                                                 vfeat_dim=vfeat, max_frm_seq_len=max_frm, **args)
    return model.to(device), P, O.cfg_from_json(cfgj)


def elem_rel_err(a, b, mask=None):
    """ELEMENT-WISE relative error, max over elements of |a-b| / (|b| + rms(b)) - stricter than `rel_err`, which
    divides by the global maximum: every element must be right relative to its own size (floored at the tensor's rms
    so that exact zeros do not make it meaningless)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if mask is not None:
        m = mask.bool().cpu()
        a, b = a[m], b[m]
    rms = b.pow(2).mean().sqrt().clamp_min(1e-12)
    return float(((a - b).abs() / (b.abs() + rms)).max())


def to_dev(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def rel_err(a, b, mask=None):
    """max |a-b| / max |b| over the masked rows (the reference's padded rows hold finite garbage,
    SURVEY.md §7 — parity is judged under the attention mask)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if mask is not None:
        m = mask.bool().cpu()
        a, b = a[m], b[m]
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
