"""CPU oracle for the HERO hierarchical-encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 *restatement* of the reference algorithm
(linjieli222/HERO), written functionally over a flat ``{name: tensor}`` parameter
dictionary that uses the reference's state-dict names.  It exists to check the
HIP path; nothing under ``hero_amd/`` imports it.  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Parity status: **pinned** — every function below is checked by
``tests/test_oracle_golden.py`` against vectors produced by importing the
reference itself (``tests/golden/make_golden.py``; the reference ships no tests
or golden vectors of its own).

Citations are ``file:line`` relative to the reference checkout.
"""
import json
import math
from collections import namedtuple

import torch
import torch.nn.functional as F

Cfg = namedtuple("Cfg", "hidden heads ff f_layers c_layers eps")


def cfg_from_json(cfg):
    """cfg: dict with f_config / c_config (config/hero_finetune.json schema)."""
    f, c = cfg["f_config"], cfg["c_config"]
    return Cfg(hidden=f["hidden_size"], heads=f["num_attention_heads"],
               ff=f["intermediate_size"], f_layers=f["num_hidden_layers"],
               c_layers=c["num_hidden_layers"],
               eps=f.get("layer_norm_eps", 1e-12))   # model/encoder.py:54,100


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def layer_norm(x, P, prefix, eps):
    """apex FusedLayerNorm == biased-variance LN over the last dim."""
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def linear(x, P, prefix):
    y = x @ P[prefix + ".weight"].t()
    b = P.get(prefix + ".bias")
    return y if b is None else y + b


def gelu_erf(x):
    """model/layers.py:16-25 (erf form, not tanh)."""
    return x * 0.5 * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def _drop(x, p):
    return F.dropout(x, p, True) if p > 0 else x


def additive_mask(mask):
    """model/layers.py:299-302: (1-m)*-10000 broadcast over heads and queries."""
    return (1.0 - mask.to(torch.float32))[:, None, None, :] * -10000.0


def self_attention(x, add_mask, P, prefix, heads, p_drop=0.0):
    """model/layers.py:124-164.  x (S,L,D); returns context (S,L,D)."""
    S, L, D = x.shape
    dh = D // heads

    def split(t):
        return t.view(S, L, heads, dh).permute(0, 2, 1, 3)

    q = split(linear(x, P, prefix + ".query"))
    k = split(linear(x, P, prefix + ".key"))
    v = split(linear(x, P, prefix + ".value"))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + add_mask
    p = _drop(torch.softmax(s, dim=-1), p_drop)
    return (p @ v).permute(0, 2, 1, 3).reshape(S, L, D)


def attention_block(x, add_mask, P, prefix, heads, eps, p_drop=0.0):
    """BertAttention = self-attention + BertSelfOutput (layers.py:175-179,217-222)."""
    ctx = self_attention(x, add_mask, P, prefix + ".self", heads, p_drop)
    y = _drop(linear(ctx, P, prefix + ".output.dense"), p_drop)
    return layer_norm(y + x, P, prefix + ".output.LayerNorm", eps)


def bert_layer(x, add_mask, P, prefix, heads, eps, p_drop=0.0):
    """model/layers.py:264-272 (post-LN)."""
    a = attention_block(x, add_mask, P, prefix + ".attention", heads, eps, p_drop)
    h = gelu_erf(linear(a, P, prefix + ".intermediate.dense"))
    y = _drop(linear(h, P, prefix + ".output.dense"), p_drop)
    return layer_norm(y + a, P, prefix + ".output.LayerNorm", eps)


def bert_encoder(x, mask, P, prefix, n_layers, heads, eps, p_drop=0.0):
    """model/layers.py:298-327."""
    am = additive_mask(mask)
    for i in range(n_layers):
        x = bert_layer(x, am, P, "%s.layer.%d" % (prefix, i), heads, eps, p_drop)
    return x


# --------------------------------------------------------------------------- #
# embeddings  (model/embed.py)
# --------------------------------------------------------------------------- #
def sub_embeddings(ids, pos_ids, P, prefix, p_drop=0.0):
    """embed.py:28-58.  token type id is 1 for text (embed.py:47-49)."""
    e = (P[prefix + ".word_embeddings.weight"][ids]
         + P[prefix + ".position_embeddings.weight"][pos_ids]
         + P[prefix + ".token_type_embeddings.weight"][1])
    return _drop(layer_norm(e, P, prefix + ".LayerNorm", 1e-5), p_drop)


def img_embeddings(feat, pos_ids, type_row, P, prefix, img_masks=None, p_drop=0.0):
    """embed.py:102-117."""
    if img_masks is not None:   # nn.Embedding(2, D, padding_idx=0): row 0 receives no gradient (embed.py:96)
        feat = feat + F.embedding(img_masks.long(), P[prefix + ".mask_embedding.weight"], padding_idx=0)
    t = linear(layer_norm(feat, P, prefix + ".img_LayerNorm", 1e-5),
               P, prefix + ".img_linear")
    e = t + P[prefix + ".position_embeddings.weight"][pos_ids] + type_row
    return _drop(layer_norm(e, P, prefix + ".LayerNorm", 1e-5), p_drop)


def frame_embeddings(x, P, prefix, p_drop=0.0):
    """embed.py:146-161: positions 0..L-1."""
    L = x.shape[1]
    e = x + P[prefix + ".position_embeddings.weight"][:L]
    return _drop(layer_norm(e, P, prefix + ".LayerNorm", 1e-5), p_drop)


# --------------------------------------------------------------------------- #
# encoders  (model/encoder.py, model/model.py)
# --------------------------------------------------------------------------- #
def f_encoder_repr(batch, P, cfg, prefix="v_encoder.f_encoder", p_drop=0.0):
    """CrossModalTrm 'repr' (encoder.py:256-285, 336-352). Returns (T, L, D)."""
    txt = sub_embeddings(batch["f_sub_input_ids"], batch["f_sub_pos_ids"],
                         P, prefix + ".embeddings", p_drop)
    type_row = P[prefix + ".embeddings.token_type_embeddings.weight"][1]
    img = img_embeddings(batch["f_v_feats"], batch["f_v_pos_ids"], type_row,
                         P, prefix + ".img_embeddings",
                         batch.get("f_v_masks"), p_drop)
    cat = torch.cat([img, txt], dim=1)
    gi = batch["f_gather_index"].unsqueeze(-1).expand(-1, -1, cfg.hidden)
    emb = torch.gather(cat, 1, gi)
    return bert_encoder(emb, batch["f_attn_masks"], P, prefix + ".encoder",
                        cfg.f_layers, cfg.heads, cfg.eps, p_drop)


def f_encoder_txt(ids, pos_ids, mask, P, cfg, prefix="v_encoder.f_encoder",
                  p_drop=0.0):
    """CrossModalTrm 'txt' (encoder.py:312-319)."""
    emb = sub_embeddings(ids, pos_ids, P, prefix + ".embeddings", p_drop)
    return bert_encoder(emb, mask, P, prefix + ".encoder",
                        cfg.f_layers, cfg.heads, cfg.eps, p_drop)


def collect_frame_outputs(f_seq, num_subs, sub2frm, n_videos, n_frames):
    """model/model.py:156-187, vectorised: out[v, f] += f_seq[row(v,sid), j]."""
    rows, cols, dst = [], [], []
    base = 0
    for v, n in enumerate(num_subs):
        for sid, frames in sub2frm[v]:
            for j, f in enumerate(frames):
                rows.append(base + sid)
                cols.append(j)
                dst.append(v * n_frames + f)
        base += n
    out = f_seq.new_zeros(n_videos * n_frames, f_seq.shape[-1])
    if rows:
        src = f_seq[torch.tensor(rows), torch.tensor(cols)]
        out = out.index_add(0, torch.tensor(dst), src)
    return out.view(n_videos, n_frames, -1)


def frame_transform(x, P, prefix="v_encoder.frame_transform", p_drop=0.0):
    """LinearLayer: LN -> Dropout -> Linear -> ReLU (layers.py:86-93)."""
    h = _drop(layer_norm(x, P, prefix + ".LayerNorm", 1e-5), p_drop)
    return torch.relu(linear(h, P, prefix + ".net.1"))


def c_encoder(x, mask, P, cfg, prefix="v_encoder.c_encoder", p_drop=0.0):
    """TemporalTrm.forward (encoder.py:413-423)."""
    emb = frame_embeddings(x, P, prefix + ".embeddings", p_drop)
    return bert_encoder(emb, mask, P, prefix + ".encoder",
                        cfg.c_layers, cfg.heads, cfg.eps, p_drop)


def forward_repr(batch, P, cfg, encode_clip=True, p_drop=0.0, taps=None):
    """HierarchicalVlModel.forward_repr (model/model.py:195-224)."""
    f_seq = f_encoder_repr(batch, P, cfg, p_drop=p_drop)
    B, NF = batch["c_v_feats"].shape[:2]
    matched = collect_frame_outputs(f_seq, batch["num_subs"],
                                    batch["sub_idx2frame_idx"], B, NF)
    pre = frame_transform(batch["c_v_feats"], P, p_drop=p_drop) + matched
    if taps is not None:
        taps["f_seq"], taps["pre_temporal"] = f_seq, pre
    if not encode_clip:
        return pre
    return c_encoder(pre, batch["c_attn_masks"], P, cfg, p_drop=p_drop)


# --------------------------------------------------------------------------- #
# pre-training heads (BASELINE.json configs[3]): MLM, MFM, FOM
# pinned by tests/golden/case_pretrain.npz (tests/golden/make_golden_pretrain.py)
# --------------------------------------------------------------------------- #
def lm_head(x, P, prefix="v_encoder.f_encoder.lm_head"):
    """BertLMPredictionHead (layers.py:330-354): dense -> gelu -> LN(1e-5) -> tied decoder + bias."""
    h = layer_norm(gelu_erf(linear(x, P, prefix + ".dense")), P, prefix + ".LayerNorm", 1e-5)
    return h @ P[prefix + ".decoder.weight"].t() + P[prefix + ".bias"]


def mlm_scores(batch, P, cfg, prefix="v_encoder.f_encoder"):
    """CrossModalTrm.forward_mlm (encoder.py:355-374): scores of the masked positions only."""
    b = {"f_sub_input_ids": batch["input_ids"], "f_sub_pos_ids": batch["position_ids"],
         "f_v_feats": batch["v_feat"], "f_v_pos_ids": batch["f_pos_ids"],
         "f_attn_masks": batch["attn_masks"], "f_gather_index": batch["gather_index"]}
    seq = f_encoder_repr(b, P, cfg, prefix)
    return lm_head(seq[batch["txt_mask_tgt"]], P, prefix + ".lm_head")


def mlm_loss(batch, P, cfg, vocab_pad=0):
    """vocab_pad: columns appended by pad_vocab() and stripped before the loss (encoder.py:224-233, 366-367)."""
    scores = mlm_scores(batch, P, cfg)
    if vocab_pad:
        scores = scores[:, :-vocab_pad]
    return F.cross_entropy(scores, batch["txt_labels"], reduction="none")


def feat_regress(x, P, prefix="v_encoder.feat_regress"):
    """FrameFeatureRegression (model.py:104-114): Linear -> GELU -> LN(1e-5) -> Linear."""
    h = layer_norm(gelu_erf(linear(x, P, prefix + ".net.0")), P, prefix + ".net.2", 1e-5)
    return linear(h, P, prefix + ".net.3")


def mfm_loss(batch, P, cfg, loss="nce", nce_temp=1.0):
    """HierarchicalVlModel.forward_mfm / mfm_nce (model.py:239-289).  batch['c_v_feats'] holds the
    frame features; masked frames are zeroed and get the mask embedding (row 1) added here; the
    per-subtitle stream gets its own mask embedding through batch['f_v_masks']."""
    cm = batch["c_v_masks"]
    b = dict(batch)
    b["c_v_feats"] = (batch["c_v_feats"].masked_fill(cm.unsqueeze(-1), 0)
                      + F.embedding(cm.long(), P["v_encoder.mask_embedding.weight"], padding_idx=0))  # model.py:133
    out = forward_repr(b, P, cfg)
    pred = feat_regress(out[cm], P)
    if loss == "regression":
        return F.mse_loss(pred, batch["feat_targets"], reduction="none")
    neg = feat_regress(out[~cm], P)
    logits = torch.cat([pred @ batch["feat_targets"].t(), pred @ neg.t()], 1)
    return F.cross_entropy(logits / nce_temp, torch.arange(pred.shape[0]), reduction="none")


def fom_logits(batch, P, cfg):
    """HierarchicalVlModel.forward_fom (model.py:306-336): frames re-ordered by scatter, temporal
    encoder, MLPLayer (layers.py:48-61: Linear -> gelu -> LN(1e-5) -> Linear)."""
    pre = forward_repr(batch, P, cfg, encode_clip=False)
    idx = batch["shuffled_orders"].unsqueeze(-1).expand_as(pre)
    shuf = torch.zeros_like(pre).scatter(1, idx, pre)
    enc = c_encoder(shuf, batch["c_attn_masks"], P, cfg)
    x = enc.reshape(-1, enc.shape[-1])
    pf = "v_encoder.fom_output"
    h = layer_norm(gelu_erf(linear(x, P, pf + ".linear_1")), P, pf + ".LayerNorm", 1e-5)
    return linear(h, P, pf + ".linear_2")


def fom_loss(batch, P, cfg):
    return F.cross_entropy(fom_logits(batch, P, cfg), batch["targets"].reshape(-1), ignore_index=-1)


# --------------------------------------------------------------------------- #
# VSM / VCMR head  (model/pretrain.py, model/encoder.py:426-485)
# --------------------------------------------------------------------------- #
def mask_logits(x, m):
    """modeling_utils.py:42-43."""
    return x * m + (1 - m) * -1e4


def query_feat_encoder(q_seq, q_mask, P, cfg, prefix="q_feat_attn", p_drop=0.0):
    h = _drop(layer_norm(q_seq, P, prefix + ".query_input_proj.LayerNorm", 1e-5),
              p_drop)
    h = torch.relu(linear(h, P, prefix + ".query_input_proj.net.1"))
    L = h.shape[1]
    h = h + P[prefix + ".query_pos_embed.position_embeddings.weight"][:L]
    h = _drop(layer_norm(h, P, prefix + ".query_pos_embed.LayerNorm", 1e-5), p_drop)
    m = q_mask.to(torch.float32)
    a = attention_block(h, additive_mask(m), P, prefix + ".query_self_attention",
                        cfg.heads, cfg.eps, p_drop)
    w = a @ P[prefix + ".modular_vector_mapping.weight"].t()       # (N,L,1)
    w = torch.softmax(mask_logits(w, m.unsqueeze(2)), dim=1)
    return torch.einsum("blm,bld->bmd", w, a)[:, 0]


def st_ed_logits(mod_q, ctx, ctx_mask, P, q_vidx=None):
    """get_pred_from_mod_query + _get_st_ed_prob (pretrain.py:118-166, 188-201): the matched branch when there is one query
    per video; otherwise (several queries per video, data/vsm.py:105-145) the CROSS branch - every query against every
    video, `md,nld->mnl`, both Conv1d on (Nq*Nv, 1, L), mask (1, Nv, L) - followed by the caller's [row, q_vidx] selection
    (pretrain.py:93-99)."""
    q = linear(mod_q, P, "video_query_linear")
    m = ctx_mask.to(torch.float32)
    if q.shape[0] == ctx.shape[0]:
        sim = torch.einsum("bd,bld->bl", q, ctx).unsqueeze(1)
        st = F.conv1d(sim, P["video_st_predictor.weight"], padding=2).squeeze(1)
        ed = F.conv1d(sim, P["video_ed_predictor.weight"], padding=2).squeeze(1)
        return mask_logits(st, m), mask_logits(ed, m)
    sim = torch.einsum("md,nld->mnl", q, ctx)
    nq, nc, ln = sim.shape
    flat = sim.reshape(nq * nc, 1, ln)
    st = F.conv1d(flat, P["video_st_predictor.weight"], padding=2).view(nq, nc, ln)
    ed = F.conv1d(flat, P["video_ed_predictor.weight"], padding=2).view(nq, nc, ln)
    st, ed = mask_logits(st, m.unsqueeze(0)), mask_logits(ed, m.unsqueeze(0))
    rows = torch.arange(nq)
    return st[rows, q_vidx], ed[rows, q_vidx]


def video_level_scores(mod_q, ctx, ctx_mask):
    """get_video_level_scores at world size 1 (pretrain.py:364-413)."""
    q = F.normalize(mod_q, dim=-1, eps=1e-5)
    c = F.normalize(ctx, dim=-1, eps=1e-5)
    s = torch.einsum("md,nld->mln", q, c)
    m = ctx_mask.t().unsqueeze(0).to(s.dtype)
    return mask_logits(s, m).max(dim=1)[0]


def video_level_loss(scores, margin=0.1, hard=None, ranking="hinge"):
    """get_video_level_loss, use_all_neg, 'mean' (pretrain.py:203-292) with get_ranking_loss
    (pretrain.py:340-362): 'hinge' max(0, margin + neg - pos) or 'lse' log(1 + exp(neg - pos)).
    hard = (pool_size, weight) or None."""
    if ranking == "hinge":
        rl = lambda pos, neg: torch.clamp(margin + neg - pos, min=0)      # noqa: E731
    elif ranking == "lse":
        rl = lambda pos, neg: torch.log1p(torch.exp(neg - pos))           # noqa: E731
    else:
        raise NotImplementedError("Only support 'hinge' and 'lse'")
    nq, nv = scores.shape
    per = nq // nv
    if nv == 1:
        z = scores.new_zeros(())
        return z, z
    own = torch.arange(nq) // per                       # video of each query
    pos = scores[torch.arange(nq), own]                 # (nq,)
    masked = scores.clone()
    masked[torch.arange(nq), own] = 999
    neg_ctx = masked.sort(dim=1, descending=True)[0][:, 1:]          # (nq, nv-1)
    l_ctx = rl(pos[:, None], neg_ctx)
    neg_q = masked.t().sort(dim=1, descending=True)[0][:, per:]      # (nv, nq-per)
    l_q = rl(pos.view(nv, per, 1), neg_q[:, None, :])
    l_q = l_q.reshape(nq, -1)
    if hard is not None:
        def weigh(t):
            w = torch.full_like(t, 0.1)
            w[:, :hard[0]] = hard[1]
            return t * w
        l_ctx, l_q = weigh(l_ctx), weigh(l_q)
    return l_ctx.mean(1).mean(0), l_q.mean(1).mean(0)


def vsm_losses(batch, P, cfg, lw_st_ed=0.01, lw_neg_ctx=8.0, lw_neg_q=8.0,
               margin=0.1, hard=None, p_drop=0.0, ranking="hinge"):
    """HeroForPretraining.forward('vsm'), training branch (pretrain.py:62-116)."""
    frames = forward_repr(batch, P, cfg, p_drop=p_drop)
    q_seq = f_encoder_txt(batch["query_input_ids"], batch["query_pos_ids"],
                          batch["query_attn_masks"], P, cfg, p_drop=p_drop)
    mod_q = query_feat_encoder(q_seq, batch["query_attn_masks"], P, cfg,
                               p_drop=p_drop)
    st, ed = st_ed_logits(mod_q, frames, batch["c_attn_masks"], P, batch.get("q_vidx"))
    tg = batch["targets"]
    l_st_ed = (F.cross_entropy(st, tg[:, 0], ignore_index=-1)
               + F.cross_entropy(ed, tg[:, 1], ignore_index=-1))
    sc = video_level_scores(mod_q, frames, batch["c_attn_masks"])
    l_ctx, l_q = video_level_loss(sc, margin, hard, ranking)
    return lw_st_ed * l_st_ed, lw_neg_ctx * l_ctx, lw_neg_q * l_q


# --------------------------------------------------------------------------- #
# optimiser  (optim/adamw.py:43-106, optim/misc.py:14-50)
# --------------------------------------------------------------------------- #
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def adamw_step(P, G, state, lr, step, betas=(0.9, 0.98), eps=1e-6, wd=0.01):
    """Bias-corrected Adam, then decoupled decay p -= lr*wd*p (after the update)."""
    b1, b2 = betas
    for n, p in P.items():
        g = G.get(n)
        if g is None:
            continue
        m, v = state.setdefault(n, (torch.zeros_like(p), torch.zeros_like(p)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        ss = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        p.addcdiv_(m, v.sqrt().add_(eps), value=-ss)
        if not any(t in n for t in NO_DECAY):
            p.add_(p, alpha=-lr * wd)


# --------------------------------------------------------------------------- #
# fixture helpers
# --------------------------------------------------------------------------- #
def load_npz_model(path):
    import numpy as np
    z = np.load(path)
    cfg = json.loads(str(z["__config__"]))
    P = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("__")}
    return P, cfg, int(z["__vfeat__"]), int(z["__max_frm__"])


def load_npz_case(path):
    import numpy as np
    z = np.load(path)
    batch, outs = {}, {}
    for k in z.files:
        a = z[k]
        if k.startswith("in."):
            batch[k[3:]] = (json.loads(str(a)) if a.dtype.kind == "U"
                            else torch.from_numpy(a))
        else:
            outs[k] = (json.loads(str(a)) if a.dtype.kind == "U"
                       else torch.from_numpy(a))
    return batch, outs
