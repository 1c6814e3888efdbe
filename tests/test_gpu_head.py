"""VSM / VCMR head kernels (include/hero_hip.h "task head") against the PyTorch formulation of the
same maths that hero_amd/model/pretrain.py keeps for the non-training configurations - forward
values and every gradient, fp32 (tolerance: summation order only) and bf16 inputs."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import hero_oracle as O
from tests.util import GOLDEN, load_tiny, rel_err, to_dev

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def mask_logits(x, m):
    return x * m + (1 - m) * -1e4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,D", [(5, 15, 768), (3, 7, 128), (2, 70, 256)])
def test_query_pool(dtype, B, L, D):
    from hero_amd.head import QueryPoolFn
    q = rnd(B, L, D, seed=1, dtype=dtype).requires_grad_(True)
    w = rnd(1, D, seed=2, scale=0.2).requires_grad_(True)
    m = torch.ones(B, L).cuda()
    m[0, L - 2:] = 0
    g = rnd(B, D, seed=3)
    out = QueryPoolFn.apply(q, m, w)
    out.backward(g)
    qr = q.detach().float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    att = F.softmax(mask_logits(qr @ wr.t(), m.unsqueeze(2)), dim=1)
    ref = torch.einsum("blm,bld->bmd", att, qr)[:, 0]
    ref.backward(g)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(out, ref) < tol
    assert rel_err(q.grad, qr.grad) < tol and rel_err(w.grad, wr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rownorm(dtype):
    from hero_amd.head import RowNormFn
    x = rnd(37, 5, 768, seed=1, dtype=dtype)
    x[3, 2] = 0                                         # a zero row: the eps clamp
    x = x.requires_grad_(True)
    g = rnd(37, 5, 768, seed=2)
    y = RowNormFn.apply(x, 1e-5)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    yr = F.normalize(xr, dim=-1, eps=1e-5)
    yr.backward(g)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(y, yr) < tol and rel_err(x.grad, xr.grad) < tol


def torch_rank_losses(q2v, per, margin, lse, hard, pool, hard_w):
    """model/pretrain.py:203-264 (use_all_neg) as kept in hero_amd.model.pretrain."""
    nq, nv = q2v.shape
    own = torch.arange(nq, device=q2v.device) // per
    is_pos = own.unsqueeze(1) == torch.arange(nv, device=q2v.device).unsqueeze(0)
    pos = torch.where(is_pos, q2v, torch.zeros_like(q2v)).sum(1)
    masked = torch.where(is_pos, torch.full_like(q2v, 999), q2v)
    rl = (lambda p, n: torch.log1p(torch.exp(n - p))) if lse else (lambda p, n: torch.clamp(margin + n - p, min=0))

    def weight(l):
        if not hard:
            return l
        w = torch.full_like(l, 0.1)
        w[:, :pool] = hard_w
        return w * l
    neg_ctx = masked.sort(dim=1, descending=True)[0][:, 1:]
    l_ctx = weight(rl(pos.view(nq, 1), neg_ctx))
    neg_q = masked.t().sort(dim=1, descending=True)[0][:, per:]
    l_q = rl(pos.view(nv, per, 1), neg_q.unsqueeze(1))
    l_q = weight(l_q.view(-1, l_q.size(2)))
    return l_ctx.mean(1).mean(0), l_q.mean(1).mean(0)


@pytest.mark.parametrize("per,lse,hard", [(1, False, False), (1, False, True), (2, True, False), (3, False, True)])
def test_video_rank_loss(per, lse, hard):
    from hero_amd.head import VideoRankLossFn
    N, L, D = 7, 13, 128
    M = N * per
    qn = F.normalize(rnd(M, D, seed=1), dim=-1).requires_grad_(True)
    cn = F.normalize(rnd(N, L, D, seed=2), dim=-1).requires_grad_(True)
    mask = torch.ones(N, L).cuda()
    mask[1, 5:] = 0
    mask[4, 9:] = 0
    gs = (1.7, -0.6)
    lc, lq = VideoRankLossFn.apply(qn, cn, mask, (0, N), 0.1, lse, hard, 3, 10.0)
    (gs[0] * lc + gs[1] * lq).backward()
    qr, cr = qn.detach().clone().requires_grad_(True), cn.detach().clone().requires_grad_(True)
    scores = torch.einsum("md,nld->mln", qr, cr)
    q2v = mask_logits(scores, mask.t().unsqueeze(0)).max(dim=1)[0]
    rc, rq = torch_rank_losses(q2v, per, 0.1, lse, hard, 3, 10.0)
    (gs[0] * rc + gs[1] * rq).backward()
    assert abs(float(lc.detach()) - float(rc.detach())) < 1e-5 and abs(float(lq.detach()) - float(rq.detach())) < 1e-5
    assert rel_err(qn.grad, qr.grad) < 1e-4 and rel_err(cn.grad, cr.grad) < 1e-4


def test_video_rank_loss_own_slice():
    """A data-parallel rank asks only for its own videos' rows of d(contexts)."""
    from hero_amd.head import VideoRankLossFn
    N, L, D = 6, 9, 64
    qn = F.normalize(rnd(N, D, seed=1), dim=-1).requires_grad_(True)
    cn = F.normalize(rnd(N, L, D, seed=2), dim=-1).requires_grad_(True)
    mask = torch.ones(N, L).cuda()
    lc, lq = VideoRankLossFn.apply(qn, cn, mask, (0, N), 0.1, False, False, 20, 10.0)
    (lc + lq).backward()
    full = cn.grad.clone()
    cn.grad = None
    lc, lq = VideoRankLossFn.apply(qn, cn, mask, (2, 3), 0.1, False, False, 20, 10.0)
    (lc + lq).backward()
    torch.testing.assert_close(cn.grad[2:5], full[2:5])
    assert float(cn.grad[:2].abs().sum()) == 0 and float(cn.grad[5:].abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_st_ed_loss(dtype):
    from hero_amd.head import StEdLossFn
    B, L, D, K = 6, 23, 256, 5
    q2 = rnd(B, D, seed=1, scale=0.2).requires_grad_(True)
    ctx = rnd(B, L, D, seed=2, dtype=dtype).requires_grad_(True)
    w_st = rnd(1, 1, K, seed=3, scale=0.5).requires_grad_(True)
    w_ed = rnd(1, 1, K, seed=4, scale=0.5).requires_grad_(True)
    mask = torch.ones(B, L).cuda()
    mask[2, 15:] = 0
    tg = torch.tensor([[0, 3], [5, 9], [2, 14], [-1, 4], [22, 22], [7, -1]]).cuda()
    loss = StEdLossFn.apply(q2, ctx, mask, w_st, w_ed, tg)
    (2.5 * loss).backward()
    q2r = q2.detach().clone().requires_grad_(True)
    cr = ctx.detach().float().requires_grad_(True)
    wsr, wer = w_st.detach().clone().requires_grad_(True), w_ed.detach().clone().requires_grad_(True)
    sim = torch.einsum("bd,bld->bl", q2r, cr).unsqueeze(1)
    st = mask_logits(F.conv1d(sim, wsr, padding=K // 2).squeeze(1), mask)
    ed = mask_logits(F.conv1d(sim, wer, padding=K // 2).squeeze(1), mask)
    ref = F.cross_entropy(st, tg[:, 0], ignore_index=-1) + F.cross_entropy(ed, tg[:, 1], ignore_index=-1)
    (2.5 * ref).backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert abs(float(loss.detach()) - float(ref.detach())) < tol * max(1.0, abs(float(ref.detach())))
    assert rel_err(q2.grad, q2r.grad) < tol and rel_err(ctx.grad, cr.grad) < tol
    assert rel_err(w_st.grad, wsr.grad) < tol and rel_err(w_ed.grad, wer.grad) < tol


@pytest.mark.parametrize("hard,loss_type", [(False, "hinge"), (True, "hinge"), (False, "lse")])
def test_model_fused_head_equals_torch_head(hard, loss_type):
    """Whole model, tiny golden weights, fp32: the three losses and the gradients of every parameter
    agree between the HIP head and the PyTorch head."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
    b = to_dev(batch, "cuda")
    res = []
    for fused in (True, False):
        HF.set_grad_sink(None)
        model, _, _ = load_tiny("cuda", ranking_loss_type=loss_type, use_hard_negative=hard, hard_pool_size=1)
        model.train()
        set_dropout(model, 0.0)
        model.fused_head = fused
        model.q_feat_attn.fused_pool = fused
        losses = model(b, task="tvr", compute_loss=True)
        sum(l.sum() for l in losses).backward()
        res.append(([float(l.sum()) for l in losses], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    (lf, gf), (lt, gt) = res
    for a, c in zip(lf, lt):
        assert abs(a - c) < 1e-5 * max(1.0, abs(c)), (lf, lt)
    assert set(gf) == set(gt)
    worst = max(rel_err(gf[k], gt[k]) for k in gt if float(gt[k].abs().max()) > 1e-8)
    assert worst < 2e-4, worst


@pytest.mark.parametrize("w_ctx,w_q", [(0.0, 8.0), (8.0, 0.0)])
def test_video_rank_loss_zero_weight_switches_that_term_off(w_ctx, w_q):
    """the backward kernels read a loss weight of exactly 0 as 1, so a configuration with ONE of
    lw_neg_ctx / lw_neg_q at zero (model/pretrain.py:80,112 allows it) got the full unweighted gradient of the term the
    forward had switched off.  The scales are plain multipliers since ABI version 3."""
    from hero_amd.head import VideoRankLossFn
    N, L, D = 7, 13, 128
    qn = F.normalize(rnd(N, D, seed=1), dim=-1).requires_grad_(True)
    cn = F.normalize(rnd(N, L, D, seed=2), dim=-1).requires_grad_(True)
    mask = torch.ones(N, L).cuda()
    mask[1, 5:] = 0
    lc, lq = VideoRankLossFn.apply(qn, cn, mask, (0, N), 0.1, False, False, 3, 10.0, w_ctx, w_q)
    (lc + lq).backward()                    # autograd sends g = 1 for BOTH outputs
    qr, cr = qn.detach().clone().requires_grad_(True), cn.detach().clone().requires_grad_(True)
    q2v = mask_logits(torch.einsum("md,nld->mln", qr, cr), mask.t().unsqueeze(0)).max(dim=1)[0]
    rc, rq = torch_rank_losses(q2v, 1, 0.1, False, False, 3, 10.0)
    (w_ctx * rc + w_q * rq).backward()
    assert abs(float(lc) - w_ctx * float(rc)) < 1e-5 and abs(float(lq) - w_q * float(rq)) < 1e-5
    assert (float(lc) == 0.0) == (w_ctx == 0.0) and (float(lq) == 0.0) == (w_q == 0.0)
    assert rel_err(qn.grad, qr.grad) < 1e-4 and rel_err(cn.grad, cr.grad) < 1e-4


@pytest.mark.parametrize("lw", [dict(lw_neg_ctx=0.0, lw_neg_q=8.0), dict(lw_neg_ctx=8.0, lw_neg_q=0.0)])
def test_model_fused_head_with_one_ranking_weight_zero(lw):
    """The same at model level: losses and every parameter gradient of the HIP head == the PyTorch head with one of the two
    ranking-loss weights at zero."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
    b = to_dev(batch, "cuda")
    res = []
    for fused in (True, False):
        HF.set_grad_sink(None)
        model, _, _ = load_tiny("cuda", **lw)
        model.train()
        set_dropout(model, 0.0)
        model.fused_head = fused
        model.q_feat_attn.fused_pool = fused
        losses = model(b, task="tvr", compute_loss=True)
        sum(l.sum() for l in losses).backward()
        res.append(([float(l.sum()) for l in losses], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    (lf, gf), (lt, gt) = res
    for a, c in zip(lf, lt):
        assert abs(a - c) < 1e-5 * max(1.0, abs(c)), (lf, lt)
    assert set(gf) == set(gt)
    worst = max(rel_err(gf[k], gt[k]) for k in gt if float(gt[k].abs().max()) > 1e-8)
    assert worst < 2e-4, worst


def test_expand_rows_forward_and_backward():
    """ExpandRowsFn (round 6): out[r] = x[idx[r]] with repeated sources; backward = the per-source sum in pair order."""
    from hero_amd import functional as HF
    x = rnd(5, 7, 64, seed=1).requires_grad_(True)
    idx = torch.tensor([0, 0, 3, 1, 3, 3, 4, 0]).cuda()                 # source 2 is never referenced
    out = HF.ExpandRowsFn.apply(x, idx)
    assert torch.equal(out, x.detach()[idx])
    g = rnd(8, 7, 64, seed=2)
    out.backward(g)
    ref = torch.zeros_like(x).index_add_(0, idx, g)
    torch.testing.assert_close(x.grad, ref, rtol=1e-6, atol=1e-6)
    assert float(x.grad[2].abs().sum()) == 0
    xb = x.detach().to(torch.bfloat16).requires_grad_(True)
    HF.ExpandRowsFn.apply(xb, idx).backward(g.to(torch.bfloat16))
    assert rel_err(xb.grad.float(), ref) < 2e-2


@pytest.mark.parametrize("q_vidx", ["grouped", "shuffled"])
@pytest.mark.parametrize("loss_type", ["hinge", "lse"])
def test_model_fused_head_with_several_queries_per_video(q_vidx, loss_type):
    """Round 6: the HIP head with query_per_video > 1 (data/vsm.py:105-145; BASELINE.json configs[3] has 5) == the PyTorch
    head: three losses and every parameter gradient.  "shuffled": a q_vidx that is NOT arange // per - the start / end term
    follows q_vidx, the ranking terms take m // per, in both heads as in the reference (model/pretrain.py:188-264)."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.utils.misc import set_dropout
    from tests.test_oracle_golden import _vsm_batch
    hero_amd.set_compute_dtype(torch.float32)
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_pretrain.npz"))
    b = to_dev(_vsm_batch(batch), "cuda")
    nq, nv = b["query_input_ids"].shape[0], b["c_attn_masks"].shape[0]
    assert nq > nv and nq % nv == 0
    if q_vidx == "shuffled":
        b["q_vidx"] = b["q_vidx"].flip(0).contiguous()
        b["targets"] = torch.zeros_like(b["targets"])                    # frame 0 is valid in every video
    res = []
    for fused in (True, False):
        HF.set_grad_sink(None)
        HF.reset_caches()
        model, _, _ = load_tiny("cuda", ranking_loss_type=loss_type)
        model.train()
        set_dropout(model, 0.0)
        model.fused_head = fused
        model.q_feat_attn.fused_pool = fused
        assert model._head_is_fusable(torch.empty(nv, 1, 1, device="cuda"), torch.empty(nq, 1, device="cuda"), b) == fused
        losses = model(b, task="tvr", compute_loss=True)
        sum(l.sum() for l in losses).backward()
        res.append(([float(l.sum()) for l in losses], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    (lf, gf), (lt, gt) = res
    for a, c in zip(lf, lt):
        assert abs(a - c) < 1e-5 * max(1.0, abs(c)), (lf, lt)
    assert set(gf) == set(gt)
    worst = max(rel_err(gf[k], gt[k]) for k in gt if float(gt[k].abs().max()) > 1e-8)
    assert worst < 2e-4, worst
