"""Autograd nodes of the VSM / VCMR task head on the libhero_hip.so head kernels
(include/hero_hip.h "VSM / VCMR task head"; reference: model/pretrain.py:62-292,
model/encoder.py:460-471).  hero_amd.model.pretrain.HeroForPretraining uses them when the
configuration is the training one (all in-batch negatives, hinge / lse, matched query-video pairs);
every other configuration keeps the PyTorch formulation in that module, which is also what the
parity tests compare these nodes against."""
import ctypes as C

import torch

from . import _lib as L
from . import functional as HF


def _f32c(t):
    if t.dtype == torch.float32 and t.is_contiguous():
        return t.detach()
    if not t.is_floating_point():                       # 0/1 masks: converted once per batch object
        return HF.memo("mask_f32c", (t,), lambda: t.to(torch.float32).contiguous(), spec=(L.DERIVE_F32, 0, 0, 0))
    return t.detach().to(torch.float32).contiguous()


class QueryPoolFn(torch.autograd.Function):
    """pooled[b] = sum_l softmax_l(mask_logits(<q[b,l], w>, mask))[l] * q[b,l]  (fp32 out)."""

    @staticmethod
    def forward(ctx, q, mask, w):
        B, Lq, D = q.shape
        q = q.contiguous()
        mask = _f32c(mask)
        pooled = torch.empty((B, D), dtype=torch.float32, device=q.device)
        att = torch.empty((B, Lq), dtype=torch.float32, device=q.device)
        a = L.QueryPool()
        a.q, a.mask, a.w, a.pooled, a.att = L.ptr(q), L.ptr(mask), L.ptr(w.detach().contiguous()), L.ptr(pooled), L.ptr(att)
        a.B, a.L, a.D, a.dtype = B, Lq, D, L.dt(q)
        L.check(L.lib().hero_query_pool_fwd(C.byref(a), L.stream()))
        ctx.save_for_backward(q, mask, att)
        ctx.w = w
        HF._use(w)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        q, mask, att = ctx.saved_tensors
        w = ctx.w
        B, Lq, D = q.shape
        dq = torch.empty_like(q)
        to_sink = HF._is_param(w)
        part = torch.empty((B, D), dtype=torch.float32, device=q.device)     # per-query shares of dw, folded in a fixed order
        a = L.QueryPool()
        a.q, a.mask, a.w, a.att = L.ptr(q), L.ptr(mask), L.ptr(w.detach().contiguous()), L.ptr(att)
        a.dpooled, a.dq, a.dw = L.ptr(_f32c(dpooled)), L.ptr(dq), L.ptr(part)
        a.B, a.L, a.D, a.dtype = B, Lq, D, L.dt(q)
        L.check(L.lib().hero_query_pool_bwd(C.byref(a), L.stream()))
        if to_sink:
            HF.k_colsum(part, out=HF.SINK.dst(w).view(-1), beta=1.0, on_done=lambda: HF.SINK.done(w))
            return dq, None, None
        return dq, None, HF.k_colsum(part).view_as(w)


class RowNormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1, eps) with fp32 output; x may be bf16."""

    @staticmethod
    def forward(ctx, x, eps):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        y = torch.empty(x2.shape, dtype=torch.float32, device=x.device)
        rn = torch.empty((x2.shape[0],), dtype=torch.float32, device=x.device)
        a = L.RowNorm()
        a.x, a.y, a.rnorm = L.ptr(x2), L.ptr(y), L.ptr(rn)
        a.rows, a.cols, a.x_dtype, a.eps = x2.shape[0], x2.shape[1], L.dt(x2), eps
        L.check(L.lib().hero_rownorm_fwd(C.byref(a), L.stream()))
        ctx.save_for_backward(x2, rn)
        ctx.shp, ctx.eps = shp, eps
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, rn = ctx.saved_tensors
        dx = torch.empty_like(x2)
        a = L.RowNorm()
        a.x, a.rnorm, a.dy, a.dx = L.ptr(x2), L.ptr(rn), L.ptr(_f32c(dy).view(x2.shape)), L.ptr(dx)
        a.rows, a.cols, a.x_dtype, a.eps = x2.shape[0], x2.shape[1], L.dt(x2), ctx.eps
        L.check(L.lib().hero_rownorm_bwd(C.byref(a), L.stream()))
        return dx.view(ctx.shp), None


class VideoRankLossFn(torch.autograd.Function):
    """(normalised queries [M,D], normalised frames [N,L,D], frame mask [N,L]) ->
    (loss_neg_ctx, loss_neg_q): scores GEMM, mask_logits + max over frames, in-batch ranking loss
    over all negatives (model/pretrain.py:203-264 with use_all_neg, 364-382).  `own` = (first video,
    number of videos) of this rank inside the gathered N (gradients of foreign rows are dropped by
    the gather's backward anyway, model/pretrain.py:442-447)."""

    @staticmethod
    def forward(ctx, qn, cn, mask, own, margin, lse, hard, pool, hard_w, w_ctx=1.0, w_q=1.0):
        """w_ctx / w_q: the loss weights (model/pretrain.py:283-290), folded into the final reductions and into the backward
        kernels' upstream-gradient scalars - the returned losses are already weighted."""
        M, D = qn.shape
        N, Lc, _ = cn.shape
        qn, cn, mask = _f32c(qn), _f32c(cn), _f32c(mask)
        dev = qn.device
        rows_c = N * Lc
        ld = (rows_c + 3) & ~3                      # hero_gemm wants a multiple-of-4 output width
        cn2 = cn.view(rows_c, D)
        if ld != rows_c:
            cn2 = torch.cat([cn2, cn2.new_zeros(ld - rows_c, D)], 0)
        s = torch.empty((M, ld), dtype=torch.float32, device=dev)
        HF.k_gemm(qn, cn2, s, M, ld, D, D, D, ld, L.LAYOUT_K, L.LAYOUT_K, L.F32)
        q2v = torch.empty((M, N), dtype=torch.float32, device=dev)
        arg = torch.empty((M, N), dtype=torch.int32, device=dev)
        a = L.ScoreMax()
        a.s, a.mask, a.out, a.arg = L.ptr(s), L.ptr(mask), L.ptr(q2v), L.ptr(arg)
        a.M, a.N, a.L, a.D, a.ld_s = M, N, Lc, D, ld
        L.check(L.lib().hero_score_max_fwd(C.byref(a), L.stream()))
        rows = torch.empty((2, M), dtype=torch.float32, device=dev)
        ds = torch.empty((2, M, N), dtype=torch.float32, device=dev)
        r = L.RankLoss()
        r.s, r.loss_ctx_rows, r.loss_q_rows = L.ptr(q2v), L.ptr(rows[0]), L.ptr(rows[1])
        r.ds_ctx, r.ds_q, r.nq, r.nv = L.ptr(ds[0]), L.ptr(ds[1]), M, N
        r.margin, r.lse, r.hard, r.pool, r.hard_w, r.easy_w = margin, int(lse), int(hard), pool, hard_w, 0.1
        L.check(L.lib().hero_rank_loss(C.byref(r), L.stream()))
        ctx.save_for_backward(qn, cn, mask, arg, ds)
        ctx.own = own
        ctx.w = (float(w_ctx), float(w_q))
        both = torch.empty((2,), dtype=torch.float32, device=dev)
        L.check(L.lib().hero_sums_scaled(L.ptr(rows), 2, M, (C.c_float * 2)(float(w_ctx) / M, float(w_q) / M), L.ptr(both), L.stream()))
        return both[0], both[1]

    @staticmethod
    def backward(ctx, g_ctx, g_q):
        qn, cn, mask, arg, ds = ctx.saved_tensors
        M, D = qn.shape
        N, Lc, _ = cn.shape
        n0, n_own = ctx.own
        dev = qn.device
        g_ctx, g_q = _f32c(g_ctx).reshape(1), _f32c(g_q).reshape(1)      # two device scalars, read where they are (no stack)
        dqn = torch.empty_like(qn)
        full = n0 == 0 and n_own == N
        dcn = (torch.empty if full else torch.zeros)((N, Lc, D), dtype=torch.float32, device=dev)
        a = L.ScoreMax()
        a.mask, a.arg, a.ds_ctx, a.ds_q = L.ptr(mask), L.ptr(arg), L.ptr(ds[0]), L.ptr(ds[1])
        a.gc, a.gq = L.ptr(g_ctx), L.ptr(g_q)
        a.gc_scale, a.gq_scale = ctx.w
        a.qn, a.cn, a.dqn = L.ptr(qn), L.ptr(cn), L.ptr(dqn)
        a.dcn = dcn.data_ptr() + n0 * Lc * D * 4
        a.M, a.N, a.L, a.D, a.n0, a.n_own, a.ld_s = M, N, Lc, D, n0, n_own, N * Lc
        L.check(L.lib().hero_score_max_bwd(C.byref(a), L.stream()))
        return dqn, dcn, None, None, None, None, None, None, None, None, None


_STED_WS = {}


def _sted_workspace(B, dev):
    """Scratch of hero_st_ed_bwd (per-pair shares + arrival counter): zero once, the kernel leaves the counter at zero;
    reuse is stream-ordered."""
    key = (dev.index, B)
    t = _STED_WS.get(key)
    if t is None:
        t = _STED_WS[key] = torch.zeros(L.lib().hero_st_ed_bwd_workspace_bytes(B) // 4, dtype=torch.float32, device=dev)
    return t


class StEdLossFn(torch.autograd.Function):
    """loss_st_ed of matched (query, video) pairs: similarity, the two 1-D convolutions, mask_logits
    and both cross-entropies (model/pretrain.py:96-110, 128-166) -> scalar."""

    @staticmethod
    def forward(ctx, q2, ctxf, mask, w_st, w_ed, targets, weight=1.0):
        """weight: lw_st_ed, folded into the final sum and the backward's upstream gradient."""
        B, Lc, D = ctxf.shape
        q2, ctxf, mask = _f32c(q2), ctxf.contiguous(), _f32c(mask)
        dev = q2.device
        tg = targets.to(torch.int64).contiguous()
        K = w_st.numel()
        rows = torch.empty((B,), dtype=torch.float32, device=dev)
        saved = torch.empty((3, B, Lc), dtype=torch.float32, device=dev)      # p_st, p_ed, sim
        a = StEdLossFn._args(q2, ctxf, mask, w_st, w_ed, tg, saved, K)
        a.loss_rows = L.ptr(rows)
        L.check(L.lib().hero_st_ed_fwd(C.byref(a), L.stream()))
        ctx.save_for_backward(q2, ctxf, mask, tg, saved)
        ctx.ws = (w_st, w_ed)
        ctx.weight = float(weight)
        HF._use(w_st, w_ed)
        out = torch.empty((1,), dtype=torch.float32, device=dev)
        L.check(L.lib().hero_sums_scaled(L.ptr(rows), 1, B, (C.c_float * 1)(float(weight)), L.ptr(out), L.stream()))
        return out[0]

    @staticmethod
    def _args(q2, ctxf, mask, w_st, w_ed, tg, saved, K):
        B, Lc, D = ctxf.shape
        a = L.StEd()
        a.q2, a.ctx, a.mask = L.ptr(q2), L.ptr(ctxf), L.ptr(mask)
        a.w_st, a.w_ed = L.ptr(w_st.detach().contiguous()), L.ptr(w_ed.detach().contiguous())
        a.targets = L.ptr(tg)
        a.p_st, a.p_ed, a.sim = L.ptr(saved[0]), L.ptr(saved[1]), L.ptr(saved[2])
        a.B, a.L, a.D, a.K, a.dtype = B, Lc, D, K, L.dt(ctxf)
        return a

    @staticmethod
    def backward(ctx, g):
        q2, ctxf, mask, tg, saved = ctx.saved_tensors
        w_st, w_ed = ctx.ws
        K = w_st.numel()
        dev = q2.device
        dq2 = torch.empty_like(q2)
        dctx = torch.empty_like(ctxf)
        sink = HF._is_param(w_st) and HF._is_param(w_ed)
        if sink:
            dws, dwe = HF.SINK.dst(w_st).view(-1), HF.SINK.dst(w_ed).view(-1)
        else:
            dws = torch.zeros(K, dtype=torch.float32, device=dev)
            dwe = torch.zeros(K, dtype=torch.float32, device=dev)
        a = StEdLossFn._args(q2, ctxf, mask, w_st, w_ed, tg, saved, K)
        a.g = L.ptr(_f32c(g).reshape(1))
        a.g_scale = ctx.weight
        a.dq2, a.dctx, a.dw_st, a.dw_ed = L.ptr(dq2), L.ptr(dctx), L.ptr(dws), L.ptr(dwe)
        a.ws = L.ptr(_sted_workspace(q2.shape[0], dev))
        L.check(L.lib().hero_st_ed_bwd(C.byref(a), L.stream()))
        if sink:
            HF.SINK.done(w_st)
            HF.SINK.done(w_ed)
            return dq2, dctx, None, None, None, None, None
        return dq2, dctx, None, dws.view_as(w_st), dwe.view_as(w_ed), None, None
