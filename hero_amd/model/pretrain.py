"""VSM / VCMR task head (reference: model/pretrain.py).

The encoders run on the HIP kernels; the head itself is a handful of small fp32 ops on
(Nq, L, Nv)-sized score tensors (einsum, Conv1d k=5, sort, hinge) that stay PyTorch-ROCm calls
(SURVEY.md §8 row H2).  Cross-GPU negatives use torch.distributed (RCCL) instead of Horovod.
"""
import random
from collections import defaultdict

import torch
from torch import nn
from torch.nn import functional as F

from .. import functional as HF
from ..utils import distributed as dist_utils
from .encoder import QueryFeatEncoder
from .model import HeroModel
from .modeling_utils import mask_logits


class HeroForPretraining(HeroModel):
    def __init__(self, config, vfeat_dim, max_frm_seq_len, conv_stride=1, conv_kernel_size=5,
                 ranking_loss_type="hinge", margin=0.1, lw_neg_ctx=0, lw_neg_q=0, lw_st_ed=0.01,
                 drop_svmr_prob=0, use_hard_negative=False, hard_pool_size=20, hard_neg_weight=10,
                 use_all_neg=True):
        super().__init__(config, vfeat_dim, max_frm_seq_len)
        self.config = config
        self.lw_st_ed, self.lw_neg_q, self.lw_neg_ctx = lw_st_ed, lw_neg_q, lw_neg_ctx
        self.ranking_loss_type = ranking_loss_type
        self.use_hard_negative = use_hard_negative
        self.hard_pool_size = hard_pool_size
        self.hard_neg_weight = hard_neg_weight
        self.margin = margin
        self.use_all_neg = use_all_neg
        self.drop_svmr_prob = drop_svmr_prob
        self.gather_gpus = True
        self.fuse_query_pass = True   # subtitle rows and query rows share the cross-modal layer launches
        self.fused_head = True        # loss head on the hero_* head kernels when the configuration allows
        self.video_query_linear = nn.Linear(config.q_config.hidden_size, config.c_config.hidden_size)
        conv = dict(in_channels=1, out_channels=1, kernel_size=conv_kernel_size, stride=conv_stride,
                    padding=conv_kernel_size // 2, bias=False)
        self.video_st_predictor = nn.Conv1d(**conv)
        self.video_ed_predictor = nn.Conv1d(**conv)
        self.qfeat_dim = config.f_config.hidden_size
        self.q_feat_attn = QueryFeatEncoder(config.q_config, self.qfeat_dim)

    # ------------------------------------------------------------------------------------------
    def forward(self, batch, task="vsm", compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        if task != "vsm":
            if task.startswith("mlm") or task in ("mffr", "mfm-nce", "fom"):
                return self.v_encoder(batch, task, compute_loss)
            raise ValueError(f"Unrecognized task {task}")

        if self.fuse_query_pass:
            # same math as the two reference calls (model/pretrain.py:65-70), one pass over the 6 layers
            fe = self.v_encoder.f_encoder
            emb_v = fe._compute_img_txt_embeddings(
                batch["f_sub_input_ids"], batch["f_sub_pos_ids"], batch["f_v_feats"], batch["f_v_pos_ids"],
                batch["f_gather_index"], img_masks=batch["f_v_masks"],
                tail_rows=batch["query_input_ids"].numel(),          # the query rows are stacked right behind (StackRowsFn)
                attention_mask=batch["f_attn_masks"])
            emb_q = fe._compute_txt_embeddings(batch["query_input_ids"], batch["query_pos_ids"])
            seq_v, seq_q = fe.encoder.forward_multi([emb_v, emb_q],
                                                    [batch["f_attn_masks"], batch["query_attn_masks"]])
            frame_embeddings = self.v_encoder.forward_repr(batch, f_seq=seq_v)
            modularized_query = self.q_feat_attn(seq_q, batch["query_attn_masks"])
        else:
            frame_embeddings = self.v_encoder(batch, "repr")
            modularized_query = self.encode_txt_inputs(
                batch["query_input_ids"], batch["query_pos_ids"], batch["query_attn_masks"],
                attn_layer=self.q_feat_attn)

        if compute_loss and self._head_is_fusable(frame_embeddings, modularized_query, batch):
            return self._fused_losses(batch, frame_embeddings, modularized_query)

        q2video_scores = st_prob = ed_prob = None
        if self.lw_st_ed != 0:
            if random.random() > self.drop_svmr_prob or not self.training:
                st_prob, ed_prob = self.get_pred_from_mod_query(
                    frame_embeddings, batch["c_attn_masks"], modularized_query)
        if self.lw_neg_ctx != 0 or self.lw_neg_q != 0:
            q2video_scores = self.get_video_level_scores(modularized_query, frame_embeddings,
                                                         batch["c_attn_masks"])
        if not compute_loss:
            return q2video_scores, st_prob, ed_prob

        z = frame_embeddings.new_zeros(1)
        loss_st_ed, loss_neg_ctx, loss_neg_q = z, z, z
        reduction = "mean" if self.training else "sum"
        if st_prob is not None:
            if st_prob.dim() == 3:
                rows = torch.arange(len(st_prob), device=st_prob.device)
                st_prob, ed_prob = st_prob[rows, batch["q_vidx"]], ed_prob[rows, batch["q_vidx"]]
            tg = batch["targets"]
            loss_st_ed = (F.cross_entropy(st_prob, tg[:, 0].long(), reduction=reduction, ignore_index=-1)
                          + F.cross_entropy(ed_prob, tg[:, 1].long(), reduction=reduction, ignore_index=-1))
        if q2video_scores is not None:
            loss_neg_ctx, loss_neg_q = self.get_video_level_loss(q2video_scores, reduction)
        return (self.lw_st_ed * loss_st_ed, self.lw_neg_ctx * loss_neg_ctx,
                self.lw_neg_q * loss_neg_q)

    # ------------------------------------------------------------------------------------------
    def _head_is_fusable(self, frame_embeddings, modularized_query, batch=None):
        """The HIP head covers the training configuration: 'mean' reduction, all in-batch negatives,
        hinge / lse, stride-1 odd kernels, and either matched (query, video) pairs or (round 6) query_per_video queries per
        video (data/vsm.py:105-145).  The start / end term then runs on the pairs (query m, video q_vidx[m]) for ANY q_vidx;
        the ranking terms take query m's video to be m // per, exactly as the reference's do (model/pretrain.py:203-264)."""
        convs = (self.video_st_predictor, self.video_ed_predictor)
        nv = frame_embeddings.shape[0] * (dist_utils.world_size() if self.gather_gpus else 1)
        nq_, nv_ = modularized_query.shape[0], frame_embeddings.shape[0]
        if nq_ != nv_:
            qv = batch.get("q_vidx") if batch is not None else None
            if qv is None or nv_ == 0 or nq_ % nv_ != 0 or qv.numel() != nq_:
                return False
        return (self.fused_head and self.training and frame_embeddings.is_cuda and self.use_all_neg
                and self.ranking_loss_type in ("hinge", "lse")
                and (nv > 1 or (self.lw_neg_ctx == 0 and self.lw_neg_q == 0))
                and all(c.stride[0] == 1 and c.kernel_size[0] % 2 == 1 and c.kernel_size[0] <= 15
                        for c in convs))

    def _fused_losses(self, batch, frame_embeddings, modularized_query):
        from ..head import RowNormFn, StEdLossFn, VideoRankLossFn
        loss_st_ed = loss_neg_ctx = loss_neg_q = None        # a zero tensor only for a term that is really absent (no fill otherwise)
        cmask = batch["c_attn_masks"]
        if self.lw_st_ed != 0 and random.random() > self.drop_svmr_prob:       # model/pretrain.py:74-75
            q2 = HF.linear(modularized_query, self.video_query_linear.weight, self.video_query_linear.bias)
            ctx_pairs, mask_pairs = frame_embeddings, cmask
            if modularized_query.shape[0] != frame_embeddings.shape[0]:
                # several queries per video: the reference scores every query against every video and keeps [row, q_vidx]
                # (model/pretrain.py:93-99, 188-201) - the same numbers as the matched computation on the pairs
                ctx_pairs = HF.ExpandRowsFn.apply(frame_embeddings, batch["q_vidx"])
                mask_pairs = HF.memo("pair_mask", (cmask, batch["q_vidx"]),          # fp32 at once: what the head kernels read
                                       lambda: cmask.index_select(0, batch["q_vidx"].reshape(-1)).to(torch.float32).contiguous())
            loss_st_ed = StEdLossFn.apply(q2, ctx_pairs, mask_pairs, self.video_st_predictor.weight,
                                          self.video_ed_predictor.weight, batch["targets"], float(self.lw_st_ed))
        if self.lw_neg_ctx != 0 or self.lw_neg_q != 0:
            qn = RowNormFn.apply(modularized_query, 1e-5)
            cn = RowNormFn.apply(frame_embeddings, 1e-5)
            own = (0, cn.shape[0])
            if self.gather_gpus and dist_utils.world_size() > 1:
                qn, cn, cmask, own = dist_utils.gather_negatives(qn, cn, cmask, return_own=True)
            loss_neg_ctx, loss_neg_q = VideoRankLossFn.apply(
                qn, cn, cmask, own, float(self.margin), self.ranking_loss_type == "lse",
                bool(self.use_hard_negative), int(self.hard_pool_size), float(self.hard_neg_weight),
                float(self.lw_neg_ctx), float(self.lw_neg_q))
        # the loss weights (model/pretrain.py:283-290) are folded into the head's own final reductions and backward kernels
        # (round 5: sum / mean / 3 x mul and their backward nodes were ~12 tiny aten launches per micro-step)
        if loss_st_ed is None or loss_neg_ctx is None:
            z = frame_embeddings.new_zeros((), dtype=torch.float32)
            loss_st_ed = z if loss_st_ed is None else loss_st_ed
            loss_neg_ctx, loss_neg_q = (z, z) if loss_neg_ctx is None else (loss_neg_ctx, loss_neg_q)
        return (loss_st_ed, loss_neg_ctx, loss_neg_q)

    @staticmethod
    def _conv5(conv, x):
        """nn.Conv1d(1, 1, k, padding=k//2, bias=False) on (N, 1, L) as unfold + dot: a handful of
        elementwise launches instead of MIOpen's im2col/winograd pipeline for a 5-tap filter."""
        k = conv.kernel_size[0]
        if conv.stride[0] != 1:
            return conv(x)
        xp = F.pad(x, (k // 2, k // 2))
        return (xp.unfold(-1, k, 1) * conv.weight.view(1, 1, 1, k)).sum(-1)

    def _get_st_ed_prob(self, modularized_query, context_feat2, context_mask, cross=False):
        query = self.video_query_linear(modularized_query)
        if cross:
            sim = torch.einsum("md,nld->mnl", query, context_feat2)
            n_q, n_c, ln = sim.shape
            flat = sim.reshape(n_q * n_c, 1, ln)
            st = self._conv5(self.video_st_predictor, flat).view(n_q, n_c, ln)
            ed = self._conv5(self.video_ed_predictor, flat).view(n_q, n_c, ln)
            context_mask = context_mask.unsqueeze(0)
        else:
            sim = torch.einsum("bd,bld->bl", query, context_feat2).unsqueeze(1)
            st = self._conv5(self.video_st_predictor, sim).squeeze(1)
            ed = self._conv5(self.video_ed_predictor, sim).squeeze(1)
        m = context_mask.to(st.dtype)
        return mask_logits(st, m), mask_logits(ed, m)

    def encode_txt_inputs(self, input_ids, pos_ids, attn_masks, attn_layer=None, normalized=False):
        feats = self.v_encoder.f_encoder(
            {"input_ids": input_ids, "pos_ids": pos_ids, "attn_masks": attn_masks}, "txt")[0]
        if normalized:
            feats = F.normalize(feats.float(), dim=-1, eps=1e-5)
        if attn_layer is not None:
            return attn_layer(feats, attn_masks)
        return feats.float()

    def get_pred_from_mod_query(self, frame_embeddings, c_attn_masks, modularized_query, cross=False):
        cross = cross or frame_embeddings.shape[0] != modularized_query.shape[0]
        return self._get_st_ed_prob(modularized_query, frame_embeddings, c_attn_masks, cross=cross)

    def get_ranking_loss(self, pos_score, neg_score):
        if self.ranking_loss_type == "hinge":
            return torch.clamp(self.margin + neg_score - pos_score, min=0)
        if self.ranking_loss_type == "lse":
            return torch.log1p(torch.exp(neg_score - pos_score))
        raise NotImplementedError("Only support 'hinge' and 'lse'")

    def _weight_hard(self, loss):
        if not self.use_hard_negative:
            return loss
        w = torch.full_like(loss, 0.1)
        w[:, :self.hard_pool_size] = self.hard_neg_weight
        return w * loss

    def get_video_level_loss(self, query_context_scores, reduction="mean"):
        """Ranking loss over in-batch negatives (pretrain.py:203-292), vectorised."""
        nq, nv = query_context_scores.shape
        per = nq // nv
        dev = query_context_scores.device
        if nv == 1:
            return torch.tensor(0, device=dev), torch.tensor(0, device=dev)
        own = torch.arange(nq, device=dev) // per                    # video of each query
        is_pos = own.unsqueeze(1) == torch.arange(nv, device=dev).unsqueeze(0)
        # (select/where instead of indexed assignment: no host sync, hipGraph-capturable)
        pos = torch.where(is_pos, query_context_scores, torch.zeros_like(query_context_scores)).sum(1)
        masked = torch.where(is_pos, torch.full_like(query_context_scores, 999), query_context_scores)
        if self.use_all_neg:
            neg_ctx = masked.sort(dim=1, descending=True)[0][:, 1:]
            l_ctx = self._weight_hard(self.get_ranking_loss(pos.view(nq, 1), neg_ctx))
            neg_q = masked.t().sort(dim=1, descending=True)[0][:, per:]
            l_q = self.get_ranking_loss(pos.view(nv, per, 1), neg_q.unsqueeze(1))
            l_q = self._weight_hard(l_q.view(-1, l_q.size(2)))
        else:
            l_ctx = self.get_ranking_loss(
                pos.view(nq, 1), self.get_sampled_neg_scores(masked, 1).unsqueeze(-1))
            l_q = self.get_ranking_loss(
                pos.view(nv, per), self.get_sampled_neg_scores(masked.t(), per).unsqueeze(-1))
        if reduction == "sum":
            return l_ctx.mean(1), l_q.mean(1)
        if reduction == "mean":
            return l_ctx.mean(1).mean(0), l_q.mean(1).mean(0)
        if reduction is None:
            return l_ctx, l_q
        raise NotImplementedError(f"reduction {reduction} not supported")

    def get_sampled_neg_scores(self, scores_masked, sample_min_idx=1):
        bsz, n = scores_masked.shape
        assert n > sample_min_idx, "Unable to sample negative when bsz==sample_min_idx"
        order = scores_masked.argsort(dim=1, descending=True)
        hi = min(sample_min_idx + self.hard_pool_size, n) if self.use_hard_negative else n
        rows = torch.arange(bsz, device=scores_masked.device)
        pick = torch.randint(sample_min_idx, hi, (bsz,), device=scores_masked.device)
        return scores_masked[rows, order[rows, pick]]

    def get_video_level_scores(self, modularized_query, context_feat1, context_mask,
                               val_gather_gpus=True):
        q = F.normalize(modularized_query, dim=-1, eps=1e-5)
        ctx = F.normalize(context_feat1, dim=-1, eps=1e-5)
        gather = (self.training and self.gather_gpus) or (not self.training and val_gather_gpus)
        if gather and dist_utils.world_size() > 1:
            q, ctx, context_mask = dist_utils.gather_negatives(q, ctx, context_mask)
        scores = torch.einsum("md,nld->mln", q, ctx)
        m = context_mask.transpose(0, 1).unsqueeze(0).to(scores.dtype)
        return mask_logits(scores, m).max(dim=1)[0]

    def set_hard_negative(self, use_hard_negative, hard_pool_size, hard_neg_weight):
        self.use_hard_negative = use_hard_negative
        self.hard_pool_size = hard_pool_size
        self.hard_neg_weight = hard_neg_weight

    def set_train_st_ed(self, lw_st_ed):
        self.lw_st_ed = lw_st_ed
