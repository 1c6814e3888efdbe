"""Hierarchical video+language model (reference: model/model.py) on the HIP kernels."""
import json
import logging
from collections import defaultdict

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from .. import _lib as L
from .. import functional as HF
from .encoder import CrossModalTrm, RobertaModelConfig, RobertaPreTrainedModel, TemporalTrm
from .layers import GELU, LayerNorm, LinearLayer, MLPLayer
from .modeling_utils import load_partial_checkpoint, load_pretrained_weight

logger = logging.getLogger(__name__)


class VideoModelConfig(object):
    """f_config / c_config / q_config / d_config bundle (model/model.py:31-61)."""

    def __init__(self, config_json_file):
        assert isinstance(config_json_file, str)
        with open(config_json_file, "r", encoding="utf-8") as f:
            cfg = json.load(f)
        self.f_config = RobertaModelConfig.from_dict(cfg["f_config"])
        self.c_config = RobertaModelConfig.from_dict(cfg["c_config"])
        self.q_config = RobertaModelConfig.from_dict(cfg["q_config"]) if "q_config" in cfg else None
        self.d_config = RobertaModelConfig.from_dict(cfg["d_config"]) if "d_config" in cfg else None
        self.initializer_range = self.f_config.initializer_range

    @classmethod
    def from_json_file(cls, json_file):
        return cls(json_file)


class VideoPreTrainedModel(RobertaPreTrainedModel):
    def __init__(self, config, *inputs, **kwargs):
        if not isinstance(config, VideoModelConfig):
            raise ValueError("Parameter config in `%s(config)` should be an instance of class "
                             "`VideoModelConfig`." % self.__class__.__name__)
        super().__init__(config.f_config)
        self.config = config

    @classmethod
    def load_config(cls, config_file):
        return VideoModelConfig.from_json_file(config_file)

    @classmethod
    def from_pretrained(cls, config_file, state_dict, *inputs, **kwargs):
        model = cls(cls.load_config(config_file), *inputs, **kwargs)
        if state_dict == {}:
            logger.info("No pretrained weights loaded")
            return model
        return load_pretrained_weight(model, state_dict)


class FrameFeatureRegression(nn.Module):
    """Linear -> GELU -> LN -> Linear (model/model.py:104-114); `net.{0,2,3}` keep the names."""

    def __init__(self, hidden_size, feat_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(hidden_size, hidden_size), GELU(),
                                 LayerNorm(hidden_size, eps=1e-5), nn.Linear(hidden_size, feat_dim))

    def forward(self, x):
        x = HF.cast(x, HF.compute_dtype())
        h = HF.linear(x, self.net[0].weight, self.net[0].bias, act=L.ACT_GELU)
        h = self.net[2](h)
        return HF.cast(HF.linear(h, self.net[3].weight, self.net[3].bias), torch.float32)


def build_frame_map(num_subs, sub_idx2frame_idx, n_videos, n_frames, seq_len, device):
    """Host-side index build for collect_frame_outputs (model/model.py:156-187).

    Returns CSR (offsets, entries) over output frames [n_videos*n_frames] whose entries are flat
    rows of the (total_subs*seq_len) cross-modal output, and the inverse map (flat source row ->
    output row or -1).  Built once per batch on the host from the collate's python lists, instead
    of one H2D copy + index_put per subtitle."""
    dst, src = [], []
    base = 0
    for v, n in enumerate(num_subs):
        for sid, frames in sub_idx2frame_idx[v]:
            row = (base + sid) * seq_len
            for j, f in enumerate(frames):
                dst.append(v * n_frames + f)
                src.append(row + j)
        base += n
    total_src = base * seq_len
    if any(not 0 <= f < n_frames for _, rows in enumerate(sub_idx2frame_idx) for _, fr in rows for f in fr):
        raise IndexError("sub_idx2frame_idx names a frame outside [0, %d): the reference's index_put in "
                         "collect_frame_outputs (model/model.py:183) fails on it too" % n_frames)
    dst = np.asarray(dst, dtype=np.int64)
    src = np.asarray(src, dtype=np.int64)
    order = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=n_videos * n_frames)
    offsets = np.zeros(n_videos * n_frames + 1, dtype=np.int32)
    np.cumsum(counts, out=offsets[1:])
    entries = src[order].astype(np.int32)
    inverse = np.full(total_src, -1, dtype=np.int32)
    inverse[src] = dst.astype(np.int32)
    if len(np.unique(src)) != len(src):
        raise ValueError("a (subtitle, slot) pair is matched to more than one frame")
    if entries.size == 0:
        entries = np.zeros(1, dtype=np.int32)
    t = lambda a: torch.from_numpy(a).to(device, non_blocking=True)  # noqa: E731
    return t(offsets), t(entries), t(inverse)


class HierarchicalVlModel(VideoPreTrainedModel):
    def __init__(self, config, vfeat_dim, max_frm_seq_len, max_clip_len=100, nce_temp=1.0):
        super().__init__(config)
        self.f_encoder = CrossModalTrm(config.f_config, vfeat_dim, max_frm_seq_len)
        self.frame_transform = LinearLayer(vfeat_dim, config.f_config.hidden_size, layer_norm=True,
                                           dropout=config.f_config.hidden_dropout_prob, relu=True)
        self.c_encoder = TemporalTrm(config.c_config)
        self.feat_regress = FrameFeatureRegression(config.f_config.hidden_size, vfeat_dim)
        self.nce_temp = nce_temp
        self.mask_embedding = nn.Embedding(2, vfeat_dim, padding_idx=0)
        self.fom_output = MLPLayer(config.c_config.hidden_size, max_clip_len)
        self.register_buffer("pad", torch.zeros(8, config.c_config.hidden_size))
        self.output_fp32 = True       # hand fp32 to the task heads whatever the compute dtype
        self._frame_maps = {}

    def forward(self, batch, task="repr", compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        if task == "repr":
            return self.forward_repr(batch)
        if task.startswith("mlm"):
            return self.f_encoder(batch, task, compute_loss)
        if task == "mffr":
            return self.forward_mfm(batch, compute_loss, loss="regression")
        if task == "mfm-nce":
            return self.forward_mfm(batch, compute_loss, loss="nce")
        if task == "fom":
            return self.forward_fom(batch, compute_loss)
        raise ValueError(f"Unrecognized task {task}")

    # -- collect_frame_outputs ---------------------------------------------------------------------
    def _frame_map(self, num_subs, sub2frm, B, NF, Lf, device):
        key = (id(sub2frm), tuple(num_subs), B, NF, Lf, str(device))
        hit = self._frame_maps.get(key)
        if hit is None or hit[0] is not sub2frm:      # the entry keeps its source alive: ids are not reused
            if len(self._frame_maps) > 64:
                self._frame_maps.clear()
            hit = (sub2frm, build_frame_map(num_subs, sub2frm, B, NF, Lf, device))
            self._frame_maps[key] = hit
        return hit[1]

    def collect_frame_outputs(self, out_shape, frame_sequence_output, num_subs, sub_idx2frame_idx, frame_map=None):
        """frame_map: (offsets, entries, inverse) already on the device (hero_amd.collate.DeviceCollate) instead of
        the collate's python lists."""
        B, NF, D = out_shape
        Lf = frame_sequence_output.shape[1]
        offs, ent, inv = frame_map if frame_map is not None else \
            self._frame_map(num_subs, sub_idx2frame_idx, B, NF, Lf, frame_sequence_output.device)
        out = HF.CsrGatherSumFn.apply(frame_sequence_output, offs, ent, inv, B * NF)
        return out.view(B, NF, D)

    def forward_repr(self, batch, encode_clip=True, f_seq=None):
        """f_seq: optionally the cross-modal encoder output computed by the caller (see
        HeroForPretraining.forward, which runs subtitle rows and query rows through the layer stack
        in one pass)."""
        if f_seq is None:
            f_seq = self.f_encoder(batch, "repr")[0]                     # (total_subs, L_f, D)
        c_v_feats, c_attn_masks = batch["c_v_feats"], batch["c_attn_masks"]
        shape = list(c_v_feats.shape[:2]) + [f_seq.shape[-1]]
        matched = self.collect_frame_outputs(shape, f_seq, batch["num_subs"],
                                             batch["sub_idx2frame_idx"], frame_map=batch["frame_map"])
        # ReLU(Linear(drop(LN(c_v_feats)))) + matched, residual fused into the GEMM epilogue
        # (forward_mfm: + mask_embedding[c_v_masks], model/model.py:244-247, summed inside the LayerNorm kernel)
        mfm = batch.get("_c_v_mask_rows")
        if mfm is not None:
            fused = self.frame_transform(c_v_feats, residual=matched, tables=(self.mask_embedding.weight,), idxs=(mfm,), skip_idx=(0,))
        else:
            fused = self.frame_transform(c_v_feats, residual=matched)
        if not encode_clip:
            return HF.cast(fused, torch.float32) if self.output_fp32 else fused
        out = self.c_encoder(clip_level_frame_feat=fused, clip_level_pos_ids=None,
                             attention_mask=c_attn_masks)
        return HF.cast(out, torch.float32) if self.output_fp32 else out

    def forward_vsm(self, batch):
        clip_outputs = self.forward_repr(batch)
        q = self.f_encoder({"input_ids": batch["vsm_query_input_ids"],
                            "pos_ids": batch["vsm_query_pos_ids"],
                            "attn_masks": batch["vsm_query_attn_masks"]}, "txt")[0]
        return clip_outputs, q

    # -- pre-training heads (config 4) ---------------------------------------------------------------
    def _compute_masked_hidden(self, hidden, mask, invert=False):
        """Rows of `hidden` where mask (or ~mask) is set (model/model.py:_compute_masked_hidden).  The row list is
        derived once per batch object (torch.nonzero synchronises and cannot be captured in a hipGraph)."""
        def build():
            m = ~mask if invert else mask
            return torch.nonzero(m.reshape(-1), as_tuple=False).reshape(-1).to(torch.int32).contiguous()
        rows = HF.memo("mask_rows", (mask,), build, (invert,))
        return HF.GatherRowsFn.apply(hidden.reshape(-1, hidden.shape[-1]).contiguous(), None, rows)

    def forward_mfm(self, batch, compute_loss=True, loss="regression"):
        assert loss in ("regression", "nce")
        c_v_mask = batch["c_v_masks"]
        c_v_feats = batch["c_v_feats"]
        c_v_feats.masked_fill_(c_v_mask.unsqueeze(-1), 0)                 # model/model.py:244 (in place, like the reference)
        # model/model.py:245-247 rebinds batch['c_v_feats'] = c_v_feats + mask_embedding(c_v_masks) in its LOCAL copy of
        # the batch dict; here the embedding row is added inside frame_transform's LayerNorm kernel (no 33 MB temporary,
        # no aten embedding backward: 112 us per MFM micro-step, profiles/r03_kernel_stats_D3.csv), row 0 = padding_idx
        batch["_c_v_mask_rows"] = HF.memo("c_v_mask_rows", (c_v_mask,), lambda: c_v_mask.reshape(-1).to(torch.int32).contiguous())
        clip_outputs = self.forward_repr(batch)
        pred = self.feat_regress(self._compute_masked_hidden(clip_outputs, c_v_mask))
        neg = self.feat_regress(self._compute_masked_hidden(clip_outputs, c_v_mask, invert=True)) \
            if loss == "nce" else None
        if not compute_loss:
            return pred if loss == "regression" else (pred, neg)
        targets = batch["feat_targets"]
        if loss == "regression":
            return F.mse_loss(pred, targets, reduction="none")
        return self.mfm_nce(pred, targets, neg)

    def mfm_nce(self, masked_output, pos_output, neg_output, compute_loss=True):
        cd = HF.compute_dtype()
        cand = torch.cat([pos_output, neg_output], 0)
        n = cand.shape[0]
        if n % 8:                                               # GEMM column granularity; the loss ignores the padding
            cand = torch.cat([cand, cand.new_zeros(8 - n % 8, cand.shape[1])], 0)
        logits = HF.matmul_nt(HF.cast(masked_output, cd), HF.cast(cand, cd))      # [n_masked, n_masked + n_neg (+pad)]
        if not compute_loss:
            return logits[:, :n].float()
        tgt = torch.arange(masked_output.size(0), device=logits.device)
        return HF.cross_entropy(logits, tgt, ncols=n, inv_temp=1.0 / self.nce_temp)

    def forward_fom(self, batch, compute_loss=True):
        order = batch["shuffled_orders"]
        feats = self.forward_repr(batch, encode_clip=False)               # (B, L, D) fp32
        B, Lc, D = feats.shape
        # out[b, order[b, i]] = feats[b, i]  (scatter_ of a permutation) as a row gather
        base = torch.arange(B, device=order.device).unsqueeze(1) * Lc
        inv = torch.full((B * Lc,), -1, dtype=torch.int32, device=order.device)
        inv[(base + order).reshape(-1)] = (base + torch.arange(Lc, device=order.device)
                                           ).reshape(-1).to(torch.int32)
        shuffled = HF.GatherRowsFn.apply(feats.reshape(B * Lc, D), None, inv).view(B, Lc, D)
        enc = self.c_encoder(clip_level_frame_feat=shuffled, clip_level_pos_ids=None,
                             attention_mask=batch["c_attn_masks"])
        logits = HF.cast(self.fom_output(enc.reshape(B * Lc, D)), torch.float32)
        if compute_loss:                                        # mean over the rows that carry a target
            tgt = batch["targets"].reshape(-1)
            rows = HF.cross_entropy(logits, tgt, ignore_index=-1)
            return rows.sum() / (tgt != -1).sum().clamp(min=1)
        return logits

    def initialize(self):
        self.apply(self.init_weights)
        self.f_encoder.apply(self.f_encoder.init_weights)
        self.c_encoder.apply(self.c_encoder.init_weights)

    def init_type_embedding(self):
        self.f_encoder.init_type_embedding()
        self.mask_embedding.weight.data[0].fill_(0)


class HeroModel(VideoPreTrainedModel):
    def __init__(self, config, vfeat_dim, max_frm_seq_len):
        super().__init__(config)
        self.config = config
        self.v_encoder = HierarchicalVlModel(config, vfeat_dim, max_frm_seq_len)
        self.v_encoder.initialize()

    def load_partial_pretrained(self, checkpoint, vfeat_dim, max_frm_seq_len, skip_layers=True):
        partial = load_partial_checkpoint(checkpoint, self.config.f_config.num_hidden_layers,
                                          skip_layers)
        self.v_encoder.f_encoder = CrossModalTrm.from_pretrained(
            self.config.f_config, state_dict=partial, vfeat_dim=vfeat_dim,
            max_img_seq_len=max_frm_seq_len)
        self.v_encoder.f_encoder.pad_vocab()
        self.v_encoder.init_type_embedding()
