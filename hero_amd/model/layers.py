"""Transformer building blocks with the reference's module / parameter names
(linjieli222/HERO model/layers.py) whose forwards run on the libhero_hip.so kernels.

State-dict keys are identical to the reference (separate query/key/value Linear parameters,
`LayerNorm.weight/bias`, `dense.weight/bias`, `net.1.weight`), so its checkpoints load unchanged.
`nn.Dropout` children are kept as configuration holders (utils/misc.py:set_dropout rewrites `.p`);
the dropout itself is applied inside the HIP kernels by a counter-based RNG.
"""
import torch
from torch import nn

from .. import _lib as L
from .. import functional as HF


class LayerNorm(nn.Module):
    """Parameter holder + standalone forward for apex FusedLayerNorm call sites."""

    def __init__(self, size, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(size))
        self.bias = nn.Parameter(torch.zeros(size))
        self.eps = eps
        self.normalized_shape = (size,)

    def forward(self, x, dropout=None, out_dtype=None, tables=(), idxs=(), skip_idx=None):
        """tables / idxs / skip_idx: embedding rows added to x in front of the normalisation (HF.embed_ln)."""
        shp = x.shape
        y = HF.embed_ln(x, self.weight, self.bias, self.eps, dropout,
                        out_dtype or (x.dtype if x.dtype == torch.bfloat16 else HF.compute_dtype()),
                        tables=tables, idxs=idxs, skip_idx=skip_idx)
        return y.view(shp)


BertLayerNorm = LayerNorm


def _drop(mod, device):
    return HF.RNG.make(mod.p, mod.training, device)


class GELU(nn.Module):
    """model/layers.py:64-67 — only ever used right behind an nn.Linear; see MLPLayer/regression
    heads, which fuse it into that Linear's epilogue."""

    def forward(self, x):  # pragma: no cover - kept for API parity
        raise RuntimeError("hero_amd: GELU is fused into the preceding Linear (use linear_gelu)")


class LinearLayer(nn.Module):
    """LN -> Dropout -> Linear -> ReLU (model/layers.py:70-93); optional fused residual."""

    def __init__(self, in_hsz, out_hsz, layer_norm=True, dropout=0.1, relu=True):
        super().__init__()
        self.relu = relu
        self.layer_norm = layer_norm
        if layer_norm:
            self.LayerNorm = LayerNorm(in_hsz, eps=1e-5)
        self.net = nn.Sequential(nn.Dropout(dropout), nn.Linear(in_hsz, out_hsz))

    def forward(self, x, residual=None, tables=(), idxs=(), skip_idx=None):
        cd = HF.compute_dtype()
        if self.layer_norm:
            x = self.LayerNorm(x, dropout=_drop(self.net[0], x.device), out_dtype=cd, tables=tables, idxs=idxs, skip_idx=skip_idx)
        elif tables:
            raise NotImplementedError("LinearLayer(layer_norm=False) with embedding tables")
        else:
            if self.net[0].training and self.net[0].p > 0:
                raise NotImplementedError("LinearLayer(layer_norm=False) with dropout")
            x = HF.cast(x, cd)
        lin = self.net[1]
        if residual is not None:
            residual = HF.cast(residual, cd)
        return HF.linear(x, lin.weight, lin.bias, act=L.ACT_RELU if self.relu else L.ACT_NONE,
                         residual=residual)


class MLPLayer(nn.Module):
    """model/layers.py:48-61: Linear -> gelu -> LN -> Linear."""

    def __init__(self, in_hsz, out_hsz):
        super().__init__()
        self.linear_1 = nn.Linear(in_hsz, in_hsz * 2)
        self.LayerNorm = LayerNorm(in_hsz * 2, eps=1e-5)
        self.linear_2 = nn.Linear(in_hsz * 2, out_hsz)

    def forward(self, x):
        x = HF.cast(x, HF.compute_dtype())
        h = HF.linear(x, self.linear_1.weight, self.linear_1.bias, act=L.ACT_GELU)
        h = self.LayerNorm(h)
        return HF.linear(h, self.linear_2.weight, self.linear_2.bias)


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        if self.attention_head_size != 64:
            raise ValueError("hero_amd attention kernels are specialised for head size 64 "
                             "(HERO-base); got %d" % self.attention_head_size)
        self.all_head_size = config.hidden_size
        self.output_attentions = getattr(config, "output_attentions", False)
        if self.output_attentions:
            raise NotImplementedError("output_attentions is not supported by the fused attention")
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def qkv_params(self):
        return (self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                self.value.weight, self.value.bias)

    def forward(self, hidden_states, attention_mask=None, head_mask=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is always None in HERO (model/layers.py:311)")
        x = HF.cast(hidden_states, HF.compute_dtype())
        S, Lq, _ = x.shape
        m = HF.as_mask_add(attention_mask, S, Lq)
        return (HF.SelfAttentionFn.apply(x, m, self.num_attention_heads,
                                         _drop(self.dropout, x.device), *self.qkv_params()),)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return HF.ProjResLnFn.apply(hidden_states, input_tensor, self.LayerNorm.eps,
                                    _drop(self.dropout, hidden_states.device), self.dense.weight,
                                    self.dense.bias, self.LayerNorm.weight, self.LayerNorm.bias)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask=None, head_mask=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is always None in HERO")
        x = HF.cast(input_tensor, HF.compute_dtype())
        S, Lq, _ = x.shape
        m = HF.as_mask_add(attention_mask, S, Lq)
        return (self.forward_rows(x, ((S, Lq),), (m,)),)

    def forward_rows(self, x, segs, masks):
        """x: [rows, D] or (S, L, D) holding the sequence groups `segs` = ((S, L), ...) stacked along
        the rows; masks: additive fp32 [S, L] per group."""
        so, o = self.self, self.output
        return HF.AttnBlockFn.apply(
            x, tuple(segs), tuple(masks), so.num_attention_heads, o.LayerNorm.eps,
            tuple(_drop(so.dropout, x.device) for _ in segs), _drop(o.dropout, x.device),
            *so.qkv_params(), o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError("hero_amd implements the erf GELU HERO uses (hidden_act='gelu')")
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return HF.linear(hidden_states, self.dense.weight, self.dense.bias, act=L.ACT_GELU)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    forward = BertSelfOutput.forward


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask=None, head_mask=None):
        a = self.attention(hidden_states, attention_mask, head_mask)[0]
        return (self._ffn(a),)

    def _ffn(self, a):
        o = self.output
        return HF.FfnBlockFn.apply(a, o.LayerNorm.eps, _drop(o.dropout, a.device),
                                   self.intermediate.dense.weight, self.intermediate.dense.bias,
                                   o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias)

    def forward_rows(self, x, segs, masks):
        return self._ffn(self.attention.forward_rows(x, segs, masks))


class BertPooler(nn.Module):
    """model/layers.py:275-287.  Dead value on the 'repr'/'txt' paths (its output is discarded by
    every caller), so the encoders do not run it unless asked; kept for state-dict parity."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        first = hidden_states[:, 0].contiguous()
        return self.activation(HF.cast(HF.linear(first, self.dense.weight, self.dense.bias),
                                       torch.float32))


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        if getattr(config, "output_attentions", False) or getattr(config, "output_hidden_states", False):
            raise NotImplementedError("output_attentions / output_hidden_states are off in HERO")
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask=None, head_mask=None):
        if (self.pack_ragged and BertEncoder.allow_packing and BertEncoder.packing_mode != "none" and attention_mask is not None
                and hidden_states.is_cuda and attention_mask.dim() == 2 and attention_mask.dtype != torch.float32):
            return (self.forward_multi([hidden_states], [attention_mask])[0],)
        x = HF.cast(hidden_states, HF.compute_dtype())
        S, Lq, _ = x.shape
        m = HF.as_mask_add(attention_mask, S, Lq)        # (1-m)*-10000, model/layers.py:299-302
        m4 = m.view(S, 1, 1, Lq) if m is not None else None
        for layer in self.layer:
            x = layer(x, m4, None)[0]
        return (x,)

    # Drop masked positions before the layer stack when that saves > 10 % of the rows.  Only valid where
    # nothing downstream reads the outputs at masked positions: true for the cross-modal encoder
    # (CrossModalTrm turns it on), NOT for the temporal one - the start/end Conv1d of the VSM head mixes
    # the (finite garbage) outputs at padded frames into the logits of the last valid frames
    # (model/pretrain.py:128-166), so those rows must keep the reference's values.
    pack_ragged = False
    _PLANS = {}
    allow_packing = True    # global switch (tests compare against the padded formulation)

    @staticmethod
    def _pack_plan(mask_list, lens, max_len=64):
        """Row maps between the padded layout (groups stacked, row = position) and the packed one
        (masked positions removed, sequences back to back).  None when (almost) every position is
        valid.  One small device->host read per distinct batch; cached on the mask tensors."""
        key = tuple((m.data_ptr(), m._version, tuple(m.shape)) for m in mask_list) + (max_len,)
        hit = BertEncoder._PLANS.get(key)
        if hit is not None:
            return hit[0]
        dev = mask_list[0].device
        flat = torch.cat([(m != 0).reshape(-1) for m in mask_list]).cpu()
        total = flat.numel()
        valid = int(flat.sum())
        plan = None
        counts = torch.cat([(m != 0).sum(1).reshape(-1) for m in mask_list]).cpu()
        lmax = int(counts.max()) if counts.numel() else 0
        if 0 < valid < 0.9 * total and lmax <= max_len:    # the variable-length attention kernels' limit
            gather = torch.nonzero(flat, as_tuple=False).reshape(-1).to(torch.int32)
            inverse = torch.full((total,), -1, dtype=torch.int32)
            inverse[gather.long()] = torch.arange(valid, dtype=torch.int32)
            off = torch.zeros(counts.numel() + 1, dtype=torch.int32)
            off[1:] = torch.cumsum(counts, 0).to(torch.int32)
            inv, back, r0 = [], [], 0
            for n in lens:                                   # per group: padded -> packed and packed -> padded
                inv.append(inverse[r0:r0 + n].contiguous().to(dev))
                g = gather - r0
                back.append(torch.where((g >= 0) & (g < n), g, torch.full_like(g, -1)).to(dev))
                r0 += n
            plan = (gather.to(dev), inverse.to(dev), inv, back, off.to(dev), int(counts.numel()), lmax)
        if len(BertEncoder._PLANS) > 64:
            BertEncoder._PLANS.clear()
        BertEncoder._PLANS[key] = (plan, mask_list)          # keep the masks alive: ids stay unique
        return plan

    # ---- a pack plan with FIXED buffer sizes, refreshed from the host (round 6) ---------------------------------------
    # The plan above is derived from the masks on the host per batch and fixes the packed row COUNT: a captured hipGraph
    # would replay the capture batch's plan.  A static plan has a row CAPACITY instead (a bucket, >= the valid rows of
    # every batch it serves): the rows behind the valid ones are zero rows grouped into pad "sequences" of <= PAD_CHUNK
    # rows, so every row of every per-layer buffer is written by some kernel (a row no kernel writes would hand
    # uninitialised memory to the weight-gradient reduction), their upstream gradient is exactly zero, and nothing
    # downstream reads them.  All of it - gather / inverse / per-group maps / sequence offsets - lives in ONE int32
    # buffer whose contents a feeder (hero_amd.loader.StaticBatchFeeder(packed_rows=...)) rebuilds on the host from the
    # next batch's masks and copies in; shapes, sequence count and the attention length class never change.
    PAD_CHUNK = 32
    _STATIC_PLANS = {}      # addresses of the mask tensors of the groups -> (plan tuple, masks kept alive)
    packing_mode = "all"    # "all" | "static" (registered static plans only: batches whose buffers are rewritten) | "none"

    @staticmethod
    def static_plan_layout(group_shapes, rows_cap, min_rows=0, lmax=None):
        """Offsets (int32 elements) of the sections of a static plan's buffer.  group_shapes: ((S, L), ...) of the padded
        groups; rows_cap: packed rows (valid + pad); min_rows: the smallest number of valid rows this plan will see (bounds the
        number of pad sequences); lmax: attention length class (default: the longest group row, at most 64)."""
        r4 = lambda n: (n + 3) & ~3            # noqa: E731
        total = sum(S * L_ for S, L_ in group_shapes)
        n_real = sum(S for S, _ in group_shapes)
        n_pad = -(-max(rows_cap - min_rows, 0) // BertEncoder.PAD_CHUNK)
        lay, off = {}, 0
        for name, n in [("gather", rows_cap), ("inverse", total)] + [("back%d" % g, rows_cap) for g in range(len(group_shapes))] + \
                [("off", n_real + n_pad + 1)]:
            lay[name] = (off, n)
            off += r4(n)
        lay.update(size=off, total=total, n_real=n_real, n_pad=n_pad, n_seq=n_real + n_pad, rows_cap=rows_cap, min_rows=min_rows,
                   groups=tuple(tuple(g) for g in group_shapes),
                   lmax=lmax or min(64, max(max(L_ for _, L_ in group_shapes), BertEncoder.PAD_CHUNK)))
        return lay

    @staticmethod
    def fill_static_plan(flat, lay, masks):
        """Write the plan of one batch into `flat` (a numpy int32 array of lay['size'] elements: pinned host memory).
        masks: one 0/1 array [S, L] per group, already padded to the layout's group shapes.  Same maps as `_pack_plan`
        (valid positions in row-major order, groups back to back), + the pad rows.  Returns the number of valid rows."""
        import numpy as np
        cap, chunk = lay["rows_cap"], BertEncoder.PAD_CHUNK
        valid_flat = np.concatenate([np.asarray(m).reshape(-1) != 0 for m in masks])
        counts = np.concatenate([(np.asarray(m) != 0).sum(1).reshape(-1) for m in masks]).astype(np.int64)
        if valid_flat.size != lay["total"] or counts.size != lay["n_real"]:
            raise ValueError("static pack plan: masks of %d positions / %d rows, the plan was laid out for %d / %d" %
                             (valid_flat.size, counts.size, lay["total"], lay["n_real"]))
        valid = int(valid_flat.sum())
        if not lay["min_rows"] <= valid <= cap:
            raise ValueError("static pack plan: %d valid rows outside this bucket's [%d, %d]" % (valid, lay["min_rows"], cap))
        if counts.size and int(counts.max()) > lay["lmax"]:
            raise ValueError("static pack plan: a sequence of %d valid positions, the attention class is %d" % (int(counts.max()), lay["lmax"]))
        sec = lambda name: flat[lay[name][0]:lay[name][0] + lay[name][1]]      # noqa: E731
        gather = np.flatnonzero(valid_flat).astype(np.int32)
        g = sec("gather")
        g[:valid] = gather
        g[valid:] = -1
        inv = sec("inverse")
        inv[:] = -1
        inv[gather] = np.arange(valid, dtype=np.int32)
        r0 = 0
        for gi, (S, L_) in enumerate(lay["groups"]):
            b = sec("back%d" % gi)
            rel = gather.astype(np.int64) - r0
            b[:valid] = np.where((rel >= 0) & (rel < S * L_), rel, -1).astype(np.int32)
            b[valid:] = -1
            r0 += S * L_
        off = sec("off")
        off[0] = 0
        off[1:lay["n_real"] + 1] = np.cumsum(counts)
        pad = cap - valid                                    # pad rows as sequences of <= chunk rows, the rest empty
        steps = np.minimum(np.arange(1, lay["n_pad"] + 1, dtype=np.int64) * chunk, pad)
        off[lay["n_real"] + 1:] = valid + steps
        assert lay["n_pad"] * chunk >= pad and int(off[-1]) == cap
        return valid

    @staticmethod
    def register_static_plan(masks, flat, lay):
        """flat: the DEVICE int32 buffer (lay['size'] elements) a feeder refreshes; masks: the device mask tensors of the
        groups, in forward_multi's order - the plan is found by their addresses (their contents / versions change per batch)."""
        v = lambda name: flat[lay[name][0]:lay[name][0] + lay[name][1]]      # noqa: E731
        inverse = v("inverse")
        inv, r0 = [], 0
        for S, L_ in lay["groups"]:
            inv.append(inverse[r0:r0 + S * L_])
            r0 += S * L_
        plan = (v("gather"), inverse, inv, [v("back%d" % g) for g in range(len(lay["groups"]))], v("off"), lay["n_seq"], lay["lmax"])
        BertEncoder._STATIC_PLANS[tuple(m.data_ptr() for m in masks)] = (plan, list(masks), lay)
        return plan

    @staticmethod
    def _static_plan_for(mask_list, segs):
        hit = BertEncoder._STATIC_PLANS.get(tuple(m.data_ptr() for m in mask_list))
        if hit is None or hit[2]["groups"] != tuple(segs):
            return None
        return hit[0]

    def forward_multi(self, hidden_list, mask_list):
        """Run several independent sequence groups (tensors (S_i, L_i, D) + their (S_i, L_i) 0/1
        masks) through the SAME layer stack as one stacked row batch: GEMMs, LayerNorms and their
        backward kernels are launched once for all groups, attention once per group.  Numerically
        identical to calling forward() per group (every other op is row-wise).

        Ragged batches (SURVEY 8d 'ragged' variant; every real TVR batch): masked positions carry no
        information for the valid ones (their keys get exp(-1e4) = 0 weight, model/layers.py:299-302)
        and the reference's outputs at those positions are unused garbage, so they are dropped before
        the stack - the rows are PACKED, attention runs variable-length (hero_attention seq_off) and
        the result is scattered back to the padded layout with zeros at the masked positions."""
        cd = HF.compute_dtype()
        xs = [HF.cast(h, cd) for h in hidden_list]
        segs = tuple((x.shape[0], x.shape[1]) for x in xs)
        D = xs[0].shape[-1]
        x = HF.StackRowsFn.apply(*[t.reshape(-1, D) for t in xs]) if len(xs) > 1 else xs[0].reshape(-1, D)
        plan = None
        mode = BertEncoder.packing_mode if BertEncoder.allow_packing else "none"
        if self.pack_ragged and mode != "none" and all(m is not None for m in mask_list) and x.is_cuda:
            max_len = L.lib().hero_attention_max_packed_len(L.BF16 if cd == torch.bfloat16 else L.F32)
            plan = self._static_plan_for(mask_list, segs)
            if plan is not None and plan[6] > max_len:
                plan = None
            if plan is None and mode == "all":
                plan = self._pack_plan(mask_list, [s[0] * s[1] for s in segs], max_len=max_len)
        if plan is not None:
            gather, inverse, inv, back, off, n_seq, lmax = plan
            x = HF.PermuteRowsFn.apply(x.contiguous(), gather, inverse)
            packed = (("packed", n_seq, lmax, off),)
            for layer in self.layer:
                x = layer.forward_rows(x, packed, (None,))
            return [HF.PermuteRowsFn.apply(x, i_, b_).view(S, Lq, D) for i_, b_, (S, Lq) in zip(inv, back, segs)]
        masks = tuple(HF.as_mask_add(m, s[0], s[1]) for m, s in zip(mask_list, segs))
        for layer in self.layer:
            x = layer.forward_rows(x, segs, masks)
        if len(segs) == 1:
            return [x.view(segs[0][0], segs[0][1], D)]
        blocks = HF.SplitRowsFn.apply(x, *[S * Lq for (S, Lq) in segs])
        outs = [b.view(S, Lq, D) for b, (S, Lq) in zip(blocks, segs)]
        # the gradient of the first (large) block can be produced with room for the others behind it (functional.SplitRowsFn)
        outs[0]._hero_tail_rows = sum(S * Lq for (S, Lq) in segs[1:])
        return outs


class BertLMPredictionHead(nn.Module):
    """model/layers.py:330-354; decoder weight tied to the word embedding."""

    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=1e-5)
        self.decoder = nn.Linear(bert_model_embedding_weights.size(1),
                                 bert_model_embedding_weights.size(0), bias=False)
        self.decoder.weight = bert_model_embedding_weights
        self.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))

    def forward(self, hidden_states, raw=False):
        """raw: the logits in the compute dtype (input of the fused cross-entropy) instead of an fp32 copy."""
        x = HF.cast(hidden_states, HF.compute_dtype())
        h = HF.linear(x, self.dense.weight, self.dense.bias, act=L.ACT_GELU)
        h = self.LayerNorm(h)
        logits = HF.linear(h, self.decoder.weight, self.bias)
        return logits if raw else HF.cast(logits, torch.float32)
