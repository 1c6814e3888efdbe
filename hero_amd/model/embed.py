"""Input embeddings (reference: model/embed.py) — gathers, sums, LayerNorm and dropout run as ONE
HIP kernel per embedding (hero_layernorm_fwd with gathered tables)."""
import torch
from torch import nn

from .. import _lib as L
from .. import functional as HF
from .layers import LayerNorm, _drop


_ARANGE = {}


def _row_index(ids, rows, cols_per_row):
    """Per-output-row int32 table index from the reference's (1, L) or (S, L) id tensors.  A (1, L)
    tensor broadcast over the rows gives a PERIODIC index; that is recorded on the result
    (`_hero_period`) so the backward can sum the S repeats first instead of scatter-adding S-fold
    contended rows (functional.EmbedLnFn)."""
    if ids.dim() == 1:
        ids = ids.unsqueeze(0)
    period = cols_per_row if (ids.shape[0] == 1 and rows > 1) else 0
    if ids.shape[0] == 1 and rows > 1:
        ids = ids.expand(rows, -1)
    if ids.shape != (rows, cols_per_row):
        raise ValueError("index tensor of shape %s does not match (%d, %d)" %
                         (tuple(ids.shape), rows, cols_per_row))
    out = HF.memo("row_index", (ids,), lambda: ids.reshape(-1).to(torch.int32).contiguous(), (rows, cols_per_row),
                  spec=(L.DERIVE_I32, 0, 0, 0))
    out._hero_period = period
    return out


class SubEmbeddings(nn.Module):
    """LN(word[ids] + pos[pos_ids] + type[1]) -> dropout   (model/embed.py:28-58).
    Token-type id is 1 for text unless token_type_ids is given; positions come from the collate
    (0-based, data/data.py:427-429) — RoBERTa's padding-offset helpers are not on this path."""

    def __init__(self, config):
        super().__init__()
        self.padding_idx = 1
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size,
                                            padding_idx=self.padding_idx)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=1e-5)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids=None, position_ids=None, token_type_ids=None, inputs_embeds=None):
        if input_ids is None or inputs_embeds is not None:
            raise NotImplementedError("hero_amd SubEmbeddings takes input_ids (HERO never passes inputs_embeds)")
        S, Lt = input_ids.shape
        if position_ids is None:
            raise NotImplementedError("position_ids are always supplied by HERO's collate functions")
        wid = _row_index(input_ids, S, Lt)
        pid = _row_index(position_ids, S, Lt)
        if token_type_ids is None:
            type_tab, tid, type_row = self.token_type_embeddings.weight, 1, 1            # fixed row 1 of the table
        else:
            type_tab, tid, type_row = self.token_type_embeddings.weight, _row_index(token_type_ids, S, Lt), 0
        y = HF.embed_ln(None, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps,
                        _drop(self.dropout, input_ids.device), HF.compute_dtype(),
                        tables=(self.word_embeddings.weight, self.position_embeddings.weight, type_tab),
                        idxs=(wid, pid, tid), skip_idx=(self.padding_idx, -1, -1))
        return y.view(S, Lt, -1)


class ImageEmbeddings(nn.Module):
    """LN_768(Linear(LN_4352(feat [+ mask_emb])) + pos[img_pos] + type) -> dropout
    (model/embed.py:102-117)."""

    def __init__(self, config, img_dim, max_img_seq_len):
        super().__init__()
        self.img_linear = nn.Linear(img_dim, config.hidden_size)
        self.img_LayerNorm = LayerNorm(img_dim, eps=1e-5)
        self.position_embeddings = nn.Embedding(max_img_seq_len, config.hidden_size)
        self.mask_embedding = nn.Embedding(2, img_dim, padding_idx=0)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=1e-5)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, img_feat, type_embeddings, img_pos_ids=None, img_masks=None):
        S, Lv, Dv = img_feat.shape
        cd = HF.compute_dtype()
        if img_pos_ids is None:
            img_pos_ids = torch.arange(Lv, device=img_feat.device).unsqueeze(0)
        pid = _row_index(img_pos_ids, S, Lv)
        if img_masks is not None:
            # feat + mask_embedding[img_masks] fused into the first LayerNorm's input sum
            mid = img_masks.reshape(-1).to(torch.int32).contiguous()
            normed = HF.embed_ln(img_feat, self.img_LayerNorm.weight, self.img_LayerNorm.bias, 1e-5, None, cd,
                                 tables=(self.mask_embedding.weight,), idxs=(mid,), skip_idx=(0,))
        else:
            normed = HF.embed_ln(img_feat, self.img_LayerNorm.weight, self.img_LayerNorm.bias, 1e-5, None, cd)
        t = HF.linear(normed.view(S * Lv, Dv), self.img_linear.weight, self.img_linear.bias)
        if isinstance(type_embeddings, tuple):                    # (table parameter, fixed row): no autograd slice
            type_tab, type_idx = type_embeddings
        elif type_embeddings.numel() != t.shape[1]:
            raise NotImplementedError("per-token img_type_ids are not used by HERO (a single type row is)")
        else:
            type_tab, type_idx = type_embeddings.reshape(1, -1), None
        y = HF.embed_ln(t, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps,
                        _drop(self.dropout, img_feat.device), cd,
                        tables=(self.position_embeddings.weight, type_tab), idxs=(pid, type_idx))
        return y.view(S, Lv, -1)


class FrameEmbeddings(nn.Module):
    """LN(x + pos[arange(L)]) -> dropout   (model/embed.py:146-161)."""

    def __init__(self, config):
        super().__init__()
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=1e-5)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, frame_feat, position_ids=None):
        B, Lc, D = frame_feat.shape
        if position_ids is None:
            if Lc > self.position_embeddings.num_embeddings:       # nn.Embedding raises here; a raw kernel would read past the table
                raise IndexError("FrameEmbeddings: %d frames, the position table has %d rows (max_position_embeddings)"
                                 % (Lc, self.position_embeddings.num_embeddings))
            key = (B, Lc, str(frame_feat.device))
            pid = _ARANGE.get(key)
            if pid is None:
                pid = _ARANGE[key] = torch.arange(Lc, device=frame_feat.device, dtype=torch.int32).repeat(B)
                pid._hero_period = Lc if B > 1 else 0
        else:
            pid = _row_index(position_ids, B, Lc)
        x = HF.cast(frame_feat, HF.compute_dtype())
        y = HF.embed_ln(x, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps,
                        _drop(self.dropout, x.device), HF.compute_dtype(),
                        tables=(self.position_embeddings.weight,), idxs=(pid,))
        return y.view(B, Lc, D)


class QueryFeatEmbeddings(FrameEmbeddings):
    """Same computation as FrameEmbeddings (model/embed.py:164-188)."""
