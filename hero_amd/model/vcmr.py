"""HERO for video corpus moment retrieval (reference: model/vcmr.py) — a task router over the
VSM head."""
from .pretrain import HeroForPretraining

_TASKS = ("tvr", "how2r", "didemo_video_sub", "didemo_video_only")


class HeroForVcmr(HeroForPretraining):
    def forward(self, batch, task="tvr", compute_loss=True):
        if task not in _TASKS:
            raise ValueError(f"Unrecognized task {task}")
        return super().forward(batch, task="vsm", compute_loss=compute_loss)

    def get_pred_from_raw_query(self, frame_embeddings, c_attn_masks, query_input_ids, query_pos_ids,
                                query_attn_masks, cross=False, val_gather_gpus=False):
        mod_q = self.encode_txt_inputs(query_input_ids, query_pos_ids, query_attn_masks,
                                       attn_layer=self.q_feat_attn, normalized=False)
        st, ed = self.get_pred_from_mod_query(frame_embeddings, c_attn_masks, mod_q, cross=cross)
        scores = None
        if self.lw_neg_ctx != 0 or self.lw_neg_q != 0:
            scores = self.get_video_level_scores(mod_q, frame_embeddings, c_attn_masks, val_gather_gpus)
        return scores, st, ed
