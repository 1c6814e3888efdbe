"""BASELINE.json configurations at (or near) full size on a real MI355X, through the C ABI, against the
CPU oracle (oracle == reference: tests/test_oracle_golden.py):

  * configs[1]: HERO-base, bf16, the FULL D2 batch the bench runs (32 videos: the benched kernels -
    wave-specialised 192x192 GEMMs, matrix-core attention, fused LayerNorm - are the ones under test):
    forward, the three VSM losses and a handful of gradients, with the bf16 tolerance stated below;
  * configs[4] shapes: 256-frame Temporal Transformer (64 subtitles x 4 frames per video), regular and ragged,
    fp32 forward against the oracle, bf16 against fp32, bf16 gradients against the oracle's;
  * the hipGraph regression of round 1 (7a0be53): >= 300 replayed steps with a device-wide synchronise in
    the middle must converge like the eager run.
"""
import json

import pytest
import torch

from oracle import hero_oracle as O
from tests.util import rel_err, to_dev

pytestmark = pytest.mark.gpu

# bf16 storage (2^-9 relative per rounding) through 9 post-LN layers.  Round 3: set from the measured per-layer error
# growth (test_bf16_error_growth_per_layer prints it: L2 error 2.0e-3 after the embeddings, 4.0 / 5.3 / 6.3 / 7.1 / 7.9 /
# 8.6e-3 after 1..6 cross-modal layers, 9.5e-3 after the 3 temporal ones; max-norm 0.5e-2 -> 1.8e-2) with ~1.6x headroom;
# round 2 allowed 6e-2 / 3e-2.
BF16_MAX_TOL = 3e-2      # max |a-b| / max |b| under the mask (same metric as test_gpu_parity)
BF16_L2_TOL = 1.5e-2     # ||a-b||_2 / ||b||_2 under the mask: an element-weighted relative error
FP32_TOL = 5e-4          # north_star: 1e-3 relative fp32

HERO_BASE = {
    "f_config": dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                     hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                     max_position_embeddings=514, num_attention_heads=12, num_hidden_layers=6,
                     type_vocab_size=2, vocab_size=2048),
    "c_config": dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                     hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                     max_position_embeddings=514, num_attention_heads=12, num_hidden_layers=3,
                     type_vocab_size=2),
    "q_config": dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                     hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                     num_attention_heads=12, max_position_embeddings=514, num_hidden_layers=0,
                     type_vocab_size=1, vocab_size=2048)}


def l2_err(a, b, mask=None):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if mask is not None:
        m = mask.bool().cpu()
        a, b = a[m], b[m]
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def hero_base(seed=0, cls=None, **head):
    """HERO-base with a 2048-word vocabulary (the embedding table is a lookup, not on the GEMM path) and
    non-trivial LayerNorm / bias parameters; returns (cpu state dict, cuda model in train mode, p = 0)."""
    from hero_amd.model import HeroForVcmr
    from hero_amd.utils.misc import set_dropout
    HeroForVcmr = cls or HeroForVcmr
    path = "/tmp/hero_base_small_vocab_cfg.json"
    with open(path, "w") as f:
        json.dump(HERO_BASE, f)
    torch.manual_seed(seed)
    args = dict(lw_neg_ctx=8.0, lw_neg_q=8.0, lw_st_ed=0.01, ranking_loss_type="hinge", use_hard_negative=False,
                hard_pool_size=20, margin=0.1, use_all_neg=True, drop_svmr_prob=0.0)
    args.update(head)
    model = HeroForVcmr.from_pretrained(path, {}, vfeat_dim=4352, max_frm_seq_len=100, **args)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    set_dropout(model, 0.0)
    return P, model


GRAD_NAMES = ["v_encoder.f_encoder.encoder.layer.0.attention.self.query.weight",
              "v_encoder.f_encoder.encoder.layer.5.intermediate.dense.weight",
              "v_encoder.f_encoder.encoder.layer.2.output.dense.bias",
              "v_encoder.f_encoder.encoder.layer.3.attention.output.LayerNorm.weight",
              "v_encoder.f_encoder.img_embeddings.img_linear.weight",
              "v_encoder.c_encoder.encoder.layer.1.attention.self.value.weight",
              "v_encoder.c_encoder.encoder.layer.2.output.dense.weight",
              "v_encoder.frame_transform.net.1.weight",
              "video_query_linear.weight"]


def probe(shape, mask, seed=21):
    """Fixed random weights for the smooth part of the gradient objective (zero at padded frames)."""
    R = torch.randn(shape, generator=torch.Generator().manual_seed(seed))
    return R * mask.unsqueeze(-1).float()


def objective(l_st_ed, frames, R):
    """J = start/end cross-entropy (through queries, both encoders and the head) + <frames, R> / #frames.
    The ranking losses are left out of the GRADIENT check on purpose: max(0, margin + s_neg - s_pos) has a
    gradient that jumps with the sign of its argument, and at random initialisation most of the B x B score
    differences sit within bf16 noise of the margin (measured: 35 % gradient difference at 2e-4 loss
    difference).  Their VALUES are checked."""
    return l_st_ed.mean() + (frames.float() * R.to(frames.device)).sum() / R.shape[0] / R.shape[1]


def oracle_losses_and_grads(batch, P, names):
    cfg = O.cfg_from_json(HERO_BASE)
    Pq = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("pad")) for k, v in P.items()}
    losses = O.vsm_losses(batch, Pq, cfg)
    frames = O.forward_repr(batch, Pq, cfg)
    R = probe(frames.shape, batch["c_attn_masks"])
    objective(losses[0], frames, R).backward()
    return [l.detach() for l in losses], {n: Pq[n].grad for n in names}, frames.detach(), R


def hip_report(model, b, batch, ref_frames_for_fwd, ref_losses, ref_grads, R, names):
    with torch.no_grad():
        frames = model.v_encoder(b, "repr")
    report = {"repr.max": rel_err(frames, ref_frames_for_fwd, batch["c_attn_masks"]),
              "repr.l2": l2_err(frames, ref_frames_for_fwd, batch["c_attn_masks"])}
    losses = model(b, task="tvr", compute_loss=True)
    for k, got, want in zip(("st_ed", "neg_ctx", "neg_q"), losses, ref_losses):
        report["loss." + k] = abs(float(got.detach()) - float(want)) / (abs(float(want)) + 1e-4)
    objective(losses[0], model.v_encoder(b, "repr"), R).backward()
    params = dict(model.named_parameters())
    for n in names:
        report["grad." + n] = l2_err(params[n].grad, ref_grads[n])
    print(json.dumps(report, indent=1))
    return report


def test_hero_base_bf16_full_d2_batch_vs_oracle():
    """configs[1] at the size the bench runs, in the dtype the bench runs."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.synth import make_batch
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    P, model = hero_base()
    batch = make_batch("D2", vocab=2048, seed=7)
    ref_losses, ref_grads, ref_frames, R = oracle_losses_and_grads(batch, P, GRAD_NAMES)
    b = to_dev(batch, "cuda")
    report = hip_report(model, b, batch, ref_frames, ref_losses, ref_grads, R, GRAD_NAMES)
    assert report["repr.max"] < BF16_MAX_TOL and report["repr.l2"] < BF16_L2_TOL, report
    assert all(v < 2e-2 for k, v in report.items() if k.startswith("loss.")), report
    # bf16 activations + bf16 activation gradients through 9 layers (relative L2, printed above).  Measured in round 4:
    # 0.7 - 1.4 % for eight of the nine probes, 3.8 % for frame_transform.net.1.weight (its input is the bf16-rounded
    # LayerNorm of 4352 raw features, its output gradient the sum of the whole temporal stack's): gate 4 % (was 6 %),
    # 2 % for the others
    grads = {k: v for k, v in report.items() if k.startswith("grad.")}
    assert all(v < 0.04 for v in grads.values()), report
    assert all(v < 0.02 for k, v in grads.items() if "frame_transform" not in k), report


@pytest.mark.parametrize("formulation", ["packed", "padded"])
def test_hero_base_bf16_full_ragged_d2_batch_vs_oracle(formulation):
    """VERDICT r5 weak #1a: the workload behind `secondary.D2r` - the RAGGED TVR batch at 32 videos, HERO-base, bf16 - against
    the oracle, in both formulations the product runs: packed (valid rows only through the six cross-modal layers:
    ~14 000 GEMM rows on the 128 x 192 / 64-row tile choices, variable-length attention in two length classes incl. the
    two-wave 64-row kernels) and padded (what a feeder-owned batch runs).  Same gates as the regular batch."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.model.layers import BertEncoder
    from hero_amd.synth import make_batch
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    P, model = hero_base()
    batch = make_batch("D2", vocab=2048, seed=7, ragged=True)
    assert batch["c_v_feats"].shape[0] == 32
    n_valid, n_pad = int(batch["f_attn_masks"].sum()), batch["f_attn_masks"].numel()
    assert n_valid > 10000 and n_pad > 1.3 * n_valid                       # really ragged: > 30 % of the positions are padding
    assert int(batch["f_attn_masks"].sum(1).max()) > 32                    # ... with subtitles in the 64-row attention class
    ref_losses, ref_grads, ref_frames, R = oracle_losses_and_grads(batch, P, GRAD_NAMES)
    b = to_dev(batch, "cuda")
    BertEncoder.allow_packing = formulation == "packed"
    try:
        report = hip_report(model, b, batch, ref_frames, ref_losses, ref_grads, R, GRAD_NAMES)
    finally:
        BertEncoder.allow_packing = True
    assert report["repr.max"] < BF16_MAX_TOL and report["repr.l2"] < BF16_L2_TOL, report
    assert all(v < 2e-2 for k, v in report.items() if k.startswith("loss.")), report
    grads = {k: v for k, v in report.items() if k.startswith("grad.")}
    assert all(v < 0.04 for v in grads.values()), report
    assert all(v < 0.02 for k, v in grads.items() if "frame_transform" not in k), report


D3_GRADS = {
    "mfm-nce": ["v_encoder.feat_regress.net.0.weight", "v_encoder.feat_regress.net.3.weight", "v_encoder.mask_embedding.weight",
                "v_encoder.f_encoder.img_embeddings.mask_embedding.weight", "v_encoder.c_encoder.encoder.layer.0.intermediate.dense.weight",
                "v_encoder.f_encoder.encoder.layer.5.intermediate.dense.weight"],
    "fom": ["v_encoder.fom_output.linear_1.weight", "v_encoder.fom_output.linear_2.weight", "v_encoder.fom_output.LayerNorm.weight",
            "v_encoder.frame_transform.net.1.weight", "v_encoder.c_encoder.embeddings.position_embeddings.weight",
            "v_encoder.f_encoder.encoder.layer.0.attention.self.query.weight"],
    "vsm": ["video_query_linear.weight", "q_feat_attn.modular_vector_mapping.weight",
            "v_encoder.c_encoder.encoder.layer.1.attention.self.value.weight",
            "v_encoder.f_encoder.encoder.layer.5.intermediate.dense.weight"],
}


@pytest.mark.parametrize("task", ["mfm-nce", "fom", "vsm"])
def test_hero_base_bf16_pretraining_heads_at_bench_size_vs_oracle(task):
    """VERDICT r5 weak #1b: configs[3]'s task batches AS `bench.py --workload D3` BUILDS THEM (make_pretrain_batches: 32 videos x
    60 frames, 15 % of the frames masked / shuffled, 5 queries per video), HERO-base, bf16, against the oracle's mfm_loss /
    fom_loss / vsm_losses (pinned to the reference by tests/golden/case_pretrain.npz, incl. the several-queries-per-video
    branch).  MLM has its own full-vocabulary test below.  Losses by value; gradients of the task's mean loss (for VSM: of
    the smooth start / end term - the hinge terms' gradients flip with bf16 noise, see `objective`) as relative L2."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.model import HeroForPretraining
    from hero_amd.synth import make_pretrain_batches
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    P, model = hero_base(cls=HeroForPretraining)
    batch = make_pretrain_batches("D2", vocab=2048, seed=5)[task]
    cfg = O.cfg_from_json(HERO_BASE)
    names = D3_GRADS[task]
    Pq = {k: v.clone().requires_grad_(k in names) for k, v in P.items()}
    b = to_dev(batch, "cuda")
    report = {}
    if task == "mfm-nce":
        n_masked = int(batch["c_v_masks"].sum())
        assert n_masked > 200
        ref = O.mfm_loss(batch, Pq, cfg, loss="nce")
        b["c_v_feats"] = b["c_v_feats"].clone()                  # forward_mfm mutates it (model/model.py:244-247)
        got = model(b, task="mfm-nce", compute_loss=True)
        assert got.shape == ref.shape == (n_masked,)
        report["loss.l2"], report["loss.mean"] = l2_err(got, ref), abs(float(got.detach().mean()) - float(ref.detach().mean())) / float(ref.detach().mean())
        ref.mean().backward()
        got.mean().backward()
    elif task == "fom":
        assert int((batch["targets"] >= 0).sum()) > 200
        ref = O.fom_loss(batch, Pq, cfg)
        got = model(b, task="fom", compute_loss=True)
        report["loss.mean"] = abs(float(got) - float(ref)) / float(ref)
        with torch.no_grad():
            lg, lr = model(b, task="fom", compute_loss=False), O.fom_logits(batch, P, cfg)
        keep = (batch["targets"].reshape(-1) >= 0)
        report["logits.l2"] = l2_err(lg.reshape(lr.shape)[keep], lr[keep])
        ref.backward()
        got.backward()
    else:
        assert batch["query_input_ids"].shape[0] == 5 * batch["c_v_feats"].shape[0]
        ref = O.vsm_losses(batch, Pq, cfg)
        got = model(b, task="vsm", compute_loss=True)
        for k, g_, w_ in zip(("st_ed", "neg_ctx", "neg_q"), got, ref):
            report["loss." + k] = abs(float(g_.detach().sum()) - float(w_.detach())) / (abs(float(w_.detach())) + 1e-4)
        ref[0].backward()
        got[0].sum().backward()
    params = dict(model.named_parameters())
    for n in names:
        assert params[n].grad is not None and Pq[n].grad is not None, n
        report["grad." + n] = l2_err(params[n].grad, Pq[n].grad)
    print(task, json.dumps(report, indent=1))
    HF.clear_weight_cache()
    # measured on MI355X (round 6): loss values 5e-5 ... 2.6e-3 (FOM logits 1.1e-2 L2); gradients, relative L2: MFM-NCE 3.3 - 4.1 %
    # (its logits are inner products with the RAW 4352-wide feature targets: |logit| ~ 60, one bf16 ulp of a logit is 0.25),
    # FOM / VSM 0.9 - 1.9 % and 3.9 % for frame_transform.net.1.weight (as on the D2 batch)
    assert all(v < 5e-3 for k, v in report.items() if k.startswith("loss.")), report
    assert all(v < 2e-2 for k, v in report.items() if k.startswith("logits.")), report
    gate = 0.055 if task == "mfm-nce" else 0.03
    assert all(v < (0.05 if "frame_transform" in k else gate) for k, v in report.items() if k.startswith("grad.")), report


def test_bf16_error_growth_per_layer():
    """Where the bf16 error of the benched path comes from: the cross-modal stack truncated after 0..6 layers (HIP bf16
    vs the fp32 oracle on the same D2 slice).  The growth must stay roughly linear in depth (post-LN renormalises every
    layer, so errors add, they do not compound); the gates above are set from these numbers."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.synth import make_batch
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    P, model = hero_base()
    batch = make_batch("D2", vocab=2048, seed=7, videos=4)
    b = to_dev(batch, "cuda")
    cfg = O.cfg_from_json(HERO_BASE)
    enc = model.v_encoder.f_encoder.encoder
    layers = list(enc.layer)
    growth = []
    try:
        for n in range(0, 7):
            enc.layer = torch.nn.ModuleList(layers[:n])
            with torch.no_grad():
                got = model.v_encoder.f_encoder(b, "repr")[0]
                want = O.f_encoder_repr(batch, P, cfg._replace(f_layers=n))
            growth.append((n, l2_err(got, want, batch["f_attn_masks"]), rel_err(got, want, batch["f_attn_masks"])))
    finally:
        enc.layer = torch.nn.ModuleList(layers)
    print("bf16 error after n cross-modal layers (n, L2, max-norm):", [(n, round(a, 5), round(m, 5)) for n, a, m in growth])
    l2 = [a for _, a, _ in growth]
    assert l2[0] < 4e-3 and l2[6] < BF16_L2_TOL
    assert all(l2[n + 1] - l2[n] < 3e-3 for n in range(6)), growth             # additive, not compounding
    assert max(m for _, _, m in growth) < BF16_MAX_TOL


def long_video_batch(ragged, videos=2, vocab=2048, seed=11):
    """configs[4] shapes: 256 frames, 64 subtitles x 4 frames, 20 tokens per subtitle, 15-token queries.
    ragged: 256 / 201 frames, subtitles of 0-6 frames and 4-30 tokens, some frames matched to no subtitle."""
    from hero_amd import synth
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))   # noqa: E731
    subs, n_frames, qlens = [], [], []
    for v in range(videos):
        if not ragged:
            nf = 256
            cur = [(list(range(s * 4, s * 4 + 4)), 20) for s in range(64)]
            ql = 15
        else:
            nf = 256 if v == 0 else 201
            cur, f0 = [], 0
            for _ in range(64):
                k = ri(0, 6)
                fr = list(range(f0, min(f0 + k, nf - 3)))          # the last frames stay unmatched
                f0 += len(fr)
                cur.append((fr, ri(4, 30)))
            ql = ri(5, 25)
        subs.append(cur)
        n_frames.append(nf)
        qlens.append(ql)
    batch = synth.video_batch(subs, n_frames, 4352, vocab, gen)
    batch.update(synth.query_batch(videos, qlens, vocab, gen))
    st = torch.tensor([ri(0, nf - 2) for nf in n_frames])
    batch["targets"] = torch.stack([st, torch.minimum(st + 2, torch.tensor(n_frames) - 1)], dim=1)
    batch["q_vidx"] = torch.arange(videos)
    return batch


@pytest.mark.parametrize("ragged", [False, True])
def test_long_video_256_frame_temporal_transformer(ragged):
    import hero_amd
    from hero_amd import functional as HF
    HF.set_grad_sink(None)
    P, model = hero_base(seed=1)
    batch = long_video_batch(ragged)
    assert batch["c_v_feats"].shape[1] == 256
    names = [n for n in GRAD_NAMES if "c_encoder" in n or "frame_transform" in n or "layer.5" in n]
    ref_losses, ref_grads, ref_frames, R = oracle_losses_and_grads(batch, P, names)
    b = to_dev(batch, "cuda")
    # fp32 compute (exact-f32 MFMA GEMMs, fp32 attention at L = 256): forward parity with the oracle
    hero_amd.set_compute_dtype(torch.float32)
    HF.clear_weight_cache()
    with torch.no_grad():
        f32 = model.v_encoder(b, "repr")
    assert rel_err(f32, ref_frames, batch["c_attn_masks"]) < FP32_TOL
    # fp32 GRADIENTS at the config-5 length (round 3: the fp32 attention backward reaches L = 256 by reading Q / dO from
    # global memory): the smooth objective of `oracle_losses_and_grads` through the 256-frame Temporal Transformer
    model.zero_grad()
    l32 = model(b, task="tvr", compute_loss=True)
    objective(l32[0], model.v_encoder(b, "repr"), R).backward()
    p32 = dict(model.named_parameters())
    g32 = {n: l2_err(p32[n].grad, ref_grads[n]) for n in names}
    print("fp32 gradient parity at L = 256:", json.dumps(g32))
    assert all(v < 2e-3 for v in g32.values()), g32
    model.zero_grad()
    # bf16 compute: matrix-core attention at L = 256 (forward + backward) against fp32 / the oracle
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.clear_weight_cache()
    report = hip_report(model, b, batch, f32, ref_losses, ref_grads, R, names)
    HF.clear_weight_cache()
    assert report["repr.max"] < BF16_MAX_TOL and report["repr.l2"] < BF16_L2_TOL, report
    assert all(v < 3e-2 for k, v in report.items() if k.startswith("loss.")), report
    assert all(v < 0.06 for k, v in report.items() if k.startswith("grad.")), report


def _train(use_graph, n_micro, sync_at):
    """The bench's own configuration (HERO-base, bf16, D2 batch, dropout 0.1, TVR options) for n_micro micro-steps."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.step import TrainStep
    from hero_amd.synth import make_batch
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    _, model = hero_base(seed=2)
    set_dropout(model, 0.1)
    HF.manual_seed(5, "cuda")
    b = make_batch("D2", vocab=2048, seed=9, device="cuda")
    ts = TrainStep(model, use_graph=use_graph, static_usage=True)
    losses = []
    for i in range(n_micro):
        losses.append(ts.micro_step(b).clone())
        if i in sync_at:
            torch.cuda.synchronize()                 # the queue runs dry: the pattern that exposed the memset-node race
    torch.cuda.synchronize()
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    return torch.stack(losses).float().cpu()


def test_training_step_is_bit_reproducible_run_to_run():
    """Round 4: no kernel of the 1-GPU step accumulates with fp32 atomics any more (the FFN1 bias gradient rides on the
    batched weight-gradient launch, embedding-table gradients are segmented sums over sorted rows, the sliced tail of
    the batched wgrad applies its atomics in slice order) - two runs from the same seeds give the SAME losses, bit for
    bit, through optimiser steps, eagerly and in hipGraph replay."""
    a, b = _train(False, 24, set()), _train(False, 24, set())
    print("eager run-to-run max |diff|:", float((a - b).abs().max()))
    assert torch.equal(a, b)
    c, d = _train(True, 24, set()), _train(True, 24, set())
    print("graph run-to-run max |diff|:", float((c - d).abs().max()), "graph vs eager[4:]:", float((c[:20] - a[4:]).abs().max()))
    assert torch.equal(c, d)
    # ... and the replayed step IS the eager step (same kernels, same dropout sites and seeds, same optimiser arithmetic):
    # graph mode spends its 4 warm-up micro-steps eagerly, then follows the eager run's losses
    torch.testing.assert_close(c[:20], a[4:], rtol=2e-2, atol=2e-3)


def test_graph_replay_long_run_with_midrun_sync_converges_like_eager():
    """Regression of 7a0be53 (a hipMemsetAsync NODE raced with the kernel accumulating into its buffer whenever
    the queue had been idle): 5 of 13 replayed runs with a synchronise in them collapsed to the margin
    solution (loss ~ 1.69, once NaN) after ~200 micro-steps, 0 of 29 eager runs did.  Both modes over-fit the one
    synthetic batch within ~200 micro-steps when they are healthy."""
    n, sync_at = 320, {40, 150, 151, 260}
    l_e = _train(False, n + 4, sync_at)[4:]           # graph mode spends 4 eager warm-up micro-steps first
    l_g = _train(True, n, {s - 4 for s in sync_at})
    assert torch.isfinite(l_g).all() and torch.isfinite(l_e).all()
    rel = ((l_g - l_e).abs() / l_e.abs().clamp(min=5e-2))
    print("graph vs eager, relative difference of the losses: first 16 %.3g, first 40 %.3g, first 100 %.3g" %
          (float(rel[:16].max()), float(rel[:40].max()), float(rel[:100].max())))
    # Round 4: the step has no order-dependent sums left and the dropout sites restart per step, so the replayed run follows
    # the eager one (measured: identical losses over all 320 micro-steps; rounds 1-3 needed 5 % / 25 % over the first 16 / 40
    # steps because fp32 atomics reordered sums and the two modes drew different masks)
    torch.testing.assert_close(l_g, l_e, rtol=2e-2, atol=2e-3)
    tail_e, tail_g = float(l_e[-20:].mean()), float(l_g[-20:].mean())
    assert float(l_e[0]) > 1.0
    assert tail_e < 0.3, tail_e                        # the eager run has over-fitted its one batch ...
    assert tail_g < 0.3, tail_g                        # ... and the replayed run did too (not stuck at the margin terms)


def test_lse_hard_negative_losses_and_gradients_vs_reference_fixture():
    """ranking_loss_type = "lse" with hard-negative weighting on the REFERENCE's numbers (tests/golden/case_collate.npz,
    the narrow batch): fp32 HIP path, losses and four gradients."""
    import os
    import numpy as np
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.utils.misc import set_dropout
    from tests.test_cpu_collate import Z, ref_batch
    from tests.util import load_tiny
    hero_amd.set_compute_dtype(torch.float32)
    HF.set_grad_sink(None)
    try:
        model, _, _ = load_tiny("cuda", ranking_loss_type="lse", use_hard_negative=True, hard_pool_size=1, hard_neg_weight=10)
        model.train()
        set_dropout(model, 0.0)
        b = to_dev({k: v for k, v in ref_batch("narrow").items() if k != "vids"}, "cuda")
        losses = model(b, task="tvr", compute_loss=True)
        for got, key in zip(losses, ("loss_st_ed", "loss_neg_ctx", "loss_neg_q")):
            np.testing.assert_allclose(got.detach().cpu().numpy(), Z["narrow.lse." + key], rtol=2e-4, atol=1e-6)
        sum(losses).mean().backward()
        params = dict(model.named_parameters())
        for k in Z.files:
            if k.startswith("narrow.lse.grad."):
                n = k[len("narrow.lse.grad."):]
                assert rel_err(params[n].grad, torch.from_numpy(Z[k])) < 1e-3, n
    finally:
        hero_amd.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("mode", ["f32-grads", "bf16-grads", "bf16-hard-values"])
def test_hero_base_lse_ranking_loss(mode):
    """configs[1]'s loss options beyond the benched ones: 'lse' is the smooth ranking loss, hard negatives (pool 20 x
    weight 10, config/train-tvr-8gpu.json:54-62) start at step 2000.  HERO-base on D2 batches vs the oracle.
      f32-grads        fp32 compute, 8 videos: losses and the gradient of the WHOLE loss, ranking terms included, to
                       fp32 accuracy - the backward of the ranking head at full model size;
      bf16-grads       bf16, the full batch, same objective: the gradient bound is wider than for the start/end +
                       probe objective (0.06): every query-video score is a MAX over frames
                       (model/pretrain.py:364-413), the gradient flows to the arg-max frame only, and which of two
                       near-tied frames wins flips with bf16 noise (measured 7-10 % relative L2 at 3e-5 loss error);
      bf16-hard-values bf16 + hard negatives: loss VALUES only (pool membership comes from a sort on top of that)."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.synth import make_batch
    f32, hard = mode == "f32-grads", mode == "bf16-hard-values"
    hero_amd.set_compute_dtype(torch.float32 if f32 else torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    try:
        P, model = hero_base(ranking_loss_type="lse", use_hard_negative=hard, hard_pool_size=20, hard_neg_weight=10)
        batch = make_batch("D2", vocab=2048, seed=9, videos=8 if f32 else None)
        cfg = O.cfg_from_json(HERO_BASE)
        Pq = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("pad") and not hard) for k, v in P.items()}
        with torch.set_grad_enabled(not hard):
            ref = O.vsm_losses(batch, Pq, cfg, hard=(20, 10.0) if hard else None, ranking="lse")
        b = to_dev(batch, "cuda")
        losses = model(b, task="tvr", compute_loss=True)
        report = {}
        for k, got, want in zip(("st_ed", "neg_ctx", "neg_q"), losses, ref):
            report["loss." + k] = abs(float(got.detach()) - float(want.detach())) / (abs(float(want.detach())) + 1e-4)
        if not hard:
            sum(ref).backward()
            sum(losses).mean().backward()
            params = dict(model.named_parameters())
            for n in GRAD_NAMES:
                report["grad." + n] = l2_err(params[n].grad, Pq[n].grad)
        print(json.dumps(report, indent=1))
        assert all(v < (2e-4 if f32 else 2e-2) for k, v in report.items() if k.startswith("loss.")), report
        assert all(v < (5e-3 if f32 else 0.15) for k, v in report.items() if k.startswith("grad.")), report
    finally:
        hero_amd.set_compute_dtype(torch.bfloat16)
        HF.clear_weight_cache()


def test_hero_base_bf16_full_vocabulary_mlm_loss_and_tied_embedding_gradient():
    """configs[3] at full fidelity of its one large non-encoder contraction: HERO-base, vocabulary 50265 padded to
    50272 by pad_vocab(), the (masked rows x 768) x (768 x 50272) tied-weight GEMM + cross-entropy with the padding
    columns excluded (model/layers.py:330-354, model/encoder.py:224-233, 355-374), bf16, on an 8-video slice of the
    pre-training batch: per-token losses and the gradient of the TIED word-embedding / decoder matrix vs the oracle."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.model import HeroForPretraining
    from hero_amd.synth import make_pretrain_batches
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    cfg_d = json.loads(json.dumps(HERO_BASE))
    cfg_d["f_config"]["vocab_size"] = cfg_d["q_config"]["vocab_size"] = 50265
    path = "/tmp/hero_base_full_vocab_cfg.json"
    with open(path, "w") as f:
        json.dump(cfg_d, f)
    torch.manual_seed(0)
    model = HeroForPretraining.from_pretrained(path, {}, vfeat_dim=4352, max_frm_seq_len=100, lw_neg_ctx=8.0, lw_neg_q=8.0,
                                               lw_st_ed=0.01, ranking_loss_type="hinge", use_hard_negative=False,
                                               hard_pool_size=20, margin=0.1, use_all_neg=True, drop_svmr_prob=0.0)
    model.v_encoder.f_encoder.pad_vocab()
    assert model.v_encoder.f_encoder.embeddings.word_embeddings.weight.shape[0] == 50272
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    set_dropout(model, 0.0)
    mlm = make_pretrain_batches("D2", vocab=50265, seed=5, videos=8)["mlm"]
    n_masked = int(mlm["txt_mask_tgt"].sum())
    assert n_masked > 200
    name = "v_encoder.f_encoder.embeddings.word_embeddings.weight"
    Pq = {k: v.clone().requires_grad_(k == name or k.endswith("lm_head.dense.weight")) for k, v in P.items()}
    Pq["v_encoder.f_encoder.lm_head.decoder.weight"] = Pq[name]              # tied (model/layers.py:342-345)
    ref = O.mlm_loss(mlm, Pq, O.cfg_from_json(cfg_d), vocab_pad=7)
    ref.mean().backward()
    loss = model(to_dev(mlm, "cuda"), task="mlm", compute_loss=True)
    assert loss.shape == (n_masked,)
    loss.mean().backward()
    params = dict(model.named_parameters())
    report = {"loss.l2": l2_err(loss, ref), "loss.mean": abs(float(loss.mean()) - float(ref.mean())) / float(ref.mean()),
              "grad.word_embeddings": l2_err(params[name].grad, Pq[name].grad),
              "grad.lm_head.dense": l2_err(params["v_encoder.f_encoder.lm_head.dense.weight"].grad,
                                           Pq["v_encoder.f_encoder.lm_head.dense.weight"].grad)}
    print(json.dumps(report, indent=1))
    assert float(params[name].grad[50265:].abs().max()) == 0.0            # padding rows of the vocabulary get no gradient
    assert report["loss.l2"] < 2e-2 and report["loss.mean"] < 5e-3, report
    assert report["grad.word_embeddings"] < 0.06 and report["grad.lm_head.dense"] < 0.06, report


def test_qkv_bias_gradients_with_and_without_the_ride_on_the_batched_wgrad():
    """Round 6: the QKV bias gradients ride on hero_wgrad_batch (since late round 6 as one selector MFMA per row block in the
    compute waves - the default at every size, profiles/r06_mfma_ride_ab.txt) or are deferred column sums of their own
    (functional.WGRAD_RIDE_MAX_ROWS = 0).  Both paths on the same HERO-base bf16 backward pass (8 videos of the D2 batch):
    the bias gradients agree to fp32 summation order, everything else is bit-identical."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.synth import make_batch
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    _, model = hero_base()
    b = to_dev(make_batch("D2", vocab=2048, seed=7, videos=8), "cuda")
    keep = HF.WGRAD_RIDE_MAX_ROWS[0]
    grads = []
    try:
        for cap in (1 << 30, 0):
            HF.WGRAD_RIDE_MAX_ROWS[0] = cap
            model.zero_grad()
            losses = model(b, task="tvr", compute_loss=True)
            sum(l.sum() for l in losses).backward()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    finally:
        HF.WGRAD_RIDE_MAX_ROWS[0] = keep
        HF.clear_weight_cache()
    ride, own = grads
    assert set(ride) == set(own)
    qkv_bias = [k for k in ride if k.endswith((".self.query.bias", ".self.key.bias", ".self.value.bias"))]
    assert len(qkv_bias) == 3 * (6 + 3 + 1)                       # cross-modal, temporal and the query encoder's attention blocks
    moved = 0
    for k in ride:
        if k.endswith(".bias") and not torch.equal(own[k], ride[k]):
            moved += 1                                            # any nn.Linear bias whose sum rode on the launch (QKV, the projections)
            # two fp32 summation orders of the same bf16 dY columns; the KEY bias gradient is a sum that cancels to ~1e-7 of the
            # others (softmax is invariant to a constant added to every score of a row), so its relative agreement is looser
            assert l2_err(own[k], ride[k]) < (1e-3 if k.endswith("key.bias") else 1e-5), k
        else:
            assert torch.equal(own[k], ride[k]), k                # every weight gradient, LayerNorm parameter, embedding table: same bits
    assert moved >= len(qkv_bias) - 10 and all(float(ride[k].abs().max()) > 0 for k in qkv_bias)
