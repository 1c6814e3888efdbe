"""Collate on the device: the index tensors hero_amd.collate.DeviceCollate derives from length
arrays equal, bit for bit, the ones the REFERENCE's collate produced (tests/golden/case_collate.npz, written by the
reference's own video_collate / get_gather_index, incl. batches narrower than max_vl + max_sl) and the frame map of
hero_amd.model.model.build_frame_map; the HIP model on the reference-shaped narrow batch gives the reference model's
outputs; the model gives identical outputs from host- and device-collated batches; a captured hipGraph replays on a
NEW batch written into the static buffers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lengths(batch):
    return batch["lengths"]


def _cases():
    import json
    import os
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "case_collate.npz"))
    return json.loads(str(z["__cases__"]))


@pytest.mark.parametrize("case", _cases())
def test_device_collate_equals_reference_collate(case):
    """Reference batch (fixture) vs DeviceCollate fed with the ~1 KB of lengths + c_v_feats only."""
    from hero_amd.collate import DeviceCollate
    from hero_amd.model.model import build_frame_map
    from tests.test_cpu_collate import rebuild
    mine, want = rebuild(case)                       # host half == fixture is the CPU test; lengths come from it
    D = want["c_v_feats"].shape[2]
    dc = DeviceCollate.for_batch(mine, "cuda", vfeat_dim=D if D % 4 == 0 else None)
    assert dc.Lf == want["f_attn_masks"].shape[1]
    dc.update(mine["lengths"], c_v_feats=want["c_v_feats"].cuda() if D % 4 == 0 else None)
    torch.cuda.synchronize()
    assert torch.equal(dc.f_gather_index.cpu(), want["f_gather_index"])
    assert torch.equal(dc.f_attn_masks.cpu(), want["f_attn_masks"])
    assert torch.equal(dc.c_attn_masks.cpu(), want["c_attn_masks"])
    if D % 4 == 0:
        assert torch.equal(dc.f_v_feats.cpu(), want["f_v_feats"])
    B, NF = want["c_attn_masks"].shape
    if all(0 <= f < NF for rows in want["sub_idx2frame_idx"] for _, fr in rows for f in fr):
        offs, ent, inv = build_frame_map(want["num_subs"], want["sub_idx2frame_idx"], B, NF, dc.Lf, "cpu")
        assert torch.equal(dc.offsets.cpu(), offs)
        nnz = int(offs[-1])
        assert torch.equal(dc.entries.cpu()[:nnz], ent[:nnz])
        assert torch.equal(dc.inverse.cpu(), inv)


@pytest.mark.parametrize("device_collated", [False, True])
def test_hip_model_on_the_reference_narrow_batch(device_collated):
    """The reference model's outputs on a batch its own collate produced with f_attn_masks NARROWER than
    max_vl + max_sl (and a zero-frame subtitle, uncovered frames, a clipped video): fp32 HIP path within 2e-4."""
    import os
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.collate import DeviceCollate
    from hero_amd.utils.misc import set_dropout
    from tests.test_cpu_collate import Z, rebuild
    from tests.util import load_tiny, rel_err, to_dev
    hero_amd.set_compute_dtype(torch.float32)
    HF.set_grad_sink(None)
    try:
        model, _, _ = load_tiny("cuda")
        model.eval()
        mine, want = rebuild("narrow")
        b = to_dev({k: v for k, v in want.items() if k != "vids"}, "cuda")
        if device_collated:
            dc = DeviceCollate.for_batch(mine, "cuda", vfeat_dim=want["c_v_feats"].shape[2])
            dc.update(mine["lengths"], c_v_feats=b["c_v_feats"])
            b.update(dc.batch_entries())
            b.pop("num_subs"), b.pop("sub_idx2frame_idx")
        W = want["f_attn_masks"].shape[1]
        assert W < want["f_v_feats"].shape[1] + want["f_sub_input_ids"].shape[1]
        with torch.no_grad():
            f_seq = model.v_encoder.f_encoder(b, "repr")[0]
            rep = model.v_encoder(b, "repr")
            txt = model.v_encoder.f_encoder({"input_ids": b["query_input_ids"], "pos_ids": b["query_pos_ids"],
                                             "attn_masks": b["query_attn_masks"]}, "txt")[0]
        assert f_seq.shape == (want["f_attn_masks"].shape[0], W, 128)
        t = lambda k: torch.from_numpy(Z["narrow.model." + k])      # noqa: E731
        assert rel_err(f_seq, t("f_seq"), want["f_attn_masks"]) < 2e-4
        assert rel_err(rep, t("repr"), want["c_attn_masks"]) < 2e-4
        assert rel_err(txt, t("txt"), want["query_attn_masks"]) < 2e-4
        model.train()
        set_dropout(model, 0.0)
        with torch.no_grad():
            losses = model(b, task="tvr", compute_loss=True)
        for got, key in zip(losses, ("loss_st_ed", "loss_neg_ctx", "loss_neg_q")):
            np.testing.assert_allclose(got.cpu().numpy(), Z["narrow.model." + key], rtol=2e-4, atol=1e-6)
    finally:
        hero_amd.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("ragged", [False, True])
def test_device_collate_equals_host_collate(ragged):
    from hero_amd.collate import DeviceCollate
    from hero_amd.model.model import build_frame_map
    from hero_amd.synth import make_batch
    b = make_batch("D2", vfeat_dim=64, vocab=512, seed=3, ragged=ragged, videos=6)
    B, NF = b["c_attn_masks"].shape
    dc = DeviceCollate.for_batch(b, "cuda", vfeat_dim=64).update(_lengths(b), c_v_feats=b["c_v_feats"].cuda())
    torch.cuda.synchronize()
    assert torch.equal(dc.f_gather_index.cpu(), b["f_gather_index"])
    assert torch.equal(dc.f_attn_masks.cpu(), b["f_attn_masks"])
    assert torch.equal(dc.c_attn_masks.cpu(), b["c_attn_masks"])
    assert torch.equal(dc.f_v_feats.cpu(), b["f_v_feats"])
    offs, ent, inv = build_frame_map(b["num_subs"], b["sub_idx2frame_idx"], B, NF, dc.Lf, "cpu")
    assert torch.equal(dc.offsets.cpu(), offs)
    nnz = int(offs[-1])
    assert torch.equal(dc.entries.cpu()[:nnz], ent[:nnz])
    assert torch.equal(dc.inverse.cpu(), inv)


def test_model_output_identical_with_device_collated_batch():
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.collate import DeviceCollate
    from tests.util import load_tiny, to_dev
    from hero_amd.synth import make_batch
    hero_amd.set_compute_dtype(torch.float32)
    HF.set_grad_sink(None)
    model, _, _ = load_tiny("cuda")
    model.eval()
    b = make_batch("D1", vfeat_dim=96, vocab=160, seed=4, ragged=True, videos=3)
    while b["c_v_feats"].shape[1] > 66 or b["f_sub_input_ids"].shape[1] > 66:      # the tiny model has 66 positions
        b = make_batch("D1", vfeat_dim=96, vocab=160, seed=int(b["c_v_feats"].shape[1]) + 1000, ragged=True, videos=3)
    dc = DeviceCollate.for_batch(b, "cuda").update(_lengths(b))
    d = to_dev(b, "cuda")
    d2 = dict(d)
    d2.update(dc.batch_entries())
    d2.pop("num_subs"), d2.pop("sub_idx2frame_idx")                 # the host lists are not needed any more
    with torch.no_grad():
        torch.testing.assert_close(model.v_encoder(d2, "repr"), model.v_encoder(d, "repr"), rtol=0, atol=0)
    hero_amd.set_compute_dtype(torch.bfloat16)


def test_graph_replays_a_new_batch_written_into_the_static_buffers():
    """graph mode froze every host-derived index / mask tensor at capture.  With the batch's index
    tensors owned by DeviceCollate and refresh_memo(), a new batch of the same shape replays correctly."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.collate import DeviceCollate
    from hero_amd.step import TrainStep
    from hero_amd.synth import make_batch
    from hero_amd.utils.misc import set_dropout
    from tests.util import load_tiny, to_dev
    hero_amd.set_compute_dtype(torch.float32)

    def ragged_same_shape(seed):
        # same padded shapes (8 subtitles x <= 4 frames, <= 8 tokens, 32-frame videos), different lengths / matches
        from hero_amd import synth
        gen = torch.Generator().manual_seed(seed)
        ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))     # noqa: E731
        subs, n_frames = [], []
        for v in range(2):
            nf, cur, f0 = (32 if v == 0 else ri(20, 31)), [], 0
            for s_ in range(8):
                k = 4 if s_ == 0 else ri(0, 4)
                fr = list(range(f0, min(f0 + k, nf)))
                f0 += len(fr)
                cur.append((fr, 8 if s_ == 0 else ri(2, 8)))
            subs.append(cur)
            n_frames.append(nf)
        b = synth.video_batch(subs, n_frames, 96, 160, gen, max_frames=32)
        b.update(synth.query_batch(2, [12, ri(4, 11)], 160, gen))
        b["targets"] = torch.tensor([[1, 3], [2, 5]])
        b["q_vidx"] = torch.arange(2)
        return b

    def fresh():
        HF.set_grad_sink(None)
        HF.clear_weight_cache()
        model, _, _ = load_tiny("cuda")
        model.train()
        set_dropout(model, 0.0)
        return model

    b1, b2 = ragged_same_shape(1), ragged_same_shape(2)
    assert all(b1[k].shape == b2[k].shape for k in b1 if torch.is_tensor(b1[k]))
    assert not torch.equal(b1["f_attn_masks"], b2["f_attn_masks"])
    from hero_amd.model.layers import BertEncoder
    BertEncoder.allow_packing = False               # packing changes the row counts with the batch: eager-only feature
    try:
        # eager reference: two optimiser steps on b1 (4 warm-up micro-steps happen inside graph capture too), then b2
        model = fresh()
        ts = TrainStep(model, opts=dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100))
        d1, d2 = to_dev(b1, "cuda"), to_dev(b2, "cuda")
        for _ in range(6):
            ts.micro_step(d1)
        want = [float(ts.micro_step(d2)) for _ in range(2)]
        HF.set_grad_sink(None)

        model = fresh()
        dc = DeviceCollate.for_batch(b1, "cuda").update(_lengths(b1))
        static = to_dev(b1, "cuda")
        static.update(dc.batch_entries())
        static.pop("num_subs"), static.pop("sub_idx2frame_idx")
        ts = TrainStep(model, opts=dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100), use_graph=True)
        for _ in range(2):                           # capture (4 eager warm-up micro-steps inside) + 2 replays on b1
            ts.micro_step(static)
        # new batch, same buffers: payload copied in, index tensors rebuilt on the device, derived tensors refreshed
        for k in ("f_sub_input_ids", "f_v_feats", "c_v_feats", "query_input_ids", "query_attn_masks", "targets"):
            static[k].copy_(b2[k].to("cuda"))
        dc.update(_lengths(b2))
        HF.refresh_memo()
        got = [float(ts.micro_step(static)) for _ in range(2)]
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-5)
    finally:
        BertEncoder.allow_packing = True
        HF.set_grad_sink(None)
        hero_amd.set_compute_dtype(torch.bfloat16)
