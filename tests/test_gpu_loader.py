"""Loader half of SURVEY 8(f) N4: the reference's PrefetchLoader protocol (data/loader.py:89-144) and the static-buffer
feeder that lets a hipGraph-captured step consume a stream of DIFFERENT host batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(n, seed0=10):
    """same padded shapes (8 subtitles x <= 4 frames, <= 8 tokens, <= 32 frames), different contents / lengths"""
    from hero_amd import synth
    out = []
    for s in range(n):
        gen = torch.Generator().manual_seed(seed0 + s)
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
        b["targets"] = torch.tensor([[1, 3], [2, ri(3, 9)]])
        b["q_vidx"] = torch.arange(2)
        out.append(b)
    return out


def test_prefetch_loader_follows_the_reference_protocol():
    from hero_amd.loader import PrefetchLoader
    host = _batches(3)
    got = list(PrefetchLoader(host, "cuda"))
    assert len(got) == 3
    for h, d in zip(host, got):
        for k, v in h.items():
            if torch.is_tensor(v):
                assert d[k].is_cuda and torch.equal(d[k].cpu(), v), k
            elif k != "lengths":
                assert d[k] == v                     # host lists pass through untouched
    assert len(PrefetchLoader(host, "cuda")) == 3


@pytest.mark.parametrize("use_graph", [False, True])
def test_static_feeder_feeds_a_stream_of_batches(use_graph):
    """A (captured) training step fed by StaticBatchFeeder with rotating, different batches gives the losses of an
    eager run that moves every batch to the device the plain way."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.loader import StaticBatchFeeder, pin_batch
    from hero_amd.model.layers import BertEncoder
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    from tests.util import load_tiny, to_dev
    hero_amd.set_compute_dtype(torch.float32)
    host = _batches(4)
    assert all(h["f_attn_masks"].shape == host[0]["f_attn_masks"].shape for h in host)
    order = [0, 1, 2, 3, 1, 0]

    def fresh():
        HF.set_grad_sink(None)
        HF.clear_weight_cache()
        model, _, _ = load_tiny("cuda")
        model.train()
        set_dropout(model, 0.0)
        return model

    BertEncoder.allow_packing = False
    try:
        opts = dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100)
        ts = TrainStep(fresh(), opts=opts)
        d0 = to_dev(host[0], "cuda")
        for _ in range(4):                           # what graph capture runs as warm-up on its first batch
            ts.micro_step(d0)
        want = [float(ts.micro_step(to_dev(host[i], "cuda"))) for i in order]
        HF.set_grad_sink(None)
        # with packing at its default (ON) a captured step would bake the capture batch's pack plan in and
        # replay it on the later, differently ragged batches.  The feeder marks its static batch and TrainStep runs the
        # padded formulation for it, eagerly and captured - nothing to remember for the caller.
        BertEncoder.allow_packing = True

        ts = TrainStep(fresh(), opts=opts, use_graph=use_graph)
        feeder = StaticBatchFeeder(pin_batch(host[0]), "cuda")
        if use_graph:
            ts.prepare(feeder.static)                # 4 eager warm-up micro-steps + capture on host[0]
        else:
            for _ in range(4):
                ts.micro_step(feeder.static)
        feeder.capture()
        pinned = [pin_batch(h) for h in host]
        feeder.prefetch(pinned[order[0]])
        got = []
        for n, i in enumerate(order):
            b = feeder.commit()
            if n + 1 < len(order):
                feeder.prefetch(pinned[order[n + 1]])
            got.append(float(ts.micro_step(b)))
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-5)
        assert torch.equal(feeder.static["f_attn_masks"].cpu(), host[order[-1]]["f_attn_masks"])
        assert torch.equal(feeder.static["f_v_feats"].cpu(), host[order[-1]]["f_v_feats"])
    finally:
        BertEncoder.allow_packing = True
        HF.set_grad_sink(None)
        hero_amd.set_compute_dtype(torch.bfloat16)


def _multiq_batches(n, seed0=70, per=3):
    """_batches with `per` queries per video (data/vsm.py:105-145); every other batch names its videos in another order
    (q_vidx reversed: the start / end term follows q_vidx, the ranking terms take m // per, as in the reference)."""
    from hero_amd import synth
    out = _batches(n, seed0)
    for k, b in enumerate(out):
        gen = torch.Generator().manual_seed(seed0 + 100 + k)
        nv = b["c_attn_masks"].shape[0]
        nq = per * nv
        b.update(synth.query_batch(nq, [12] + [int(torch.randint(4, 12, (1,), generator=gen)) for _ in range(nq - 1)], 160, gen))
        b["targets"] = torch.stack([torch.randint(0, 5, (nq,), generator=gen), torch.randint(5, 15, (nq,), generator=gen)], 1)
        b["q_vidx"] = torch.arange(nq) // per
        if k % 2:
            b["q_vidx"] = b["q_vidx"].flip(0).contiguous()
    return out


@pytest.mark.parametrize("use_graph", [False, True])
def test_static_feeder_with_several_queries_per_video(use_graph):
    """Round 6: batches with three queries per video through the fused HIP head under StaticBatchFeeder - the (query, video)
    pair index, its CSR for the backward and the pair mask are memoised tensors derived from q_vidx, which the feeder's
    commit refreshes IN PLACE (inside its commit graph) when the next batch names its videos in another order."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.loader import StaticBatchFeeder, pin_batch
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    from tests.util import load_tiny, to_dev
    hero_amd.set_compute_dtype(torch.float32)
    host = _multiq_batches(4)
    order = [0, 1, 2, 3, 1, 0]

    def fresh():
        HF.set_grad_sink(None)
        HF.clear_weight_cache()
        HF.reset_caches()
        model, _, _ = load_tiny("cuda")
        model.train()
        set_dropout(model, 0.0)
        return model

    try:
        opts = dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100)
        model = fresh()
        nq, nv = host[0]["query_input_ids"].shape[0], host[0]["c_attn_masks"].shape[0]
        assert model._head_is_fusable(torch.empty(nv, 1, 1, device="cuda"), torch.empty(nq, 1, device="cuda"), to_dev(host[0], "cuda"))
        ts = TrainStep(model, opts=opts)
        d0 = to_dev(host[0], "cuda")
        for _ in range(4):
            ts.micro_step(d0)
        want = [float(ts.micro_step(to_dev(host[i], "cuda"))) for i in order]
        assert len(set(round(w, 6) for w in want[:4])) == 4                  # the batches really differ
        ts = TrainStep(fresh(), opts=opts, use_graph=use_graph)
        feeder = StaticBatchFeeder(pin_batch(host[0]), "cuda")
        if use_graph:
            ts.prepare(feeder.static)
        else:
            for _ in range(4):
                ts.micro_step(feeder.static)
        feeder.capture()
        pinned = [pin_batch(h) for h in host]
        feeder.prefetch(pinned[order[0]])
        got = []
        for n, i in enumerate(order):
            b = feeder.commit()
            if n + 1 < len(order):
                feeder.prefetch(pinned[order[n + 1]])
            got.append(float(ts.micro_step(b)))
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-5)
        assert torch.equal(feeder.static["q_vidx"].cpu(), host[order[-1]]["q_vidx"])
    finally:
        HF.set_grad_sink(None)
        hero_amd.set_compute_dtype(torch.bfloat16)


def _ragged_batches(n, seed0=40):
    """2 videos each, every batch its own shape: 4-8 subtitles per video, 0-4 frames and 2-9 tokens per subtitle, 18-32 frames"""
    from hero_amd import synth
    out = []
    for s in range(n):
        gen = torch.Generator().manual_seed(seed0 + s)
        ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))     # noqa: E731
        subs, n_frames = [], []
        for v in range(2):
            nf, cur, f0 = ri(18, 32), [], 0
            for s_ in range(ri(4, 8)):
                fr = list(range(f0, min(f0 + ri(0, 4), nf)))
                f0 += len(fr)
                cur.append((fr, ri(2, 9)))
            subs.append(cur)
            n_frames.append(nf)
        b = synth.video_batch(subs, n_frames, 96, 160, gen)
        b.update(synth.query_batch(2, [ri(4, 12), ri(4, 12)], 160, gen))
        b["targets"] = torch.tensor([[1, 3], [2, ri(3, 9)]])
        b["q_vidx"] = torch.arange(2)
        out.append(b)
    return out


def test_pad_batch_keeps_the_losses_under_the_masks():
    """What bucket padding does to a batch (loader.pad_batch): the ranking losses are those of the original batch (padded
    positions are masked out of every score), the start / end loss may move a little - the reference's own Conv1d over the
    frame axis (model/pretrain.py:128-166) reads the two positions past a video's last frame, which are zero padding of the
    convolution in a batch where that video is the longest and (finite) outputs at padded frames in any batch where it is
    not.  i.e. the padded batch gives the reference's numbers for these videos collated together with longer ones."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.loader import BucketedBatchFeeder, batch_dims, pad_batch
    from hero_amd.utils.misc import set_dropout
    from oracle import hero_oracle as O
    from tests.util import load_tiny, to_dev
    hero_amd.set_compute_dtype(torch.float32)
    HF.set_grad_sink(None)
    try:
        host = _ragged_batches(3)
        bucket = BucketedBatchFeeder.derive_buckets([batch_dims(h) for h in host], n_buckets=1, row_quantum=16, slack=1.2)[0]
        model, P, cfg = load_tiny("cuda")
        model.train()
        set_dropout(model, 0.0)
        for h in host:
            p = pad_batch(h, bucket)
            assert batch_dims(p)["T"] == bucket["T"] > batch_dims(h)["T"] and batch_dims(p)["rows"] == batch_dims(h)["rows"]
            with torch.no_grad():
                a = [float(x.sum()) for x in model(to_dev({k: v for k, v in h.items() if k != "lengths"}, "cuda"), task="tvr")]
                b = [float(x.sum()) for x in model(to_dev({k: v for k, v in p.items() if k != "lengths"}, "cuda"), task="tvr")]
                want = [float(x) for x in O.vsm_losses(p, P, cfg)]          # the oracle (== reference) ON THE PADDED BATCH
            np.testing.assert_allclose(b, want, rtol=2e-4, atol=1e-6)
            np.testing.assert_allclose(a[1:], b[1:], rtol=1e-5, atol=1e-6)   # ranking losses: untouched by the padding
            assert abs(a[0] - b[0]) < 0.1 * abs(a[0])                        # start / end: the convolution's edge only
    finally:
        hero_amd.set_compute_dtype(torch.bfloat16)


def test_bucketed_feeder_streams_ragged_batches_through_a_few_graphs():
    """VERDICT r5 "missing" #2 / next #3: eight differently shaped ragged batches through <= 3 captured step graphs
    (BucketedBatchFeeder: bucket padding, one StaticBatchFeeder + one pair of step graphs per bucket, the cross-modal layers
    PACKED through a static pack plan that the feeder rebuilds on the host per batch) give the losses of an eager run over
    the same (bucket-padded) batches moved to the device the plain way, optimiser steps included."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.loader import BucketedBatchFeeder, batch_dims, pad_batch
    from hero_amd.model.layers import BertEncoder
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    from tests.util import load_tiny, to_dev
    hero_amd.set_compute_dtype(torch.float32)
    host = _ragged_batches(8)
    dims = [batch_dims(h) for h in host]
    assert len({(d["T"], d["Lf"], d["NF"], d["Lq"]) for d in dims}) >= 6            # really different shapes
    buckets = BucketedBatchFeeder.derive_buckets(dims, n_buckets=3, row_quantum=16)
    assert 2 <= len(buckets) <= 3
    order = [0, 1, 2, 3, 4, 5, 6, 7, 2, 0, 5, 7, 1, 3, 6, 4, 0, 1]

    def fresh():
        HF.set_grad_sink(None)
        HF.clear_weight_cache()
        model, _, _ = load_tiny("cuda")
        model.train()
        set_dropout(model, 0.0)
        return model

    try:
        opts = dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100)
        feeder = BucketedBatchFeeder(buckets, "cuda")
        padded = [feeder.pad(h)[1] for h in host]
        used = {p["_bucket"] for p in padded}
        assert len(used) >= 2 and -1 not in used
        plain = lambda p: to_dev({k: v for k, v in p.items() if k not in ("lengths", "_bucket")}, "cuda")      # noqa: E731
        ts = TrainStep(fresh(), opts=opts)
        d0 = plain(padded[order[0]])
        for _ in range(4):                           # what graph capture runs as warm-up on its first batch
            ts.micro_step(d0)
        want = [float(ts.micro_step(plain(padded[i]))) for i in order]
        HF.set_grad_sink(None)

        ts = TrainStep(fresh(), opts=opts, use_graph=True)
        feeder.prefetch(padded[order[0]])
        got = []
        for n, i in enumerate(order):
            b = feeder.commit()
            assert b is not None and b["_bucket"] == padded[i]["_bucket"] and b.get("_static_plan")
            got.append(float(ts.micro_step(b)))
            if n + 1 < len(order):
                feeder.prefetch(padded[order[n + 1]])
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-5)
        assert feeder.graphs == len(used) <= 3 and len(ts._graphs) == len(used)
        # every bucket's first batch ran eagerly (the very first one through the usual warm-up + capture), some batches of a
        # bucket waited for a window start - everything else was a replay
        assert ts.counts["replayed"] >= len(order) - 3 * len(used), ts.counts
        assert ts.counts["replayed"] + ts.counts["eager_in_graph_mode"] == len(order)
        # the static plan really is what ran: the registered plan of the last bucket holds the last batch's valid-row count
        last = padded[order[-1]]
        f = feeder.feeders[last["_bucket"]]
        lay = f.plan
        off = f.plan_flat[lay["off"][0]:lay["off"][0] + lay["off"][1]].cpu()
        assert int(off[lay["n_real"]]) == batch_dims(last)["rows"] and int(off[-1]) == lay["rows_cap"]
        assert torch.equal(f.static["f_attn_masks"].cpu(), last["f_attn_masks"])
        # a batch no bucket holds is handed back for an eager micro-step inside the same accumulation bookkeeping
        big = dict(host[0])
        big["query_input_ids"] = torch.nn.functional.pad(host[0]["query_input_ids"], (0, 40), value=1)
        big["query_attn_masks"] = torch.nn.functional.pad(host[0]["query_attn_masks"], (0, 40))
        big["query_pos_ids"] = torch.arange(big["query_input_ids"].shape[1]).unsqueeze(0)
        feeder.prefetch(big)
        assert feeder.commit() is None
        loss = ts.micro_step(feeder.take_eager(), eager=True)
        assert torch.isfinite(loss)
    finally:
        BertEncoder._STATIC_PLANS.clear()
        HF.set_grad_sink(None)
        hero_amd.set_compute_dtype(torch.bfloat16)


def test_static_pack_plan_at_bench_size_equals_the_dynamic_plan():
    """The static pack plan at the size and in the dtype `secondary.feed_ragged` runs it: HERO-base, bf16, the ragged D2 batch
    (32 videos, ~14 000 valid rows) padded to a bucket with a row capacity well above its row count (pad sequences in the
    attention launches, pad rows through every GEMM / LayerNorm / weight-gradient reduction).  Against the SAME padded batch
    through the dynamic plan (masks read on the host, exactly the valid rows): the frame representations and the three losses
    are bit-identical (every op between pack and unpack is row-wise or per sequence), gradients agree to fp32 summation order
    (the weight-gradient reduction walks more - all-zero - rows)."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.loader import BucketedBatchFeeder, StaticBatchFeeder, batch_dims, pad_to_bucket, pin_batch
    from hero_amd.model.layers import BertEncoder
    from hero_amd.synth import make_batch
    from tests.test_gpu_configs import GRAD_NAMES, hero_base, l2_err
    from tests.util import to_dev
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.set_grad_sink(None)
    HF.clear_weight_cache()
    try:
        _, model = hero_base()
        host = make_batch("D2", vocab=2048, seed=7, ragged=True)
        d = batch_dims(host)
        bucket = BucketedBatchFeeder.derive_buckets([d], n_buckets=1, slack=1.05)[0]
        bucket["rows"] = (d["rows"] // 512 + 3) * 512                       # > 1000 pad rows: ~35 pad sequences of 32 rows
        bucket["min_rows"] = d["rows"] - 100
        _, padded = pad_to_bucket(host, [bucket])
        assert batch_dims(padded)["T"] == bucket["T"] >= d["T"] and bucket["rows"] - d["rows"] > 1000
        feeder = StaticBatchFeeder(pin_batch(padded), "cuda", capture_commit=False, packed_rows=bucket["rows"], min_rows=bucket["min_rows"],
                                   frm_capacity=bucket["frm"])
        res = []
        for mode, b in (("static", feeder.static), ("all", to_dev({k: v for k, v in padded.items() if k not in ("lengths", "_bucket")}, "cuda"))):
            BertEncoder.packing_mode = mode
            model.zero_grad()
            with torch.no_grad():
                frames = model.v_encoder(b, "repr")
            losses = model(b, task="tvr", compute_loss=True)
            sum(l.sum() for l in losses).backward()
            params = dict(model.named_parameters())
            res.append((frames.clone(), [float(l.detach().sum()) for l in losses], {n: params[n].grad.clone() for n in GRAD_NAMES}))
        (f_s, l_s, g_s), (f_d, l_d, g_d) = res
        m = padded["c_attn_masks"].bool()
        assert torch.equal(f_s[m], f_d[m])
        assert l_s == l_d, (l_s, l_d)
        for n in GRAD_NAMES:
            assert l2_err(g_s[n], g_d[n]) < 1e-4, (n, l2_err(g_s[n], g_d[n]))
    finally:
        BertEncoder.packing_mode = "all"
        BertEncoder._STATIC_PLANS.clear()
        HF.set_grad_sink(None)
        HF.clear_weight_cache()


def test_feeder_rejects_other_shapes():
    from hero_amd.loader import StaticBatchFeeder, pin_batch
    from hero_amd.synth import make_batch
    a = make_batch("D1", vfeat_dim=96, vocab=160, seed=1)
    b = make_batch("D1", vfeat_dim=96, vocab=160, seed=1, videos=3)
    f = StaticBatchFeeder(pin_batch(a), "cuda", capture_commit=False)
    with pytest.raises(ValueError):
        f.prefetch(pin_batch(b))


@pytest.mark.gpu
def test_prefetch_loader_over_meta_loader_moves_the_whole_batch():
    """The reference's composition `PrefetchLoader(MetaLoader(...))` (pretrain.py:177-180, train_vcmr.py:83-86): items are
    `(task, batch_dict)` tuples, nested lists / tuples included - every tensor must arrive on the device."""
    import torch
    from hero_amd.loader import MetaLoader, PrefetchLoader
    batches = [{"x": torch.full((4,), float(i)), "pair": (torch.ones(2) * i, [torch.zeros(1), "keep"]), "n": i} for i in range(3)]
    ml = MetaLoader({"tvr": (batches, 1)}, accum_steps=1)
    seen = 0
    for task, batch in PrefetchLoader(ml, device="cuda"):
        assert task == "tvr"
        assert batch["x"].is_cuda and batch["pair"][0].is_cuda and batch["pair"][1][0].is_cuda
        assert batch["pair"][1][1] == "keep" and batch["n"] == seen % 3
        assert float(batch["x"][0]) == float(seen % 3)
        seen += 1
        if seen == 5:
            break
    assert seen == 5


def test_host_segment_order_is_the_device_sort():
    """StaticBatchFeeder sorts the token ids of the NEXT batch on the host (numpy stable argsort) instead of re-sorting on the
    device inside every commit: same order as hero_segment_sort (rows by (id, row), padding / negative ids last)."""
    from hero_amd import functional as HF
    g = torch.Generator().manual_seed(5)
    for n, vocab, skip in ((9600, 50272, 1), (480, 50272, 1), (777, 13, -1), (5, 3, 0)):
        ids = torch.randint(0, vocab, (n,), generator=g)
        ids[::7] = 2                                     # a run that spans many blocks (the SEP token)
        if skip >= 0:
            ids[3::11] = skip
        idx = ids.to(torch.int32).cuda()
        dev = HF.segment_order(idx, vocab, skip)
        host = torch.from_numpy(HF.host_segment_order(ids.numpy(), skip))
        assert torch.equal(dev.cpu(), host), (n, vocab, skip)
