"""Training-step harness on the GPU: hipGraph replay must reproduce eager execution, and the step
semantics (accumulation, clipping, untouched parameters) must match the reference loop."""
import os

import pytest
import torch

from oracle import hero_oracle as O
from tests.util import GOLDEN, load_tiny, rel_err, to_dev

pytestmark = pytest.mark.gpu


def _run(use_graph, n_micro):
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    HF.clear_weight_cache()
    model, _, _ = load_tiny("cuda")
    model.train()
    set_dropout(model, 0.0)
    batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
    b = to_dev(batch, "cuda")
    ts = TrainStep(model, opts=dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100), use_graph=use_graph)
    losses = []
    for _ in range(n_micro):
        losses.append(ts.micro_step(b).clone())
    torch.cuda.synchronize()
    HF.set_grad_sink(None)
    return model, torch.stack(losses).cpu()


def test_graph_replay_matches_eager():
    m_e, l_e = _run(False, 14)
    m_g, l_g = _run(True, 10)         # graph mode runs 4 eager warm-up micro-steps inside its first call
    l_e = l_e[4:]
    torch.testing.assert_close(l_g, l_e, rtol=2e-4, atol=1e-5)
    assert float(l_e[-1]) < float(l_e[0])                     # it trains
    pe, pg = dict(m_e.named_parameters()), dict(m_g.named_parameters())
    worst = max(rel_err(pg[k], pe[k]) for k in pe)
    assert worst < 5e-4, worst
    # parameters the step never touches keep their initial values (reference: p.grad is None -> skip)
    _, P, _ = load_tiny("cpu")
    for k in ("v_encoder.f_encoder.pooler.dense.weight", "v_encoder.fom_output.linear_1.weight"):
        torch.testing.assert_close(pe[k].detach().cpu(), P[k])
        torch.testing.assert_close(pg[k].detach().cpu(), P[k])


def test_first_optimizer_step_matches_reference_numbers():
    """Two micro-steps of accumulation on the same batch == gradient 2x the golden one; with clip
    1.0 the AdamW result equals the oracle's update for that gradient."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    model, P0, cfg = load_tiny("cuda")
    model.train()
    set_dropout(model, 0.0)
    batch, outs = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
    b = to_dev(batch, "cuda")
    ts = TrainStep(model, opts=dict(learning_rate=1e-3, warmup_steps=1, num_train_steps=100))
    l1 = ts.micro_step(b)
    name = "v_encoder.f_encoder.encoder.layer.1.output.dense.weight"
    g1 = dict(model.named_parameters())[name].grad.clone()
    assert rel_err(g1, outs["grad." + name]) < 1e-3
    ts.micro_step(b)                                           # boundary: optimiser ran, grads zeroed
    assert float(dict(model.named_parameters())[name].grad.abs().sum()) == 0.0
    # oracle: same accumulated gradient (2x), clipped, one AdamW step at lr(step 1) = 1e-3 * 1/1 -> 0 floor...
    Pq = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pad")) for k, v in P0.items()}
    sum(O.vsm_losses(batch, Pq, cfg)).backward()
    G = {k: 2 * p.grad for k, p in Pq.items() if p.requires_grad and p.grad is not None}
    gn = torch.sqrt(sum(g.double().pow(2).sum() for g in G.values())).float()
    G = {k: g * min(1.0, 1.0 / (float(gn) + 1e-6)) for k, g in G.items()}
    from hero_amd.optim import get_lr_sched
    lr = get_lr_sched(1, ts.opts)
    with torch.no_grad():
        O.adamw_step({k: p for k, p in Pq.items() if p.requires_grad}, G, {}, lr=lr, step=1)
    got = dict(model.named_parameters())
    for k in (name, "v_encoder.f_encoder.embeddings.word_embeddings.weight", "video_query_linear.weight",
              "v_encoder.c_encoder.encoder.layer.0.attention.self.key.bias"):
        assert rel_err(got[k], Pq[k]) < 2e-4, k
    HF.set_grad_sink(None)


def test_one_launch_weight_refresh_equals_per_tensor_casts():
    """HF.refresh_weight_cache (hero_copy_multi) rewrites every cached bf16 / transposed / packed copy
    exactly like the per-tensor hero_cast / hero_transpose_cast path that first filled the cache."""
    import hero_amd
    from hero_amd import functional as HF
    hero_amd.set_compute_dtype(torch.bfloat16)
    HF.clear_weight_cache()
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(n, 96, device="cuda")) for n in (64, 130, 40)]
    bs = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (64, 130, 40)]
    try:
        W = HF.packed(ws, torch.bfloat16)
        Wt = HF.packed_t(ws, torch.bfloat16)
        B = HF.packed(bs, torch.float32)
        with torch.no_grad():
            for p in ws + bs:
                p.mul_(1.5).add_(0.25)
        HF.notify_weights_updated()
        HF.refresh_weight_cache()
        ref = torch.cat([p.detach() for p in ws], 0)
        assert HF.packed(ws, torch.bfloat16) is W and HF.packed_t(ws, torch.bfloat16) is Wt   # cache hits, same buffers
        torch.testing.assert_close(W, ref.to(torch.bfloat16), rtol=0, atol=0)
        torch.testing.assert_close(Wt, ref.t().contiguous().to(torch.bfloat16), rtol=0, atol=0)
        torch.testing.assert_close(HF.packed(bs, torch.float32), torch.cat([p.detach() for p in bs]), rtol=0, atol=0)
    finally:
        HF.clear_weight_cache()


def test_checkpoint_restore_after_steps_matches_uninterrupted_run():
    """optimiser.load_state_dict replaces the moment tensors; the AdamW descriptor tables must not
    keep pointing at the old ones.  save -> 2 more optimiser steps -> restore -> the same 2 steps == the first time."""
    import copy
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    hero_amd.set_compute_dtype(torch.float32)
    HF.clear_weight_cache()
    try:
        model, _, _ = load_tiny("cuda")
        model.train()
        set_dropout(model, 0.0)
        batch, _ = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
        b = to_dev(batch, "cuda")
        ts = TrainStep(model, opts=dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100))
        for _ in range(4):
            ts.micro_step(b)
        saved_model = copy.deepcopy(model.state_dict())
        saved_train = copy.deepcopy(ts.state_dict())
        first = [float(ts.micro_step(b)) for _ in range(4)]
        after_first = {k: v.detach().clone() for k, v in model.named_parameters()}
        model.load_state_dict(saved_model)
        ts.load_state_dict(saved_train)
        second = [float(ts.micro_step(b)) for _ in range(4)]
        import numpy as np
        np.testing.assert_allclose(second, first, rtol=1e-4)          # split-K atomics reorder fp32 additions
        for k, v in model.named_parameters():
            assert rel_err(v, after_first[k]) < 1e-4, k
        # the restored moments are the ones being updated (not the freed pre-restore buffers)
        p = dict(model.named_parameters())["video_query_linear.weight"]
        st = ts.optimizer.state[p]
        assert float(st["exp_avg"].abs().sum()) > 0 and st["step"] == 4
    finally:
        HF.set_grad_sink(None)
        HF.clear_weight_cache()
        hero_amd.set_compute_dtype(torch.bfloat16)


def test_multi_task_graph_replay_keeps_per_parameter_adam_steps():
    """in graph mode the Adam bias-correction step of a parameter that only ONE task updates must advance
    only in that task's optimiser steps (the reference's state['step'], optim/adamw.py:71-72).  Per-parameter counters
    live on the device and are advanced by the captured AdamW launch for its active parameters only: a two-task run
    (vsm / mlm windows interleaved) replayed from per-task graphs ends with the parameters of the eager run."""
    import hero_amd
    from hero_amd import functional as HF
    from hero_amd.model import HeroForPretraining
    from hero_amd.model.layers import BertEncoder
    from hero_amd.step import TrainStep
    from hero_amd.utils.misc import set_dropout
    from tests.test_oracle_golden import _task_batches
    hero_amd.set_compute_dtype(torch.float32)
    pre, _ = O.load_npz_case(os.path.join(GOLDEN, "case_pretrain.npz"))
    mlm, _, _ = _task_batches(pre)
    vsm, _ = O.load_npz_case(os.path.join(GOLDEN, "case_train.npz"))
    batches = {"mlm": to_dev(mlm, "cuda"), "vsm": to_dev(vsm, "cuda")}
    windows = ["vsm", "mlm", "mlm", "vsm", "mlm"]

    def run(use_graph):
        HF.set_grad_sink(None)
        HF.clear_weight_cache()
        model, _, _ = load_tiny("cuda", cls=HeroForPretraining)
        model.train()
        set_dropout(model, 0.0)
        ts = TrainStep(model, opts=dict(learning_rate=1e-3, warmup_steps=2, num_train_steps=100), task="vsm", use_graph=use_graph)
        if use_graph:                                      # capture both tasks first (each capture runs 2 warm-up windows)
            for t in ("vsm", "mlm"):
                ts.prepare(batches[t], t)
        else:
            for t in ("vsm", "mlm"):
                for _ in range(4):
                    ts.micro_step(batches[t], t)
        for t in windows:
            for _ in range(2):
                ts.micro_step(batches[t], t)
        torch.cuda.synchronize()
        sd = ts.state_dict()
        steps = {n: ts.optimizer.state[p]["step"] for n, p in model.named_parameters() if p in ts.optimizer.state and "step" in ts.optimizer.state[p]}
        HF.set_grad_sink(None)
        return {k: v.detach().clone() for k, v in model.named_parameters()}, steps, sd

    BertEncoder.allow_packing = False
    try:
        pe, se, _ = run(False)
        pg, sg, _ = run(True)
    finally:
        BertEncoder.allow_packing = True
        HF.clear_weight_cache()
        hero_amd.set_compute_dtype(torch.bfloat16)
    only_mlm = "v_encoder.f_encoder.lm_head.dense.weight"
    only_vsm = "video_query_linear.weight"
    both = "v_encoder.f_encoder.encoder.layer.0.attention.self.query.weight"
    assert se[only_mlm] == 2 + 3 and se[only_vsm] == 2 + 2 and se[both] == 4 + 5       # warm-up windows + the run's
    assert sg == se, {k: (sg[k], se[k]) for k in se if sg.get(k) != se[k]}
    for k in (only_mlm, only_vsm, both):
        assert rel_err(pg[k], pe[k]) < 5e-4, (k, rel_err(pg[k], pe[k]))
