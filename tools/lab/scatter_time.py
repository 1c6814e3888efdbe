#!/usr/bin/env python3
"""hero_scatter_add_sorted on the word-embedding gradients of the bench batches: 9600 sub-tokens with the SEP id 480 times (D2),
and with 15 % of them replaced by the <mask> id (MLM); us per call (hipGraph timing, kernel + fold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import functional as HF


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        with torch.cuda.graph(g, stream=gs):
            for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(gs)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps / 3


gen = torch.Generator(device="cuda").manual_seed(1)
for name, mask_frac in (("D2 sub-tokens", 0.0), ("MLM sub-tokens", 0.15), ("queries (480)", -1)):
    rows = 480 if mask_frac < 0 else 9600
    ids = torch.randint(3, 50272, (rows,), device="cuda", generator=gen, dtype=torch.int32)
    if mask_frac >= 0:
        ids[::20] = 2
        if mask_frac:
            ids[torch.rand(rows, device="cuda", generator=gen) < mask_frac] = 50264
    src = torch.randn(rows, 768, device="cuda", generator=gen).bfloat16()
    dst = torch.zeros(50272, 768, device="cuda")
    HF.k_scatter_add_sorted(src, ids, dst, skip=1)
    ref = torch.zeros_like(dst).index_add_(0, ids.long(), src.float())
    err = (dst - ref).abs().max().item()
    print("%-16s %6.1f us   max |err| vs index_add_ %.2e" % (name, t(lambda: HF.k_scatter_add_sorted(src, ids, dst, skip=1)), err), flush=True)
