#!/usr/bin/env python3
"""Short-sequence attention launches of the bench batch (512 sequences x 24 rows x 12 heads incl. the query group as one
packed launch; mask, dropout 0.1, row statistics) under 1 / 2 / 3 pairs per wave (hero_attention_force_ppw), hipGraph timing.
Run once per library build (HERO_HIP_LIB): product, and the -DHERO_ATTN_LATE_ISSUE variant (next pair's loads issued after
this pair's have landed).  `old` as argv[1]: a tree without the hook (round 4) - one timing."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from hero_amd import functional as HF, _lib as L
dt = torch.bfloat16
S, Lq, H = 480, 24, 12
D = H * 64
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(S * Lq, 3 * D, device="cuda", generator=g).to(dt)
dctx = torch.randn(S * Lq, D, device="cuda", generator=g).to(dt)
madd = torch.zeros(S, Lq, device="cuda")
madd[:, -2:] = -10000.0
drop = HF.RNG.make(0.1, True, qkv.device)


def t(fn, reps=10):
    end = time.time() + 0.25
    while time.time() < end:
        fn()
    torch.cuda.synchronize()
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        with torch.cuda.graph(gr, stream=gs):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(gs)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / reps / 5)
    return best


ctx = torch.empty(S * Lq, D, device="cuda", dtype=dt)
dq = torch.empty_like(qkv)
_, saved = HF.k_attn_fwd(qkv, madd, S, Lq, H, drop=drop, out=ctx)
tag = os.environ.get("HERO_HIP_LIB", "product").split("/")[-1]
modes = [0] if (len(sys.argv) > 1 and sys.argv[1] == "old") else [1, 2, 3, 0]
for ppw in modes:
    if modes != [0]:
        L.check(L.lib().hero_attention_force_ppw(ppw))
    f = t(lambda: HF.k_attn_fwd(qkv, madd, S, Lq, H, drop=drop, out=ctx))
    b = t(lambda: HF.k_attn_bwd(qkv, saved, dctx, S, Lq, H, drop=drop, out=dq, ctx=ctx, mask_add=madd))
    print("%-28s ppw %s  fwd %6.1f us  bwd %6.1f us   (75 / 148 MB: %.2f / %.2f TB/s)" % (tag, ppw or "auto", f, b, 75e6 / f / 1e6, 148e6 / b / 1e6), flush=True)
