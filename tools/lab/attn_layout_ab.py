#!/usr/bin/env python3
"""VERDICT r5 #2, first lever, priced before it is built: would the short-sequence attention kernels stream faster from per-head
PANELS ([3][H][M][64] / [H][M][64]) than from the fused projection's rows ([M, 3 x 768]: 128-byte pieces at a 4608-byte stride)?
The same kernels, compiled twice (product / -DHERO_ATTN_LAB_HEADMAJOR: only the address arithmetic differs; random operands, so
the numbers are the same work either way), on the bench batch's launch (480 sequences x 24 rows x 12 heads, mask, dropout, row
statistics), HOT (one buffer set, resident in the 256 MB Infinity Cache) and COLD (a ring of 8 buffer sets = 1.1 GB walked by
one hipGraph: every launch finds its operands in HBM, as inside the training step).
Run once per build:  python tools/lab/attn_layout_ab.py ; HERO_HIP_LIB=tools/lab/libhero_headmajor.so python tools/lab/attn_layout_ab.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from hero_amd import functional as HF, _lib as L
dt = torch.bfloat16
S, Lq, H = 480, 24, 12
D = H * 64
NSET = 8
g = torch.Generator(device="cuda").manual_seed(1)
sets = []
for _ in range(NSET):
    qkv = torch.randn(S * Lq, 3 * D, device="cuda", generator=g).to(dt)
    dctx = torch.randn(S * Lq, D, device="cuda", generator=g).to(dt)
    sets.append(dict(qkv=qkv, dctx=dctx, ctx=torch.empty(S * Lq, D, device="cuda", dtype=dt), dq=torch.empty_like(qkv)))
madd = torch.zeros(S, Lq, device="cuda")
madd[:, -2:] = -10000.0
drop = HF.RNG.make(0.1, True, torch.device("cuda", 0))
for s_ in sets:
    _, s_["saved"] = HF.k_attn_fwd(s_["qkv"], madd, S, Lq, H, drop=drop, out=s_["ctx"])


def t(fns, rounds=3):
    """us per launch of a hipGraph that runs every fn of `fns` once, replayed."""
    end = time.time() + 0.25
    while time.time() < end:
        for f in fns:
            f()
    torch.cuda.synchronize()
    gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        with torch.cuda.graph(gr, stream=gs):
            for _ in range(rounds):
                for f in fns:
                    f()
    torch.cuda.current_stream().wait_stream(gs)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / (rounds * len(fns)) / 5)
    return best


def fwd(s_):
    return lambda: HF.k_attn_fwd(s_["qkv"], madd, S, Lq, H, drop=drop, out=s_["ctx"])


def bwd(s_):
    return lambda: HF.k_attn_bwd(s_["qkv"], s_["saved"], s_["dctx"], S, Lq, H, drop=drop, out=s_["dq"], ctx=s_["ctx"], mask_add=madd)


tag = os.environ.get("HERO_HIP_LIB", "product").split("/")[-1]
mb_f, mb_b = 71.0, 142.0                                  # 3 M D e + M D e forward; + dctx, dqkv backward
for ppw in ([0] if "ppw" not in sys.argv else [1, 2, 3, 0]):      # `ppw`: also sweep the pairs-per-wave hook, HOT and COLD
    L.check(L.lib().hero_attention_force_ppw(ppw))
    fh, bh = t([fwd(sets[0])] * NSET), t([bwd(sets[0])] * NSET)
    fc, bc = t([fwd(s_) for s_ in sets]), t([bwd(s_) for s_ in sets])
    print("%-22s ppw %-4s HOT  fwd %5.1f us  bwd %5.1f us  (%.2f / %.2f TB/s)   COLD fwd %5.1f us  bwd %5.1f us  (%.2f / %.2f TB/s)" %
          (tag, ppw or "auto", fh, bh, mb_f / fh, mb_b / bh, fc, bc, mb_f / fc, mb_b / bc), flush=True)
