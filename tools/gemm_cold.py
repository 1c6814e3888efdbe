#!/usr/bin/env python3
"""Hot-vs-cold GEMM timing: the same forward GEMM on ONE buffer set (operands and output stay in
L2 / Infinity Cache between calls) versus a ROTATING ring of buffer sets larger than the 256 MB
Infinity Cache (every call streams its operands from HBM and its output to HBM, as inside the
training step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L


def run(M, N, K, ring, act, aux_on, n=48):
    dt = torch.bfloat16
    xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(ring)]
    ys = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(ring)]
    us = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(ring)] if aux_on else [None] * ring
    w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda")

    def call(i):
        j = i % ring
        HF.k_gemm(xs[j], w, ys[j], M, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.dt(xs[j]), bias=b,
                  act=act, aux=us[j])
    for i in range(ring):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for (M, N, K, act, aux) in [(12000, 3072, 768, L.ACT_GELU, True), (12000, 3072, 768, L.ACT_NONE, False),
                                (12000, 768, 3072, L.ACT_NONE, False), (12000, 2304, 768, L.ACT_NONE, False),
                                (12000, 768, 768, L.ACT_NONE, False), (1920, 768, 768, L.ACT_NONE, False),
                                (1920, 3072, 768, L.ACT_NONE, False), (1920, 768, 3072, L.ACT_NONE, False)]:
        fl = 2.0 * M * N * K
        byts = 2.0 * (M * K + N * K + M * N * (2 if aux else 1))
        ring = max(2, int(600e6 // byts) + 1)
        hot = run(M, N, K, 1, act, aux)
        cold = run(M, N, K, ring, act, aux)
        print("M=%5d N=%4d K=%4d gelu+aux=%d  hot %6.1f us %6.1f TF/s | cold(ring %2d) %6.1f us %6.1f TF/s  %5.2f TB/s"
              % (M, N, K, int(aux), hot, fl / hot / 1e6, ring, cold, fl / cold / 1e6, byts / cold / 1e6), flush=True)


if __name__ == "__main__":
    main()
