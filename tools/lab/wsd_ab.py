#!/usr/bin/env python3
"""In-line epilogue (hero_gemm_force_config 9 / 10) against the deferred one (11 / 12: gemm_wsd.hip) on the K,K shapes of
the step, per epilogue kind; also checks that the two give the same bits (same arithmetic, same rounding points)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import functional as HF, _lib as L
dt = torch.bfloat16


def t(fn, reps=30):
    end = time.time() + 0.2
    while time.time() < end: fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def main():
    shapes = [(12000, 2304, 768), (12000, 3072, 768), (12000, 768, 768), (12000, 768, 3072), (12000, 768, 2304),
              (14450, 2304, 768), (14450, 3072, 768), (24000, 768, 768), (24000, 768, 3072), (1920, 3072, 768)]
    drop = HF.RNG.make(0.1, True, torch.device("cuda"))
    print("%6s %5s %5s %-10s  inline us  defer us  ratio  TF/s(defer)  equal" % ("M", "N", "K", "epilogue"))
    for M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(dt)
        u = torch.randn(M, N, device="cuda").to(dt)
        aux = torch.empty(M, N, device="cuda", dtype=dt)
        kinds = {
            "none": lambda: HF.k_linear(x, w),
            "bias": lambda: HF.k_linear(x, w, b),
            "bias+d+res": lambda: HF.k_linear(x, w, b, residual=res, drop=drop),
            "bias+gelu": lambda: HF.k_linear(x, w, b, act=L.ACT_GELU, aux=aux),
            "res": lambda: HF.k_linear(x, w, residual=res),
            "gelu_bwd": lambda: HF.k_dgrad_t(x, w, act=L.ACT_GELU_BWD, aux=u),
        }
        small = M < 4000
        for name, fn in kinds.items():
            r, outs = {}, {}
            for tag, cfg in (("in", 10 if small else 9), ("de", 12 if small else 11)):
                L.lib().hero_gemm_force_config(cfg)
                outs[tag] = fn().clone()
                if name == "bias+gelu": outs[tag + "x"] = aux.clone()
                r[tag] = t(fn)
            L.lib().hero_gemm_force_config(-1)
            eq = torch.equal(outs["in"], outs["de"]) and (name != "bias+gelu" or torch.equal(outs["inx"], outs["dex"]))
            fl = 2.0 * M * N * K
            print("%6d %5d %5d %-10s  %8.1f  %8.1f  %5.2f  %8.0f  %s" % (M, N, K, name, r["in"], r["de"], r["de"] / r["in"], fl / r["de"] / 1e6, eq), flush=True)


if __name__ == "__main__":
    main()
