#!/usr/bin/env python3
"""GEMM micro-benchmark on the shapes of the HERO step (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hero_amd import functional as HF, _lib as L

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

def main():
    dt = torch.bfloat16
    shapes = [(12000, 3072, 768), (12000, 768, 3072), (12000, 768, 768), (12000, 2304, 768),
              (12000, 768, 2304), (1920, 768, 768), (1920, 3072, 768), (1920, 768, 4352)]
    only = sys.argv[1] if len(sys.argv) > 1 else "all"
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    L.lib().hero_gemm_force_config(cfg)
    print("config", cfg)
    for M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        dy = torch.randn(M, N, device="cuda").to(dt); b = torch.randn(N, device="cuda")
        fl = 2.0 * M * N * K
        r = {}
        # numerics under the forced geometry (fp32 reference of the same bf16 inputs)
        ref = x.float() @ w.float().t() + b
        e1 = (HF.k_linear(x, w, b).float() - ref).abs().max().item() / ref.abs().max().item()
        e2 = (HF.k_dgrad(dy, w).float() - dy.float() @ w.float()).abs().max().item()
        wr = dy.float().t() @ x.float()
        e3 = (HF.k_wgrad(dy, x) - wr).abs().max().item() / wr.abs().max().item()
        if max(e1, e3) > 2e-2 or e2 > 0.5:
            print("  !! numerics off: fwd %.3g dgrad %.3g wgrad %.3g" % (e1, e2, e3))
        if only in ("all", "fwd"):
            r["fwd"] = timeit(lambda: HF.k_linear(x, w, b))
        if only in ("all", "dgrad"):
            r["dgrad"] = timeit(lambda: HF.k_dgrad(dy, w))
        if only in ("all", "wgrad"):
            out = torch.zeros(N, K, device="cuda")
            r["wgrad"] = timeit(lambda: HF.k_wgrad(dy, x, out=out, beta=1.0))
        print("M=%5d N=%4d K=%4d  " % (M, N, K) + "  ".join("%s %7.1f us %6.1f TF/s" % (k, v, fl / v / 1e6) for k, v in r.items()), flush=True)

if __name__ == "__main__":
    main()
