import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from hero_amd import functional as HF, _lib as L
from gemm_bench import timeit
for (M, N, K, cfg) in [(12000, 3072, 768, 0), (12000, 2304, 768, 0), (12000, 768, 3072, 1), (12000, 768, 768, 1)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = []
    for grp in (1, 2, 4, 8, 16, 32, 94):
        L.lib().hero_gemm_force_config(cfg | (grp << 8))
        t = timeit(lambda: HF.k_linear(x, w, b, residual=res), n=30)
        out.append("%d:%.1f" % (grp, t))
    L.lib().hero_gemm_force_config(-1)
    print("M=%5d N=%4d K=%4d cfg%d  " % (M, N, K, cfg) + "  ".join(out), flush=True)
