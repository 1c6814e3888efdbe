#!/usr/bin/env python3
"""Do the epilogue write bursts of a multi-round K,K GEMM (every workgroup stores its tile at the same moment, HBM idle during
the main loops) overlap with MFMA work if the GEMM runs as TWO half-chip launches on two streams (HERO_WS_LAB_CUS=128), whose
workgroups are out of phase?  Prints us per (full) GEMM for: one launch on 256 CUs / two N-halves on two streams with 128
workgroups each / two M-halves likewise."""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker():
    import torch
    from hero_amd import functional as HF, _lib as L
    dt = torch.bfloat16
    half = os.environ.get("HERO_WS_LAB_CUS") == "128"
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def t(fn, reps=20):
        # captured: the fork / join of the two streams costs tens of us of host time per iteration in eager mode
        for _ in range(3): fn()
        torch.cuda.synchronize()
        gs = torch.cuda.Stream(); gs.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(gs):
            with torch.cuda.graph(graph, stream=gs):
                for _ in range(reps): fn()
        torch.cuda.current_stream().wait_stream(gs)
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): graph.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / reps / 3

    for M, N, K in [(12000, 3072, 768), (12000, 2304, 768), (12000, 768, 768), (12000, 768, 3072)]:
        x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        b = torch.randn(N, device="cuda"); aux = torch.empty(M, N, device="cuda", dtype=dt); y = torch.empty(M, N, device="cuda", dtype=dt)
        res = torch.randn(M, N, device="cuda").to(dt)
        kinds = {"bias": dict(bias=b), "bias+gelu_dg": dict(bias=b, act=L.ACT_GELU_DG, aux=aux), "res": dict(residual=res)}
        for name, kw in kinds.items():
            def one():
                HF.k_gemm(x, w, y, M, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.BF16, **kw)

            def two(split_n):
                cur = torch.cuda.current_stream()
                s1.wait_stream(cur); s2.wait_stream(cur)
                for i, st in enumerate((s1, s2)):
                    with torch.cuda.stream(st):
                        if split_n:
                            h = N // 2
                            kw2 = {k: (v if k == "act" else (L.ptr(v) + (4 if k == "bias" else 2) * h * i)) for k, v in kw.items()}
                            HF.k_gemm(x, L.ptr(w) + 2 * h * K * i, L.ptr(y) + 2 * h * i, M, h, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.BF16, **kw2)
                        else:
                            h = (M // 2 + 191) // 192 * 192
                            m0, m = (0, h) if i == 0 else (h, M - h)
                            kw2 = {k: (v if k in ("act", "bias") else (L.ptr(v) + 2 * m0 * N)) for k, v in kw.items()}
                            HF.k_gemm(L.ptr(x) + 2 * m0 * K, w, L.ptr(y) + 2 * m0 * N, m, N, K, K, K, N, L.LAYOUT_K, L.LAYOUT_K, L.BF16, **kw2)
                cur.wait_stream(s1); cur.wait_stream(s2)
            if not half:
                print("%6d %5d %5d %-13s one launch %7.1f us" % (M, N, K, name, t(one)), flush=True)
            else:
                print("%6d %5d %5d %-13s two N-halves %7.1f us   two M-halves %7.1f us   (one launch on 128 CUs %7.1f)" % (
                    M, N, K, name, t(lambda: two(True)), t(lambda: two(False)), t(one)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker()
    else:
        for cus in ("", "128"):
            env = dict(os.environ)
            if cus: env["HERO_WS_LAB_CUS"] = cus
            subprocess.run([sys.executable, os.path.abspath(__file__), "w"], env=env, check=False)
