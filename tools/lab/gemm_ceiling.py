#!/usr/bin/env python3
"""Per K,K launch of the training step: the product kernel (with the epilogue the step uses) against
  (a) the same kernel's MAIN LOOPS ALONE, outputs discarded (a -DHERO_WS_NOEPI build selected with HERO_HIP_LIB),
  (b) the vendor library's plain GEMM behind torch.nn.functional.linear (no epilogue at all),
and against what the box can do at best: max(flops / measured MFMA peak, algorithmic bytes / measured HBM copy rate)
(hero_probe_mfma / hero_probe_hbm).  One process per library build: `python gemm_ceiling.py product|noepi out.json`;
`python gemm_ceiling.py table product.json noepi.json` prints the table (tools/lab/gemm_ceiling.sh runs all three).
hipGraph timing of 20 back-to-back launches past the clock ramp; operands of one launch fit the 256 MB Infinity Cache."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

# (name, M, N, K, epilogue kind, algorithmic MB beyond A + W + C, launches per micro-step)
SHAPES = [
    ("QKV fwd", 12000, 2304, 768, "bias", 0, 6),
    ("attn-out fwd", 12000, 768, 768, "bias+drop+res", 1, 6),
    ("FFN1 fwd", 12000, 3072, 768, "bias+gelu'", 1, 6),
    ("FFN2 fwd", 12000, 768, 3072, "bias+drop+res", 1, 6),
    ("FFN2 dgrad", 12000, 3072, 768, "mul_aux", 1, 6),
    ("FFN1 dgrad", 12000, 768, 3072, "res", 1, 6),
    ("attn-out dgrad", 12000, 768, 768, "none", 0, 6),
    ("QKV dgrad", 12000, 768, 2304, "res", 1, 6),
]


def timer(torch):
    def t(fn, reps=20):
        end = time.time() + 0.25
        while time.time() < end:
            fn()
        torch.cuda.synchronize()
        gs = torch.cuda.Stream()
        gs.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(gs):
            with torch.cuda.graph(g, stream=gs):
                for _ in range(reps):
                    fn()
        torch.cuda.current_stream().wait_stream(gs)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / reps / 3)
        return best
    return t


def measure(kind, out):
    import torch
    import torch.nn.functional as F
    from hero_amd import functional as HF, _lib as L
    t = timer(torch)
    dt = torch.bfloat16
    res = {"kind": kind, "lib": L.LIB_PATH, "rows": {}}
    if kind == "product":
        n = 1 << 30
        a = torch.zeros(n // 4, device="cuda")
        b = torch.empty_like(a)
        tf, ghz, cp, rd = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        st = torch.cuda.current_stream().cuda_stream
        L.check(L.lib().hero_probe_mfma(b.data_ptr(), n, C.byref(tf), C.byref(ghz), st))
        L.check(L.lib().hero_probe_hbm(a.data_ptr(), b.data_ptr(), n, C.byref(cp), C.byref(rd), st))
        res["box"] = {"mfma_tflops": tf.value, "ghz": ghz.value, "hbm_copy_gbps": cp.value, "hbm_read_gbps": rd.value}
        del a, b
    drop = HF.RNG.make(0.1, True, torch.device("cuda"))
    for name, M, N, K, epi, extra, per_step in SHAPES:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        b = torch.randn(N, device="cuda")
        r2 = torch.randn(M, N, device="cuda").to(dt)
        aux = torch.empty(M, N, device="cuda", dtype=dt)
        fn = {"none": lambda: HF.k_linear(x, w),
              "bias": lambda: HF.k_linear(x, w, b),
              "bias+drop+res": lambda: HF.k_linear(x, w, b, residual=r2, drop=drop),
              "bias+gelu'": lambda: HF.k_linear(x, w, b, act=L.ACT_GELU_DG, aux=aux),
              "mul_aux": lambda: HF.k_dgrad_t(x, w, act=L.ACT_MUL_AUX, aux=r2),
              "res": lambda: HF.k_linear(x, w, residual=r2)}[epi]
        row = {"us": t(fn)}
        if kind == "product":
            row["us_plain"] = t(lambda: HF.k_linear(x, w))
            row["us_library"] = t(lambda: F.linear(x, w))
        res["rows"][name] = row
        print(kind, name, row, flush=True)
    json.dump(res, open(out, "w"), indent=1)


def table(prod, noepi):
    p, q = json.load(open(prod)), json.load(open(noepi))
    box = p["box"]
    print("box: %.0f TFLOP/s dense bf16 MFMA measured (hero_probe_mfma, %.2f GHz), HBM copy %.0f GB/s, read %.0f GB/s (hero_probe_hbm)"
          % (box["mfma_tflops"], box["ghz"], box["hbm_copy_gbps"], box["hbm_read_gbps"]))
    print("attainable = max(flops / measured MFMA peak, algorithmic bytes / measured copy rate); us per launch, hipGraph timing\n")
    print("%-15s %5s %5s %5s %-14s | %8s %8s %8s %8s | %6s %6s | %7s %7s %7s | %6s" % (
        "launch", "M", "N", "K", "epilogue", "product", "loop", "plain", "library", "mfma", "hbm", "prod/at", "loop/at", "lib/at", "TF/s"))
    tot = {"product": 0.0, "loop": 0.0, "att": 0.0, "lib": 0.0}
    for name, M, N, K, epi, extra, per_step in SHAPES:
        a, b = p["rows"][name], q["rows"][name]
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K + M * N) + 2.0 * extra * M * N
        t_m = fl / (box["mfma_tflops"] * 1e12) * 1e6
        t_h = by / (box["hbm_copy_gbps"] * 1e9) * 1e6
        att = max(t_m, t_h)
        print("%-15s %5d %5d %5d %-14s | %8.1f %8.1f %8.1f %8.1f | %6.1f %6.1f | %7.2f %7.2f %7.2f | %6.0f" % (
            name, M, N, K, epi, a["us"], b["us"], a["us_plain"], a["us_library"], t_m, t_h,
            a["us"] / att, b["us"] / att, a["us_library"] / att, fl / a["us"] / 1e6))
        tot["product"] += per_step * a["us"]; tot["loop"] += per_step * b["us"]; tot["att"] += per_step * att; tot["lib"] += per_step * a["us_library"]
    print("\nper micro-step (48 launches): product %.3f ms, main loops alone %.3f ms, attainable %.3f ms, library plain GEMMs %.3f ms"
          % (tot["product"] / 1e3, tot["loop"] / 1e3, tot["att"] / 1e3, tot["lib"] / 1e3))
    print("=> the product family runs at %.2f of attainable; with a FREE epilogue (loop-only) it would run at %.2f; the vendor library's "
          "plain GEMM (no epilogue) at %.2f" % (tot["att"] / tot["product"], tot["att"] / tot["loop"], tot["att"] / tot["lib"]))


if __name__ == "__main__":
    if sys.argv[1] == "table":
        table(sys.argv[2], sys.argv[3])
    else:
        measure(sys.argv[1], sys.argv[2])
