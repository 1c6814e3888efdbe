#!/usr/bin/env python3
"""a ceiling for the batched weight-gradient launch (hero_wgrad_batch / gemm_wsb_kernel), the way
profiles/r05_gemm_ceiling.txt did it for the K,K family.  Per launch of the D2 step and of the D4 step (256 videos): the
product kernel against lab builds of the SAME kernel with pieces compiled out (tools/lab/build_variants.sh; results are
garbage, timing only)
    noepi     -DHERO_WSB_NOEPI                      main loops alone (no dW read-add-write)
    dmaonly   -DHERO_WSB_NOMFMA,-DHERO_WSB_NOLDF    the L2 -> LDS fill alone (loader waves + barriers)
    mfmaonly  -DHERO_WSB_NOLOADS,-DHERO_WSB_NOLDF   MFMA issue + the per-step barrier alone
against the vendor library's dY^T X (torch.matmul on transposed views: hipBLASLt / rocBLAS of this image, bf16 output, one
call per weight, summed), and against what the box can do at best: max(flops / measured MFMA peak, algorithmic bytes / HBM
rate) with the HBM rate taken twice - the box's own streaming-copy probe and the guide's 6.3 TB/s achievable.
One process per build: `wgrad_ceiling.py product|noepi|dmaonly|mfmaonly out.json`; `wgrad_ceiling.py table a.json ...`."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LAYER = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]          # (N = dY columns, K = X columns): FFN2, FFN1, attn-out, QKV
# (name, rows, problems, launches of this kind per micro-step)
LAUNCHES = [
    ("D2 cross-modal stack: 6 layers x 4 weights", 12000, LAYER * 6, 1),
    ("D2 temporal stack: 3 layers x 4 + frame_transform", 1920, LAYER * 3 + [(768, 4352)], 1),
    ("D2 img_linear (768 x 4352)", 1920, [(768, 4352)], 1),
    ("D4 (256 videos) cross-modal: ONE layer x 4 weights", 397056, LAYER, 6),
    ("D4 (256 videos) temporal stack: 3 layers x 4 + frame_transform", 65536, LAYER * 3 + [(768, 4352)], 1),
]


# `d4probe` (argv): how the D4 queue should group its flushes - per-tile cost of several groupings at 397056 rows
D4_PROBE = [
    ("D4 probe: 1 layer (192 tiles, balanced tail)", 397056, LAYER, 1),
    ("D4 probe: 1 layer + FFN2 of the next (256 tiles = one whole round)", 397056, LAYER + [(768, 3072)], 1),
    ("D4 probe: 2 layers (384 tiles)", 397056, LAYER * 2, 1),
    ("D4 probe: 4 layers (768 tiles = three whole rounds)", 397056, LAYER * 4, 1),
]


def timer(torch):
    def t(fn, reps):
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
        g.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / reps / 2)
        return best
    return t


def measure(kind, out):
    import numpy as np
    import torch
    from hero_amd import _lib as L
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
    for name, rows, shapes, per_step in (D4_PROBE if kind == "d4probe" else LAUNCHES):
        dys = [torch.randn(rows, n_, device="cuda", dtype=dt) for n_, _ in shapes]
        xs = [torch.randn(rows, k_, device="cuda", dtype=dt) for _, k_ in shapes]
        outs = [torch.zeros(n_, k_, device="cuda") for n_, k_ in shapes]
        n = len(shapes)
        pr = (L.WgradProblem * n)(*[L.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), outs[i].shape[0], outs[i].shape[1],
                                                   outs[i].shape[0], outs[i].shape[1], outs[i].shape[1], 4) for i in range(n)])
        buf = np.zeros(8 + 8 * 256 * 64, dtype=np.int32)
        words = L.lib().hero_wgrad_batch_plan(pr, n, rows, buf.ctypes.data, buf.size)
        assert words > 0, L.lib().hero_last_error()
        plan = torch.from_numpy(buf[:words].copy()).cuda()
        fn = lambda: L.check(L.lib().hero_wgrad_batch(pr, n, rows, L.BF16, plan.data_ptr(), words, L.stream()))     # noqa: E731
        reps = 6 if rows <= 20000 else 2
        row = {"us": t(fn, reps), "tiles": int(buf[6]), "rounds": int(buf[2]), "slices": int(buf[7])}
        row["us_per_tile_round"] = row["us"] / (row["tiles"] / 256.0)
        row["tflops"] = sum(2.0 * rows * a_ * b_ for a_, b_ in shapes) / row["us"] / 1e6
        if kind == "product":
            def lib():
                for dy, x in zip(dys, xs):
                    torch.matmul(dy.t(), x)
            row["us_library"] = t(lib, reps)
        res["rows"][name] = row
        print(kind, name, row, flush=True)
        del dys, xs, outs
        torch.cuda.empty_cache()
    json.dump(res, open(out, "w"), indent=1)


def table(paths):
    data = {}
    for p in paths:
        d = json.load(open(p))
        data[d["kind"]] = d
    p = data["product"]
    box = p["box"]
    print("box: %.0f TFLOP/s dense bf16 MFMA measured (hero_probe_mfma, %.2f GHz), HBM copy %.0f GB/s, read %.0f GB/s (hero_probe_hbm)"
          % (box["mfma_tflops"], box["ghz"], box["hbm_copy_gbps"], box["hbm_read_gbps"]))
    print("us per launch (hipGraph timing, fresh random operands per weight: nothing is cache-resident between launches beyond what one launch"
          " re-reads itself);\nalgorithmic bytes = every dY and X panel once + dW read-add-write (fp32); attainable = max(flops / measured MFMA "
          "peak, bytes / HBM rate) at the box's copy probe and at 6.3 TB/s\n")
    kinds = [k for k in ("noepi", "dmaonly", "mfmaonly") if k in data]
    hdr = "%-62s %7s %6s | %9s" % ("launch", "rows", "tiles", "product") + "".join(" %9s" % k for k in kinds) + " %9s | %7s %7s %8s %8s | %8s %8s | %6s %6s" % (
        "library", "GFLOP", "MB", "t_mfma", "t_hbm6.3", "prod/att", "loop/att", "TF/s", "frac")
    print(hdr)
    tot = {}
    for name, rows, shapes, per_step in LAUNCHES:
        a = p["rows"][name]
        fl = sum(2.0 * rows * n_ * k_ for n_, k_ in shapes)
        by = sum(2.0 * rows * (n_ + k_) + 8.0 * n_ * k_ for n_, k_ in shapes)
        t_m = fl / (box["mfma_tflops"] * 1e12) * 1e6
        t_h = by / 6.3e12 * 1e6
        t_c = by / (box["hbm_copy_gbps"] * 1e9) * 1e6
        att = max(t_m, t_h)
        loop = data["noepi"]["rows"][name]["us"] if "noepi" in data else float("nan")
        print("%-62s %7d %6d | %9.1f" % (name, rows, a["tiles"], a["us"]) + "".join(" %9.1f" % data[k]["rows"][name]["us"] for k in kinds) +
              " %9.1f | %7.1f %7.1f %8.1f %8.1f | %8.2f %8.2f | %6.0f %6.3f" % (
                  a["us_library"], fl / 1e9, by / 1e6, t_m, t_h, a["us"] / att, loop / att, fl / a["us"] / 1e6, fl / a["us"] / 1e6 / 2500.0))
        print("%-62s %7s %6s   attainable at the copy probe: %.1f us (bytes / %.0f GB/s = %.1f us)" % ("", "", "", max(t_m, t_c), box["hbm_copy_gbps"], t_c))
        for k, v in (("product", a["us"]), ("att", att), ("lib", a["us_library"]), ("loop", loop)):
            key = ("D2 " if name.startswith("D2") else "D4 ") + k
            tot[key] = tot.get(key, 0.0) + per_step * v
    for w in ("D2", "D4"):
        print("\n%s per micro-step: product %.3f ms, main loops alone %.3f ms, attainable %.3f ms, library %.3f ms -> product / attainable %.2f"
              % (w, tot[w + " product"] / 1e3, tot[w + " loop"] / 1e3, tot[w + " att"] / 1e3, tot[w + " lib"] / 1e3, tot[w + " product"] / tot[w + " att"]))


if __name__ == "__main__":
    if sys.argv[1] == "table":
        table(sys.argv[2:])
    else:
        measure(sys.argv[1], sys.argv[2])
