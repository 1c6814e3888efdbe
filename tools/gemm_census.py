#!/usr/bin/env python3
"""Census of the GEMM launches of one bench micro-step: every (shape, layouts, epilogue) hero_gemm sees, how often,
and the time of each distinct one under the default geometry choice and under forced geometries
(hero_gemm_force_config 0..3 = the 4-wave tiles, 9 / 10 = wave-specialised 192 x 192 / 128 x 192 where legal).

    python tools/gemm_census.py [max_M]      # only shapes with M <= max_M are timed under forced geometries
"""
import collections, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hero_amd
from hero_amd import functional as HF, _lib as L
from hero_amd.step import TrainStep
from hero_amd.synth import make_batch

max_m = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
hero_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
cfgp = "/tmp/hero_census_cfg.json"
json.dump(bench.HERO_BASE, open(cfgp, "w"))
model = bench.build_model(dev, cfgp)
tr = TrainStep(model, use_graph=False)
batch = make_batch("D2", vfeat_dim=bench.VFEAT, vocab=50272, seed=1, device=dev)
for _ in range(3):
    tr.micro_step(batch)
torch.cuda.synchronize()

seen = collections.OrderedDict()
orig = HF.k_gemm


def spy(A, B, Cm, M, N, K, lda, ldb, ldc, al, bl, dtype_code, bias=None, residual=None, aux=None, act=L.ACT_NONE,
        out_f32=False, beta=0.0, split_k=1, drop=None, colsum=None):
    key = (M, N, K, al, bl, dtype_code, bias is not None, residual is not None, aux is not None, int(act), bool(out_f32),
           float(beta), int(split_k), drop is not None, colsum is not None)
    seen[key] = seen.get(key, 0) + 1
    return orig(A, B, Cm, M, N, K, lda, ldb, ldc, al, bl, dtype_code, bias=bias, residual=residual, aux=aux, act=act,
                out_f32=out_f32, beta=beta, split_k=split_k, drop=drop, colsum=colsum)


HF.k_gemm = spy
tr.micro_step(batch)
torch.cuda.synchronize()
HF.k_gemm = orig


def bench_one(key, cfg, reps=30):
    M, N, K, al, bl, dt, hb, hr, ha, act, of32, beta, split, hd, hc = key
    tdt = torch.bfloat16 if dt == L.BF16 else torch.float32
    a_shape = (M, K) if al == L.LAYOUT_K else (K, M)
    b_shape = (N, K) if bl == L.LAYOUT_K else (K, N)
    A = (torch.randn(a_shape, device=dev) * 0.1).to(tdt)
    B = (torch.randn(b_shape, device=dev) * 0.1).to(tdt)
    Cm = torch.zeros((M, N), device=dev, dtype=torch.float32 if of32 else tdt)
    bias = torch.zeros(N, device=dev, dtype=torch.float32) if hb else None
    res = torch.zeros((M, N), device=dev, dtype=tdt) if hr else None
    aux = torch.zeros((M, N), device=dev, dtype=tdt) if ha else None
    cs = torch.zeros(N, device=dev, dtype=torch.float32) if hc else None
    L.lib().hero_gemm_force_config(cfg)
    try:
        def run():
            orig(A, B, Cm, M, N, K, a_shape[1], b_shape[1], N, al, bl, dt, bias=bias, residual=res, aux=aux, act=act,
                 out_f32=of32, beta=beta, split_k=split, colsum=cs)
        t_end = time.time() + 0.15                     # past the clock ramp
        while time.time() < t_end:
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000.0 / reps
    except Exception as ex:                            # an illegal forced geometry
        return None
    finally:
        L.lib().hero_gemm_force_config(-1)


LAY = {L.LAYOUT_K: "K", L.LAYOUT_O: "O"}
print("%6s %6s %6s  lay dt  epilogue                      calls   default  " % ("M", "N", "K") + "  ".join("cfg%d" % c for c in (0, 1, 3, 8, 10, 13, 14)))
tot = collections.Counter()
for key, n in seen.items():
    M, N, K, al, bl, dt, hb, hr, ha, act, of32, beta, split, hd, hc = key
    epi = "+".join(x for x, f in (("bias", hb), ("res", hr), ("aux", ha), ("act%d" % act, act), ("f32out", of32), ("beta%g" % beta, beta),
                                  ("split%d" % split, split > 1), ("drop", hd), ("colsum", hc)) if f) or "-"
    t0 = bench_one(key, -1)
    forced = []
    if M <= max_m and not of32:
        for c in (0, 1, 3, 8, 10, 13, 14):
            t = bench_one(key, c)
            forced.append("%6.1f" % t if t is not None else "     -")
    gf = 2.0 * M * N * K / 1e9
    print("%6d %6d %6d  %s,%s %s  %-28s %5d  %7.1f us %5.0f TF/s  %s" % (M, N, K, LAY[al], LAY[bl], "bf" if dt == L.BF16 else "f32", epi, n, t0,
                                                                      gf / t0 * 1e3 if t0 else 0, " ".join(forced)))
    tot["all"] += n * t0
    if M <= max_m:
        tot["small"] += n * t0
print("sum of calls x isolated time: all %.2f ms, M <= %d: %.2f ms" % (tot["all"] / 1e3, max_m, tot["small"] / 1e3))
