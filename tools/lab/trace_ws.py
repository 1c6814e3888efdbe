#!/usr/bin/env python3
"""Timeline of the wave-specialised K,K GEMM (library built with -DHERO_WS_TRACE): s_memtime stamps of workgroup 0's
first items, per wave.  usage: trace_ws.py N K [epilogue: bias|gelu|res] [force config]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import functional as HF, _lib as L
N, K = int(sys.argv[1]), int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "bias"
M = 12000; dt = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt); b = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(dt); aux = torch.empty(M, N, device="cuda", dtype=dt)
L.lib().hero_gemm_force_config(int(sys.argv[4]) if len(sys.argv) > 4 else 9)
def run():
    if kind == "gelu": return HF.k_linear(x, w, b, act=L.ACT_GELU, aux=aux)
    if kind == "res": return HF.k_linear(x, w, b, residual=res, drop=HF.RNG.make(0.1, True, x.device))
    return HF.k_linear(x, w, b)
for _ in range(3): run()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 16 * 8))()
assert L.lib().hero_ws_trace_read(buf) == 0
t = [[[buf[(i * 16 + e) * 8 + wv] for wv in range(8)] for e in range(16)] for i in range(4)]
t0 = min(v for v in t[0][0] if v)
tick = 0.01   # printed unit = 100 shader-clock cycles (s_memtime runs at the core clock here: ~0.045 us per unit at 2.2 GHz)
names = {0: "item start", 15: "acc ready", 1: "main loop end", 14: "epilogue end"}
for p in range(3):
    names.update({2 + 4 * p: "p%d prefetched" % p, 3 + 4 * p: "p%d staged" % p, 4 + 4 * p: "p%d after E1" % p, 5 + 4 * p: "p%d rows done" % p})
print("N=%d K=%d %s: hundreds of shader cycles since the first stamp; waves 0-3 compute, 4-7 loaders" % (N, K, kind))
for i in range(4):
    for e in [0, 15] + list(range(1, 15)):
        if e not in names: continue
        row = t[i][e]
        if any(row): print("item %d %-16s " % (i, names[e]) + " ".join("%7.2f" % ((v - t0) * tick) if v else "      -" for v in row))
