#!/usr/bin/env python3
"""Timeline of the grouped stream-K wgrad kernel (library built with -DHERO_WS_TRACE): workgroup 0, compute waves."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hero_amd import _lib as L
rows = 12000; dt = torch.bfloat16
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
dys = [torch.randn(rows, n, device="cuda").to(dt) for n, _ in shapes]
xs = [torch.randn(rows, k, device="cuda").to(dt) for _, k in shapes]
outs = [torch.zeros(n, k, device="cuda") for n, k in shapes]
pr = (L.WgradProblem * 4)(*[L.WgradProblem(dys[i].data_ptr(), xs[i].data_ptr(), outs[i].data_ptr(), shapes[i][0], shapes[i][1],
                                            shapes[i][0], shapes[i][1], shapes[i][1], 4) for i in range(4)])
for _ in range(3):
    L.check(L.lib().hero_wgrad_group(pr, 4, rows, L.BF16, L.stream()))
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 16 * 8))()
assert L.lib().hero_ws_trace_read(buf) == 0
t = lambda i, e, w: buf[(i * 16 + e) * 8 + w]
t0 = min(t(0, 0, w) for w in range(4))
print("workgroup 0 of hero_wgrad_group (4 problems, 12000 rows): cycles since start; compute waves 0-3")
for i in range(4):
    if not t(i, 0, 0): continue
    nk = t(i, 3, 0)
    st, ml, ae = [t(i, 0, w) - t0 for w in range(4)], [t(i, 1, w) - t0 for w in range(4)], [t(i, 14, w) - t0 for w in range(4)]
    print("segment %d: %3d k-steps  start %s  loop end %s  atomics end %s  -> %.0f cycles/step, atomics %.0f cycles"
          % (i, nk, st, ml, ae, (ml[0] - st[0]) / max(nk, 1), ae[0] - ml[0]))
