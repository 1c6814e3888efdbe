#!/usr/bin/env python3
"""Idle time BETWEEN the kernels of a hipGraph replay of the step.  Reads the kernel_trace.csv of
    rocprofv3 --kernel-trace --output-format csv -d DIR -o g -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0
(graph replay: no --no-graph), takes the last `steps` replays (recognised by the AdamW launch that ends every 2nd micro-step)
and prints: span of an optimiser step, busy time (union of kernel intervals), the gap histogram, and the kernels in front of
the largest gaps.  usage: graph_gaps.py DIR"""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r[2]]
print("kernels in trace: %d, optimiser steps: %d" % (len(rows), len(ends)))
# steady state: between the 3rd-last and the last AdamW (two optimiser steps = 4 micro-steps)
a, b = ends[-3], ends[-1]
seg = rows[a + 1:b + 1]
span = seg[-1][1] - seg[0][0]
busy, cur_e = 0, seg[0][0]
gaps = []
for s, e, n in seg:
    if s > cur_e:
        gaps.append((s - cur_e, n))
    busy += max(0, e - max(s, cur_e))
    cur_e = max(cur_e, e)
nm = 4
print("launches per micro-step %.1f" % (len(seg) / nm))
print("span %.3f ms per micro-step, busy %.3f ms, idle %.3f ms (%.1f %%)" % (span / nm / 1e6, busy / nm / 1e6, (span - busy) / nm / 1e6, 100.0 * (span - busy) / span))
h = collections.Counter()
for g, _ in gaps:
    h[min(g // 1000, 20)] += 1
print("gap histogram (us: count per micro-step):", {k: round(v / nm, 1) for k, v in sorted(h.items())})
big = collections.Counter(); bigt = collections.Counter()
prev = None
for (s, e, n) in seg:
    if prev is not None and s > prev[1]:
        key = prev[2][:70] + "  ->  " + n[:70]
        big[key] += 1; bigt[key] += s - prev[1]
    prev = (s, e, n) if prev is None or e > prev[1] else prev
print("largest idle totals (us per micro-step):")
for k, v in sorted(bigt.items(), key=lambda kv: -kv[1])[:25]:
    print("  %7.2f us in %5.1f gaps  %s" % (v / nm / 1e3, big[k] / nm, k))
