#!/usr/bin/env python3
"""Condense rocprofv3 output into the small summaries committed under profiles/.

  python tools/profile_summary.py stats <dir-with-*_kernel_stats.csv> <micro-steps-in-trace> <out.csv> ["header note"]
  python tools/profile_summary.py steady <dir-with-*_kernel_trace.csv> <optimiser-steps-to-skip> <out.csv> ["header note"]
  python tools/profile_summary.py pmc <fetch-dir> <write-dir> <out.json>
  python tools/profile_summary.py mfma <pmc-dir> <out.json>

`stats`: per-kernel calls / total ms / average us PER MICRO-STEP from `rocprofv3 --kernel-trace --stats`.
`steady`: the same table from the per-dispatch kernel trace, over STEADY-STATE micro-steps only: the window
        from the (skip+1)-th to the last `adamw_multi_kernel` launch = whole optimiser steps of 2 micro-steps each, so the
        one-time fills / copies / lazy-state launches of the first steps (and of model construction) are not averaged in.
`pmc`:  per-kernel average FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them; collected in two
        separate --pmc passes because the TCC block cannot hold both) and the corrected HBM bytes per
        launch: FETCH_SIZE x 2 (gfx950 counts 128-B read requests as 64 B, MI355X_MICROARCH.md
        "HBM") + WRITE_SIZE.
"""
import collections
import csv
import glob
import json
import os
import sys


def _one(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (pat, d))
    return hits[0]


def stats(d, steps, out, note=""):
    rows = list(csv.DictReader(open(_one(d, "*kernel_stats.csv"))))
    steps = float(steps)
    total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
    with open(out, "w") as f:
        if note:
            f.write("# %s\n" % note)
        f.write("# %g micro-steps in the trace; total kernel time %.2f ms/step\n" % (steps, total))
        f.write("name,calls_per_step,total_ms_per_step,avg_us,pct\n")
        for r in rows:
            f.write('"%s",%.1f,%.3f,%.2f,%s\n' % (r["Name"], float(r["Calls"]) / steps,
                                                   float(r["TotalDurationNs"]) / steps / 1e6,
                                                   float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print("kernel time %.2f ms/step -> %s" % (total, out))


def steady(d, skip, out, note="", accum=2):
    rows = []
    with open(_one(d, "*kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r[2]]
    skip = int(skip)
    if len(marks) - skip < 2:
        raise SystemExit("steady: %d optimiser steps in the trace, cannot skip %d" % (len(marks), skip))
    a, b = marks[skip], marks[-1]
    seg = rows[a:b]                                   # [AdamW #skip, AdamW #last): whole optimiser steps
    steps = float((len(marks) - 1 - skip) * accum)
    acc = collections.OrderedDict()
    for s_, e_, n in seg:
        v = acc.setdefault(n, [0, 0])
        v[0] += 1
        v[1] += e_ - s_
    total = sum(v[1] for v in acc.values())
    span = rows[b][0] - rows[a][0]
    with open(out, "w") as f:
        if note:
            f.write("# %s\n" % note)
        f.write("# STEADY STATE: %g micro-steps between optimiser steps %d and %d of the trace (the first %d optimiser steps and "
                "everything before them are skipped); kernel time %.3f ms/step in %.1f launches/step, wall span %.3f ms/step\n"
                % (steps, skip + 1, len(marks), skip, total / steps / 1e6, len(seg) / steps, span / steps / 1e6))
        f.write("name,calls_per_step,total_ms_per_step,avg_us,pct\n")
        for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%.1f,%.3f,%.2f,%.4f\n' % (n, c / steps, t / steps / 1e6, t / c / 1e3, 100.0 * t / total))
    print("steady state: kernel time %.3f ms/step, %.1f launches/step -> %s" % (total / steps / 1e6, len(seg) / steps, out))


def _counter(d, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(_one(d, "*counter_collection.csv"))):
        if r["Counter_Name"] == name:
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def csrc_sha16():
    """Hash of the kernel sources the profile was taken with (bench.py stamps `roofline.traffic` with it)."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hero_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def pmc(fd, wd, out):
    fe, wr = _counter(fd, "FETCH_SIZE"), _counter(wd, "WRITE_SIZE")
    res = {"_meta": {"csrc_sha16": csrc_sha16(), "note": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, separate --pmc "
                     "passes; hbm_bytes_per_launch_corrected = (2 x FETCH + WRITE) x 1024 (gfx950 counts 128-B reads as 64 B)"}}
    for k, (n, tot) in fe.items():
        if "hero::" not in k or k not in wr:
            continue
        f_kb = tot / n
        w_kb = wr[k][1] / wr[k][0]
        res[k] = {"launches": n, "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb,
                  "hbm_bytes_per_launch_corrected": (2.0 * f_kb + w_kb) * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    print("%d kernels -> %s" % (len(res), out))


def mfma(d, out):
    """MFMA utilisation per kernel from one --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and
    GRBM_GUI_ACTIVE: busy cycles of the matrix pipes (summed over the chip's 1024 SIMDs; 32 per
    v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md) / (4 SIMDs x 256 CUs x GPU-active cycles of the dispatch).
    rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (checked: value / 8 / kernel duration = 2.2-2.3 GHz, and the busy
    cycles equal 32 x the kernel's MFMA count), hence the division by 8."""
    busy, act, cu = (_counter(d, "SQ_VALU_MFMA_BUSY_CYCLES"), _counter(d, "GRBM_GUI_ACTIVE"),
                     _counter(d, "SQ_BUSY_CU_CYCLES"))
    res = {}
    for k, (n, tot) in busy.items():
        if "hero::" not in k or k not in act or tot == 0:
            continue
        a = act[k][1] / act[k][0]
        res[k] = {"launches": n, "SQ_VALU_MFMA_BUSY_CYCLES_avg": tot / n, "GRBM_GUI_ACTIVE_avg": a,
                  "SQ_BUSY_CU_CYCLES_avg": cu[k][1] / cu[k][0] if k in cu else None,
                  "gpu_active_cycles": a / 8.0, "mfma_util": tot / n / (1024.0 * a / 8.0)}
    json.dump(res, open(out, "w"), indent=1)
    print("%d kernels -> %s" % (len(res), out))


if __name__ == "__main__":
    if sys.argv[1] == "mfma":
        mfma(*sys.argv[2:])
    elif sys.argv[1] == "stats":
        stats(*sys.argv[2:])
    elif sys.argv[1] == "steady":
        steady(*sys.argv[2:])
    elif sys.argv[1] == "pmc":
        pmc(*sys.argv[2:])
    else:
        raise SystemExit(__doc__)
