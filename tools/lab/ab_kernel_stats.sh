#!/bin/bash
# Steady-state per-kernel stats of the D2 step for several trees on ONE box: ab_kernel_stats.sh TREE TREE ... -> gpurun_out/abks_<n>.csv
cd "$(dirname "$0")/../.."
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for t in "$@"; do
  i=$((i+1))
  T=$(cd $t && pwd)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abks_$i -o s -- python $T/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-graph --no-box-probe > /dev/null 2> $OUT/abks_$i.log)
  python tools/profile_summary.py steady $OUT/abks_$i 2 $OUT/abks_$i.csv "tree $t" > /dev/null
  find $OUT/abks_$i -name "*kernel_trace.csv" -delete
  echo "== $t"; head -2 $OUT/abks_$i.csv | tail -1 | cut -c1-200
done
python - <<'PY'
import csv, sys, glob, collections
rows = {}
files = sorted(glob.glob("gpurun_out/abks_[0-9].csv"))
for f in files:
    for r in csv.reader(l for l in open(f) if not l.startswith("#")):
        if r[0] == "name":
            continue
        rows.setdefault(r[0], {})[f] = (float(r[1]), float(r[2]))
print("%-90s" % "kernel", "  ".join("%18s" % f.split("/")[-1] for f in files), "  diff(last-first) us")
tot = collections.Counter()
for k, v in sorted(rows.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
    a = [v.get(f, (0, 0.0)) for f in files]
    for f, x in zip(files, a):
        tot[f] += x[1]
    d = (a[-1][1] - a[0][1]) * 1e3
    if max(x[1] for x in a) > 0.008 or abs(d) > 3:
        print("%-90s" % k[:90], "  ".join("%5.1f x %8.3f ms" % x for x in a), "  %+7.1f" % d)
print("TOTAL", {f.split("/")[-1]: round(t, 3) for f, t in tot.items()})
PY
