#!/bin/bash
# same-box A/B of whole trees: round-3 final, before the determinism commit, now
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
R=$PWD
{
for rep in 1 2; do
for t in .ab/9320138 .ab/01296a0 .; do
  echo "== $t"; (cd $t && timeout 300 python bench.py --no-cpu-baseline $([ $t = . ] && echo --no-secondary) 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])")
done
done
} > gpurun_out/r4_run9.log 2>&1
cat gpurun_out/r4_run9.log
