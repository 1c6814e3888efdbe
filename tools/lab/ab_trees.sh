#!/bin/bash
# Same-box A/B of WHOLE trees (round 4: the boxes of the pool differ by up to 15 %, so "before / after" across gpurun calls
# means nothing).  Prepare on the build host:
#     for c in <old commit> ...; do n=$(git rev-parse --short $c); mkdir -p .ab/$n; git archive $c | tar -x -C .ab/$n; \
#         (cd .ab/$n && python -m hero_amd.build); done            # .ab/ is git-ignored but travels with gpurun
# then ONE call:  gpurun -- 'bash tools/lab/ab_trees.sh .ab/<sha> .ab/<sha> .'   (alternating, twice)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
trees=("$@"); [ ${#trees[@]} -eq 0 ] && trees=(.)
{
for rep in 1 2; do
for t in "${trees[@]}"; do
  echo "== $t"; (cd $t && timeout 300 python bench.py --no-cpu-baseline $(grep -q no-secondary bench.py && echo --no-secondary) 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])")
done
done
} > gpurun_out/ab_trees.log 2>&1
cat gpurun_out/ab_trees.log
