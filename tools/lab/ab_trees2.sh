#!/bin/bash
# like ab_trees.sh, for D2 and D2r, 40 timed steps each: ab_trees2.sh TREE TREE ...
cd "$(dirname "$0")/../.."
R=$PWD
for rep in 1 2; do
for t in "$@"; do
  for w in D2 D2r; do
    ms=$(cd $t && timeout 300 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-box-probe --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$t $w $ms"
  done
done
done
