#!/bin/bash
# like ab_trees2.sh over D2, D2r, D3 (40 timed steps) and D4 at 256 videos (4 steps): ab_trees3.sh TREE TREE ...
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for t in "$@"; do
  for w in D2 D2r D3 D4; do
    extra="--steps 40 --warmup 8"; [ $w = D4 ] && extra="--videos 256 --steps 4 --warmup 2"
    ms=$(cd $t && timeout 400 python bench.py --workload $w $extra --no-cpu-baseline --no-secondary --no-box-probe --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$t $w $ms"
  done
done
done
