#!/bin/bash
# Same-box A/B of whole trees on ONE workload: ab_trees_w.sh WORKLOAD STEPS TREE TREE ...  (alternating, twice)
cd "$(dirname "$0")/../.."
w=$1; steps=$2; shift 2
for rep in 1 2; do
for t in "$@"; do
  ms=$(cd $t && timeout 400 python bench.py --workload $w --steps $steps --warmup 8 --no-cpu-baseline --no-secondary --no-box-probe --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$t $w $ms"
done
done
