#!/bin/bash
# Same-box A/B of the round-4 final tree (.ab/64307b1) against this tree: D2 headline and the ragged D2r line, alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$PWD
one() {  # tree, extra args
  (cd $1 && timeout 300 python bench.py --no-cpu-baseline --no-secondary $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %-10s %8.3f ms %9.1f videos/s' % ('$1', '$2'.replace('--workload ','') or 'D2', d['ms_per_step'], d['value']))")
}
{
for rep in 1 2; do
  one .ab/64307b1 ""; one . "--no-box-probe"
  one .ab/64307b1 "--workload D2r"; one . "--workload D2r"
done
one .ab/64307b1 "--workload D3"; one . "--workload D3"
} > gpurun_out/r5_ab.log 2>&1
cat gpurun_out/r5_ab.log
