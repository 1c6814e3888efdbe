#!/bin/bash
# Same-box A/B of library builds on the D2 step: ab_libs.sh name1 name2 ...  ("product" = hero_amd/libhero_hip.so, else tools/lab/libhero_<name>.so),
# alternating, twice; prints ms per micro-step of a 60-step replayed run each.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = product ]; then lib=""; else lib=$PWD/tools/lab/libhero_$n.so; fi
    ms=$(HERO_HIP_LIB=$lib python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-box-probe --profile-steps 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$n $ms"
  done
done
