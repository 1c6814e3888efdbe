#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{ echo "== full gpu suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6; echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3; } > gpurun_out/r4_final_tests.log 2>&1
cat gpurun_out/r4_final_tests.log
bash tools/lab/final_profiles.sh r04 2>&1 | tail -40
