#!/bin/bash
# round 5, first GPU call: full GPU suite, the bench line, the steady-state aten sources, the GEMM ceiling, the wgrad A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/c1_tests.log 2>&1
(timeout 600 python bench.py) > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
(timeout 200 python tools/torch_fill_sources.py 2>&1 | tail -45) > gpurun_out/c1_fills.txt
(timeout 200 python tools/torch_fill_sources.py ragged 2>&1 | tail -45) > gpurun_out/c1_fills_ragged.txt
bash tools/lab/gemm_ceiling.sh > /dev/null 2>&1
(timeout 400 python tools/lab/ab_wgrad.py 2>&1 | grep -v Warning | tail -12) > gpurun_out/c1_ab_wgrad.txt
tail -5 gpurun_out/c1_tests.log; cut -c1-600 gpurun_out/c1_bench.json; cat gpurun_out/c1_ab_wgrad.txt
