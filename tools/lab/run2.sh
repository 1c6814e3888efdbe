cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -15) > gpurun_out/ws_tests.log 2>&1
(timeout 300 python tools/gemm_bench.py all -1; timeout 300 python tools/gemm_bench.py all 8) > gpurun_out/ws_bench.log 2>&1
(timeout 300 python bench.py --steps 20 --no-cpu-baseline) > gpurun_out/ws_step.log 2>&1
tail -5 gpurun_out/ws_tests.log; cat gpurun_out/ws_bench.log; tail -2 gpurun_out/ws_step.log
