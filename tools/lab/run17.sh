cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_configs.py -q -k "two_rank or bench_two or graph_replay" 2>&1 | tail -8) > $OUT/fix_tests.log 2>&1
tail -4 $OUT/fix_tests.log
echo "--- default lib"; timeout 300 python tools/gemm_bench.py fwd -1 2>&1 | grep "M=12000"
echo "--- nt stores"; HERO_HIP_LIB=$GRAFT_REPO_ROOT/hero_amd/libhero_hip_nt.so timeout 300 python tools/gemm_bench.py fwd -1 2>&1 | grep "M=12000"
echo "--- default lib again"; timeout 300 python tools/gemm_bench.py fwd -1 2>&1 | grep "M=12000"
(HERO_HIP_LIB=$GRAFT_REPO_ROOT/hero_amd/libhero_hip_nt.so timeout 300 python bench.py --steps 20 --no-cpu-baseline) 2>&1 | tail -1 | cut -c1-160
(timeout 300 python bench.py --steps 20 --no-cpu-baseline) 2>&1 | tail -1 | cut -c1-160
