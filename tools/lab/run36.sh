cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "layernorm or ln or embed" 2>&1 | tail -3
timeout 120 python tools/ln_bench.py 2>&1 | grep rows
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline | cut -c1-200
