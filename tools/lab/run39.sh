cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "layernorm or ln or embed or repr or reference or train" 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline | cut -c1-200
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cm -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2>&1
grep -h "ln_fwd" /tmp/cm/*/*kernel_stats.csv /tmp/cm/*kernel_stats.csv 2>/dev/null | cut -c1-150
