cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT /tmp/dw
export MASTER_ADDR=127.0.0.1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29551 tests/dist_worker.py /tmp/dw none > $OUT/dw_group.log 2>&1
grep -n "mismatching\|AssertionError" $OUT/dw_group.log | cut -c1-1200 | head -6
HERO_NOGROUP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29552 tests/dist_worker.py /tmp/dw none > $OUT/dw_nogroup.log 2>&1
echo "nogroup rc=$?"; grep -n "mismatching\|AssertionError" $OUT/dw_nogroup.log | cut -c1-600 | head -4
