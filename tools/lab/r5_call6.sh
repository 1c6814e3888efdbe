#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$PWD
{
timeout 200 python tools/lab/attn_ab.py 2>&1 | grep ppw
HERO_HIP_LIB=$R/tools/lab/libhero_late.so timeout 200 python tools/lab/attn_ab.py 2>&1 | grep ppw
(cd .ab/64307b1 && timeout 200 python $R/tools/lab/attn_ab.py old 2>&1 | grep ppw | sed 's/^product/round-4 tree/')
} > gpurun_out/c6_attn_ab.txt 2>&1
(timeout 300 python tools/lab/feedprobe.py 2>&1 | grep -v Warning | tail -22) > gpurun_out/c6_feedprobe.txt
cat gpurun_out/c6_attn_ab.txt gpurun_out/c6_feedprobe.txt
