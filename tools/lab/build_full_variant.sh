#!/bin/bash
# A complete alternate build of libhero_hip.so with extra -D flags on EVERY source (build_variants.sh recompiles
# gemm_ws.hip only): tools/lab/libhero_<name>.so, selected at run time with HERO_HIP_LIB.
#   usage: build_full_variant.sh NAME -DFLAG [-DFLAG ...]
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
D=/tmp/hero_variant_$name; mkdir -p $D
for s in api.cpp gemm.hip gemm_ws.hip layernorm.hip attention.hip attention_mfma.hip attention_mfma_long.hip rows.hip head.hip loss.hip collate.hip; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c hero_amd/csrc/$s -o $D/${s%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libhero_$name.so $D/*.o
echo tools/lab/libhero_$name.so
