#!/bin/bash
# Alternate builds of libhero_hip.so with the lab ablations of gemm_ws.hip compiled in (timing only; results are
# garbage): tools/lab/libhero_<name>.so, selected at run time with HERO_HIP_LIB.  usage: build_variants.sh NAME:-DFLAG[,-DFLAG] ...
set -e
cd "$(dirname "$0")/../.."
python -m hero_amd.build > /dev/null
OBJ=hero_amd/csrc/_obj
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c hero_amd/csrc/gemm_ws.hip -o /tmp/gemm_ws_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  objs=$(ls $OBJ/*.o | grep -v gemm_ws.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libhero_$name.so $objs /tmp/gemm_ws_$name.o
  echo tools/lab/libhero_$name.so
done
