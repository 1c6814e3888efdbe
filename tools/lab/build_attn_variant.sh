#!/bin/bash
# tools/lab/libhero_<name>.so with extra -D flags on attention_mfma.hip only (HERO_HIP_LIB selects it).  usage: NAME -DFLAG ...
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m hero_amd.build > /dev/null
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c hero_amd/csrc/attention_mfma.hip -o /tmp/attention_mfma_$name.o
objs=$(ls hero_amd/csrc/_obj/*.o | grep -v attention_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libhero_$name.so $objs /tmp/attention_mfma_$name.o -ldl
echo tools/lab/libhero_$name.so
