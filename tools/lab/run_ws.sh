#!/bin/bash
# run every ws_* lab binary; results to gpurun_out/ws.log
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
for b in ws_*; do
  [ -x "$b" ] || continue
  echo "== $b"; timeout 120 ./$b
done 2>&1 | tee ../../gpurun_out/ws.log
