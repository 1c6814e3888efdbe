#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{ echo "== copy sources ragged"; timeout 300 python tools/lab/copy_sources.py ragged 2>&1 | tail -36; } > gpurun_out/r4_run11.log 2>&1
cat gpurun_out/r4_run11.log
