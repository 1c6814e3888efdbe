#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{ timeout 300 python tools/lab/cast_sources.py 2>&1 | tail -30; } > gpurun_out/r4_run15.log 2>&1
cat gpurun_out/r4_run15.log
