#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 python tools/hip_graph_native_probe.py > gpurun_out/r06_y_hip_graph_native_probe.log 2>&1
echo "exit code $?" >> gpurun_out/r06_y_hip_graph_native_probe.log
cat gpurun_out/r06_y_hip_graph_native_probe.log | tail -30
