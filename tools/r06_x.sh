#!/bin/bash
# kernel-by-kernel timeline of the sequential parts of the forward: per-batch tables + stem, OctFormer stage, pyramid init, pooling head
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_x_trace -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r06_x_trace.log 2>&1
tr=$(find gpurun_out/r06_x_trace -name '*kernel_trace.csv' | head -1)
for ph in "tables" "OctFormer" "pyramid" "pooling" "iteration 5"; do
  python tools/forward_phases.py "$tr" "$ph" > "gpurun_out/r06_x_phases_$(echo $ph | tr ' ' '_').log" 2>&1
done
rm -rf gpurun_out/r06_x_trace
cat gpurun_out/r06_x_phases_tables.log | head -80
