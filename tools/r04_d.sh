#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $out/r04_d_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_d_stats.log 2>&1
trace=$(find $out/r04_d_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "tables" > $out/r04_d_phases_stem.log 2>&1
python tools/forward_phases.py "$trace" "pooling" > $out/r04_d_phases_head.log 2>&1
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_d_phases_it7.log 2>&1
python tools/forward_phases.py "$trace" "octf" > $out/r04_d_phases_octf.log 2>&1
rm -rf $out/r04_d_stats
python tools/host_issue_probe.py > $out/r04_d_host_issue.log 2>&1
