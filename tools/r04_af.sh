#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
rocprofv3 --kernel-trace --output-format csv -d $out/r04_af_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_af_stats.log 2>&1
trace=$(find $out/r04_af_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_af_phases_it7.log 2>&1
python tools/forward_phases.py "$trace" "pooling" > $out/r04_af_phases_head.log 2>&1
python tools/forward_phases.py "$trace" "iteration 1" > $out/r04_af_phases_it1.log 2>&1
rm -rf $out/r04_af_stats
cat $out/r04_af_phases_it7.log | cut -c1-150
