#!/bin/bash
# timeline of the stem, the head and one iteration of the current default step
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
rocprofv3 --kernel-trace --output-format csv -d $out/r04_w_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_w_stats.log 2>&1
trace=$(find $out/r04_w_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "tables" > $out/r04_w_phases_stem.log 2>&1
python tools/forward_phases.py "$trace" "pooling" > $out/r04_w_phases_head.log 2>&1
python tools/forward_phases.py "$trace" "octf" > $out/r04_w_phases_octf.log 2>&1
python tools/forward_phases.py "$trace" "init" > $out/r04_w_phases_init.log 2>&1
rm -rf $out/r04_w_stats
grep -v "^    " $out/r04_w_phases_stem.log | cut -c1-220
