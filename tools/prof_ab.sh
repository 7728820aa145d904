#!/bin/bash
# A/B of one bench flag under rocprofv3 --kernel-trace --stats: tools/prof_ab.sh <tag> <bench args...>
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --steps 16 --warmup 2 --no-extras --no-cpu-baseline "$@" > gpurun_out/${tag}_stats.log 2>&1
stats=$(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1)
cp "$stats" gpurun_out/${tag}_kernel_stats.csv
python tools/summarize_rocprof.py gpurun_out/${tag}_kernel_stats.csv 18 > gpurun_out/${tag}_summary_table.md
head -16 gpurun_out/${tag}_summary_table.md; tail -1 gpurun_out/${tag}_summary_table.md
