#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training bench: tools/prof_train.sh <tag> <bench args...>
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --train --config cs-wild-places --steps 4 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/${tag}_stats.log 2>&1
stats=$(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1)
cp "$stats" gpurun_out/${tag}_kernel_stats.csv
python tools/summarize_rocprof.py gpurun_out/${tag}_kernel_stats.csv 5 > gpurun_out/${tag}_summary_table.md
head -40 gpurun_out/${tag}_summary_table.md; tail -1 gpurun_out/${tag}_summary_table.md
grep '^{' gpurun_out/${tag}_stats.log | cut -c1-200
