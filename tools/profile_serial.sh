#!/bin/bash
# Kernel trace of the default bench workload with the step's launch schedule on ONE stream (bench.py --serial-streams): every
# kernel runs alone, so its average duration is the kernel's own (what bench.py's roofline leg times with HIP events).
#   tools/profile_serial.sh <tag>   -> gpurun_out/<tag>_serial_kernel_stats.csv, <tag>_serial_summary_table.md
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_serial_stats -- python bench.py --serial-streams --steps 16 --warmup 2 --no-extras --no-cpu-baseline > $out/${tag}_serial_stats.log 2>&1
stats=$(find $out/${tag}_serial_stats -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/${tag}_serial_kernel_stats.csv
grep '^{' $out/${tag}_serial_stats.log > $out/${tag}_serial_bench_under_profiler.json
python tools/summarize_rocprof.py $out/${tag}_serial_kernel_stats.csv 18 > $out/${tag}_serial_summary_table.md
head -14 $out/${tag}_serial_summary_table.md
