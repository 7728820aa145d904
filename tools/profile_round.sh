#!/bin/bash
# One profiling pass of the default bench for profiles/: tools/profile_round.sh <tag>   (run on the GPU box)
#  1. rocprofv3 --kernel-trace --stats    -> gpurun_out/<tag>_stats/   (+ summary table; 18 model steps: 2 warm-up + 16 timed)
#  2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) -> <tag>_pmc_traffic.json
#  3. un-profiled default bench line (with the CPU baseline)              -> <tag>_bench_default.json
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python bench.py --steps 16 --warmup 2 --no-extras --no-cpu-baseline > $out/${tag}_stats.log 2>&1
stats=$(find $out/${tag}_stats -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/${tag}_kernel_stats.csv
grep '^{' $out/${tag}_stats.log > $out/${tag}_bench_under_profiler.json
python tools/summarize_rocprof.py $out/${tag}_kernel_stats.csv 18 > $out/${tag}_summary_table.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_$c -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/${tag}_pmc_$c.log 2>&1
done
f=$(find $out/${tag}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find $out/${tag}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$f" "$w" $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_summary.txt
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
tail -c 1500 $out/${tag}_bench_default.json
