#!/bin/bash
# round-4 evidence, part A: kernel traces (overlapped + serial), PMC traffic, default bench line, full GPU test log
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests -m gpu -x -q --durations=10 > $out/r04_gputest.log 2>&1; tail -16 $out/r04_gputest.log
tools/profile_round.sh r04 > $out/r04_profile_round.log 2>&1
tools/profile_serial.sh r04 > $out/r04_profile_serial.log 2>&1
rm -rf $out/r04_stats $out/r04_serial_stats $out/r04_pmc_FETCH_SIZE $out/r04_pmc_WRITE_SIZE
head -24 $out/r04_serial_summary_table.md
tail -c 600 $out/r04_bench_default.json
