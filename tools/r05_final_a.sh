#!/bin/bash
# round-5 evidence, part A: full GPU test log, kernel traces (overlapped + serial), PMC traffic, default bench line
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests -m gpu -x -q --durations=10 > $out/r05_gputest.log 2>&1; tail -16 $out/r05_gputest.log
tools/profile_round.sh r05 > $out/r05_profile_round.log 2>&1
tools/profile_serial.sh r05 > $out/r05_profile_serial.log 2>&1
rm -rf $out/r05_stats $out/r05_serial_stats $out/r05_pmc_FETCH_SIZE $out/r05_pmc_WRITE_SIZE
head -24 $out/r05_serial_summary_table.md
tail -c 600 $out/r05_bench_default.json
