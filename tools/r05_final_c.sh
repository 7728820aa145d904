#!/bin/bash
# round-5 evidence, part C (after the training-path CPE change touched csrc/dwconv.hip): CPE counter survey with its source
# stamp, full GPU test log, default bench line
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
tools/cpe_counters.sh r05 > $out/r05_cpe_counters.log 2>&1
python tools/stamp_sources.py hotformerloc_amd/csrc/dwconv.hip >> $out/r05_cpe_counters.txt
python -m pytest tests -m gpu -x -q --durations=10 > $out/r05_gputest.log 2>&1; tail -16 $out/r05_gputest.log
python bench.py > $out/r05_bench_default.json 2> $out/r05_bench_default.err
tail -c 600 $out/r05_bench_default.json
