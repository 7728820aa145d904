#!/bin/bash
# last check at HEAD: the whole GPU suite, smoke(), the default bench line
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests -m gpu -x -q --durations=8 > $out/r04_gputest.log 2>&1; tail -14 $out/r04_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r04_smoke.log 2>&1; tail -2 $out/r04_smoke.log
python bench.py > $out/r04_bench_default.json 2> $out/r04_bench_default.err; tail -c 400 $out/r04_bench_default.json
