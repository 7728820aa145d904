#!/bin/bash
# A/B: cross-stream dependencies of the H-OSA iterations through device flags instead of events
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "events (default)" A=1
  run "flag hops" HFL_FLAG_HOPS=1
done > $out/r04_al_ab.log 2>&1
cat $out/r04_al_ab.log
HFL_FLAG_HOPS=1 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or native_block or early_phase" > $out/r04_al_tests.log 2>&1; tail -3 $out/r04_al_tests.log
