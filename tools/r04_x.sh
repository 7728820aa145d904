#!/bin/bash
# A/B: tap lists on their own stream (host not blocked by the previous step's tail) under the join-free schedule
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline $BARGS 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "default" A=1
  run "tap stream" HFL_TAP_STREAM=1
  BARGS=--resident-plan run "resident plan" A=1
done > $out/r04_x_ab.log 2>&1
cat $out/r04_x_ab.log
