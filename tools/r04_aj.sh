#!/bin/bash
# A/B: window plan built after the stem has been issued (its host work hidden behind the stem's kernels)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3 4; do
  run "plan after the stem (default)" A=1
  run "plan first" HFL_PLAN_LATE=0
done > $out/r04_aj_ab.log 2>&1
cat $out/r04_aj_ab.log
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "golden" > $out/r04_aj_tests.log 2>&1; tail -3 $out/r04_aj_tests.log
