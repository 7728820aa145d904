#!/bin/bash
# relay-first schedule again, now without the per-iteration join
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "relay_first or early_phase" > $out/r04_ah_tests.log 2>&1; tail -4 $out/r04_ah_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "default" A=1
  run "relay first" HFL_RELAY_FIRST=1
done > $out/r04_ah_ab.log 2>&1
cat $out/r04_ah_ab.log
HFL_RELAY_FIRST=1 rocprofv3 --kernel-trace --output-format csv -d $out/r04_ah_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_ah_stats.log 2>&1
trace=$(find $out/r04_ah_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_ah_phases_it7.log 2>&1
rm -rf $out/r04_ah_stats
grep -v "^    " $out/r04_ah_phases_it7.log | cut -c1-100
grep "^    " $out/r04_ah_phases_it7.log | cut -c1-110
