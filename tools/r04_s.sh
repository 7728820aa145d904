#!/bin/bash
# A/B: H-OSA iterations without the per-iteration join of the pyramid streams; timeline of the new schedule
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 900 python -m pytest tests/test_gpu_model.py -x -q > $out/r04_s_tests.log 2>&1; tail -5 $out/r04_s_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "no join (default)" A=1
  run "join every iteration" HFL_ITER_JOIN=1
  run "no join, cu reserve 16" HFL_VARIANTS=cu_reserve=16
  run "no join, mlp fused 1024" HFL_MLP_FUSED_MIN_ROWS=1024
done > $out/r04_s_ab.log 2>&1
cat $out/r04_s_ab.log
rocprofv3 --kernel-trace --output-format csv -d $out/r04_s_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_s_stats.log 2>&1
trace=$(find $out/r04_s_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_s_phases_it7.log 2>&1
rm -rf $out/r04_s_stats
cat $out/r04_s_phases_it7.log
