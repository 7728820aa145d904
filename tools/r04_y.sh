#!/bin/bash
# fused MLP with the two waves of a SIMD one stage apart; pooling with two workgroups per CU
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 600 python tools/mlp_waves_probe.py lag > $out/r04_y_probe.log 2>&1; cat $out/r04_y_probe.log | tail -10
HFL_VARIANTS=mlp_lag=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp_fused or attn_pool" > $out/r04_y_tests.log 2>&1; tail -5 $out/r04_y_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "lock-step (default)" A=1
  run "one stage apart" HFL_VARIANTS=mlp_lag=1
done > $out/r04_y_ab.log 2>&1
cat $out/r04_y_ab.log
