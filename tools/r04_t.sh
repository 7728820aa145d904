#!/bin/bash
# fused MLP: 4 waves x 2 row tiles (one wave per SIMD, 512 registers) against 8 waves x 1
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 600 python tools/mlp_waves_probe.py > $out/r04_t_probe.log 2>&1; cat $out/r04_t_probe.log | tail -12
HFL_VARIANTS=mlp_waves=4 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp_fused" > $out/r04_t_tests.log 2>&1; tail -5 $out/r04_t_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "8 waves (default)" A=1
  run "4 waves" HFL_VARIANTS=mlp_waves=4
done > $out/r04_t_ab.log 2>&1
cat $out/r04_t_ab.log
