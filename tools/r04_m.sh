#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 300 python tools/attn_fused_probe.py > $out/r04_m_probe.log 2>&1; tail -5 $out/r04_m_probe.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_fused" > $out/r04_m_tests.log 2>&1; tail -15 $out/r04_m_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'])"
}
for i in 1 2; do
  run "default" A=1
  run "x3 ring 3 stages" HFL_VARIANTS=x3_ring=3
  run "x3 ring 2 stages" HFL_VARIANTS=x3_ring=2
done > $out/r04_m_ab.log 2>&1
cat $out/r04_m_ab.log
