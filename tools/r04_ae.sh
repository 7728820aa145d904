#!/bin/bash
# relay rows' LN1 -> qkv as the fused launch (features split), ahead of their copy; bench with per-rank CPU pinning
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "qkv" > $out/r04_ae_tests.log 2>&1; tail -4 $out/r04_ae_tests.log
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q > $out/r04_ae_tests2.log 2>&1; tail -4 $out/r04_ae_tests2.log
run() { # label, args...
  label=$1; shift
  python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'], j['config'].get('host_affinity'))"
}
for i in 1 2 3; do
  run "default (pinned)"
  run "unpinned" --pin-cores 0
done > $out/r04_ae_ab.log 2>&1
cat $out/r04_ae_ab.log
