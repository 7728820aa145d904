#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests -m gpu -x -q --durations=12 > $out/r04_j_gputest.log 2>&1; tail -20 $out/r04_j_gputest.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'])"
}
for i in 1 2 3; do
  run "default" A=1
  run "gather kernel (not in GEMM)" HFL_GATHER_IN_GEMM=0
  run "rtsa mlp unfused" HFL_RTSA_MLP_FUSED=0
done > $out/r04_j_ab.log 2>&1
cat $out/r04_j_ab.log
