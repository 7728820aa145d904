#!/bin/bash
# A/B: fused MLP / fused LN->qkv launches on the coarse pyramid levels (14 276 and 2 092 rows) too
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'])"
}
for i in 1 2 3; do
  run "default (24576 rows)" A=1
  run "mlp fused from 8192 rows" HFL_MLP_FUSED_MIN_ROWS=8192
  run "mlp fused from 1024 rows" HFL_MLP_FUSED_MIN_ROWS=1024
  run "mlp + qkv fused from 1024" HFL_MLP_FUSED_MIN_ROWS=1024 HFL_QKV_FUSED_MIN_ROWS=1024
  run "mlp 1024, qkv 8192" HFL_MLP_FUSED_MIN_ROWS=1024 HFL_QKV_FUSED_MIN_ROWS=8192
done > $out/r04_q_ab.log 2>&1
cat $out/r04_q_ab.log
