#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests/test_gpu_kernels.py -x -q -k "ln_mlp_fused or ln_qkv or linear_x3 or split_precision or layer_norm or tap_lists" > $out/r04_h_tests.log 2>&1; tail -3 $out/r04_h_tests.log
python -m pytest tests/test_gpu_model.py -x -q -k "golden or native_block or early_phase" > $out/r04_h_tests2.log 2>&1; tail -3 $out/r04_h_tests2.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2; do
  run "default" A=1
  run "tap stream off" HFL_TAP_STREAM=0
  run "static units" HFL_VARIANTS=dynamic_units=0
  run "x3 ring off" HFL_VARIANTS=x3_ring=0
  run "cu reserve 16" HFL_VARIANTS=cu_reserve=16
  run "cu reserve 32" HFL_VARIANTS=cu_reserve=32
  run "rtsa mlp fused" HFL_RTSA_MLP_FUSED=1
  run "mlp fused all rows" HFL_MLP_FUSED_MIN_ROWS=0
  run "qkv fused half-full rule" HFL_QKV_FUSED_MIN_FILL=0.5
  run "no tail split" HFL_VARIANTS=tail_split=0
done > $out/r04_h_ab.log 2>&1
cat $out/r04_h_ab.log
