#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests/test_gpu_model.py -x -q -k "relay_first or golden" > $out/r04_l_tests.log 2>&1; tail -3 $out/r04_l_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'])"
}
for i in 1 2 3; do
  run "default (relay first on rs)" A=1
  run "relay first off" HFL_RELAY_FIRST=0
  run "relay first + mlp reserve 16" HFL_VARIANTS=mlp_reserve=16
done > $out/r04_l_ab.log 2>&1
cat $out/r04_l_ab.log
HFL_KNOBS=window_debug=0 python tools/attn_v5_one.py > $out/r04_l_attn_nt0.log 2>&1; tail -5 $out/r04_l_attn_nt0.log
HFL_KNOBS=window_debug=16 python tools/attn_v5_one.py > $out/r04_l_attn_nt1.log 2>&1; tail -5 $out/r04_l_attn_nt1.log
rocprofv3 --kernel-trace --output-format csv -d $out/r04_l_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_l_stats.log 2>&1
trace=$(find $out/r04_l_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_l_phases_it7.log 2>&1
rm -rf $out/r04_l_stats
rocprofv3 --kernel-trace --output-format csv -d $out/r04_l_serial -- python bench.py --serial-streams --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_l_serial.log 2>&1
trace=$(find $out/r04_l_serial -name '*kernel_trace.csv' | head -1)
python tools/kernel_shapes.py "$trace" 10 gemm_x3 ln_mlp ln_qkv cpe_fwd window_attn > $out/r04_l_shapes.md 2>&1
rm -rf $out/r04_l_serial
cat $out/r04_l_shapes.md
