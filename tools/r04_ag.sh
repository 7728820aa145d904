#!/bin/bash
# per-round launches of the chip-filling row-tile kernels; Mixer layers on the fused MLP launch
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp_fused or qkv" > $out/r04_ag_tests.log 2>&1; tail -4 $out/r04_ag_tests.log
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "golden or native_block or early_phase or parity" > $out/r04_ag_tests2.log 2>&1; tail -4 $out/r04_ag_tests2.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "default (round launches, mixer)" A=1
  run "one launch per kernel" HFL_VARIANTS=round_launches=0
  run "mixer as three launches" HFL_MIXER_FUSED=0
done > $out/r04_ag_ab.log 2>&1
cat $out/r04_ag_ab.log
rocprofv3 --kernel-trace --output-format csv -d $out/r04_ag_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_ag_stats.log 2>&1
trace=$(find $out/r04_ag_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_ag_phases_it7.log 2>&1
python tools/forward_phases.py "$trace" "pooling" > $out/r04_ag_phases_head.log 2>&1
rm -rf $out/r04_ag_stats
grep "^    " $out/r04_ag_phases_it7.log | cut -c1-110
