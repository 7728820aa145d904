#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
python -m pytest tests/test_gpu_model.py -x -q -k "relay_first or golden or early_phase or native_block or cu_partition" > $out/r04_k_tests.log 2>&1; tail -3 $out/r04_k_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'])"
}
for i in 1 2 3; do
  run "default (relay first)" A=1
  run "relay first off" HFL_RELAY_FIRST=0
  run "relay first + mlp reserve 16" HFL_VARIANTS=mlp_reserve=16
  run "relay first + mlp reserve 32" HFL_VARIANTS=mlp_reserve=32
  run "relay first + mlp,qkv reserve 16" HFL_VARIANTS=mlp_reserve=16,qkv_reserve=16
done > $out/r04_k_ab.log 2>&1
cat $out/r04_k_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04_k_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_k_stats.log 2>&1
trace=$(find $out/r04_k_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_k_phases_it7.log 2>&1
rm -rf $out/r04_k_stats
grep -v "^    " $out/r04_k_phases_it7.log
