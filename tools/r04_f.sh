#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "tap_lists or sparse_taps or ln_mlp_fused or ln_qkv" > $out/r04_f_tests.log 2>&1; tail -3 $out/r04_f_tests.log
python -m pytest tests/test_gpu_model.py -x -q -k "cu_partition or early_phase or merged_window or native_block or golden" > $out/r04_f_tests2.log 2>&1; tail -3 $out/r04_f_tests2.log
for i in 1 2; do
  for part in 192 0 160 208 224; do
    python bench.py --no-extras --no-cpu-baseline --cu-partition $part 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('partition $part', j['value'], j['ms_per_step'])"
  done
done > $out/r04_f_ab.log 2>&1
cat $out/r04_f_ab.log
rocprofv3 --kernel-trace --output-format csv -d $out/r04_f_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_f_stats.log 2>&1
trace=$(find $out/r04_f_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_f_phases_it7.log 2>&1
python tools/forward_phases.py "$trace" "tables" > $out/r04_f_phases_stem.log 2>&1
rm -rf $out/r04_f_stats
grep -v "^    " $out/r04_f_phases_it7.log
