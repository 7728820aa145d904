#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "tap_lists or sparse_taps or ln_mlp_fused or ln_qkv or layer_norm or relay" > $out/r04_g_tests.log 2>&1; tail -3 $out/r04_g_tests.log
python -m pytest tests/test_gpu_model.py tests/test_variants.py -x -q -k "cu_partition or early_phase or merged_window or native_block or golden or stage_by_stage" > $out/r04_g_tests2.log 2>&1; tail -3 $out/r04_g_tests2.log
for i in 1 2 3; do
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default        ', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
  HFL_MLP_FUSED_MIN_ROWS=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlp fused all  ', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
done > $out/r04_g_ab.log 2>&1
cat $out/r04_g_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04_g_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_g_stats.log 2>&1
trace=$(find $out/r04_g_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_g_phases_it7.log 2>&1
python tools/forward_phases.py "$trace" "tables" > $out/r04_g_phases_stem.log 2>&1
python tools/forward_phases.py "$trace" "pooling" > $out/r04_g_phases_head.log 2>&1
rm -rf $out/r04_g_stats
grep -v "^    " $out/r04_g_phases_it7.log
python tools/mlp_fused_probe.py 2>&1 | grep "^rows" > $out/r04_g_mlp_probe.log; cat $out/r04_g_mlp_probe.log
