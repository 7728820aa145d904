#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "default" A=1
  run "static units" HFL_VARIANTS=dynamic_units=0
  run "x3 ring off" HFL_VARIANTS=x3_ring=0
  run "cu reserve 16" HFL_VARIANTS=cu_reserve=16
  run "rtsa mlp fused" HFL_RTSA_MLP_FUSED=1
  run "mlp fused all rows" HFL_MLP_FUSED_MIN_ROWS=0
  run "qkv fused half-full rule" HFL_QKV_FUSED_MIN_FILL=0.5
  run "no tail split" HFL_VARIANTS=tail_split=0
done > $out/r04_i_ab.log 2>&1
cat $out/r04_i_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04_i_stats -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_i_stats.log 2>&1
trace=$(find $out/r04_i_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" "iteration 7" > $out/r04_i_phases_it7.log 2>&1
python tools/forward_phases.py "$trace" "tables" > $out/r04_i_phases_stem.log 2>&1
python tools/forward_phases.py "$trace" "pooling" > $out/r04_i_phases_head.log 2>&1
rm -rf $out/r04_i_stats
grep -v "^    " $out/r04_i_phases_it7.log
