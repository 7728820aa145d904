#!/bin/bash
# round-4 measurement batch C: tail split probes + tests of the touched kernels + step A/B + kernel traces
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python tools/ring_pf_probe.py > $out/r04_c_ring_pf.log 2>&1
python -m pytest tests/test_gpu_kernels.py -x -q -k "ln_mlp_fused or ln_qkv_fused" > $out/r04_c_tests.log 2>&1
tail -3 $out/r04_c_tests.log
for i in 1 2; do
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default            ', j['value'], j['ms_per_step'])"
  HFL_QKV_FUSED_MIN_FILL=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qkv fused always   ', j['value'], j['ms_per_step'])"
done > $out/r04_c_ab.log 2>&1
cat $out/r04_c_ab.log
python tools/torch_ops_probe.py > $out/r04_c_torch_ops.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04_c_stats -- python bench.py --steps 16 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_c_stats.log 2>&1
trace=$(find $out/r04_c_stats -name '*kernel_trace.csv' | head -1)
python tools/forward_phases.py "$trace" > $out/r04_c_phases.log 2>&1
stats=$(find $out/r04_c_stats -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/r04_c_kernel_stats.csv
python tools/summarize_rocprof.py $out/r04_c_kernel_stats.csv 18 > $out/r04_c_summary_table.md
rm -rf $out/r04_c_stats
tools/profile_serial.sh r04_c > /dev/null 2>&1
rm -rf $out/r04_c_serial_stats
head -30 $out/r04_c_serial_summary_table.md
cat $out/r04_c_phases.log
