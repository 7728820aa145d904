#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear_x3 or x3" > $out/r04_ac_tests.log 2>&1; tail -4 $out/r04_ac_tests.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or native_block" > $out/r04_ac_tests2.log 2>&1; tail -4 $out/r04_ac_tests2.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline $BARGS 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3 4; do
  run "default" A=1
done > $out/r04_ac_ab.log 2>&1
BARGS=--serial-streams run "serial streams" A=1 >> $out/r04_ac_ab.log 2>&1
cat $out/r04_ac_ab.log
rocprofv3 --kernel-trace --output-format csv -d $out/r04_ac_trace -- python bench.py --serial-streams --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $out/r04_ac_trace.log 2>&1
f=$(find $out/r04_ac_trace -name "*kernel_trace.csv" | head -1)
python tools/kernel_shapes.py $f 10 gemm_x3 > $out/r04_ac_serial_shapes.md
rm -rf $out/r04_ac_trace
head -30 $out/r04_ac_serial_shapes.md
