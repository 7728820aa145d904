#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 300 python tools/attn_fused_probe.py > $out/r04_o_probe.log 2>&1; tail -3 $out/r04_o_probe.log
HFL_VARIANTS=attn_fused_split=0 timeout 300 python tools/attn_fused_probe.py > $out/r04_o_probe_nosplit.log 2>&1; tail -3 $out/r04_o_probe_nosplit.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_fused" > $out/r04_o_tests.log 2>&1; tail -3 $out/r04_o_tests.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'])"
}
for i in 1 2 3; do
  run "default (split tail)" A=1
  run "attn fused no tail split" HFL_VARIANTS=attn_fused_split=0
done > $out/r04_o_ab.log 2>&1
cat $out/r04_o_ab.log
python bench.py --no-cpu-baseline --no-train-leg --no-oxford-leg --no-pinned-leg > $out/r04_o_bench.json 2> $out/r04_o_bench.err
python -c "
import json
j=json.loads(open('$out/r04_o_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['roofline']['frac'], j['roofline']['avg_launch_us'], j['roofline']['launches'])
print(json.dumps(j.get('roofline_fused'), indent=1))
"
