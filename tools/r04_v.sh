#!/bin/bash
# attentional pooling kernel: test + A/B; 4-wave LN->qkv probe
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
python -c "from hotformerloc_amd import _native; _native.load(); print('library ok')" || exit 1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_pool" > $out/r04_v_tests.log 2>&1; tail -15 $out/r04_v_tests.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or parity or oracle" > $out/r04_v_tests2.log 2>&1; tail -5 $out/r04_v_tests2.log
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$label', j['value'], j['ms_per_step'], j['host_issue']['ms_per_step_issue'])"
}
for i in 1 2 3; do
  run "attn_pool kernel (default)" A=1
  run "pooling as rounds 1-3" HFL_ATTN_POOL=0
done > $out/r04_v_ab.log 2>&1
cat $out/r04_v_ab.log
timeout 600 python tools/mlp_waves_probe.py > $out/r04_v_probe.log 2>&1; cat $out/r04_v_probe.log | tail -9
