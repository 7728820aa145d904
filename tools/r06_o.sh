#!/bin/bash
# tap weight gradient with pipelined 16-byte operand reads; cost of the RPE-table gradient in the attention backward
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dwconv or tap or cpe or conv" > gpurun_out/r06_o_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_o_kernel_tests.log
for i in 1 2; do
  timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_o_train_new_$i.json 2>gpurun_out/r06_o_train_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_o_train_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
timeout 600 python tools/attn_bwd_bench.py cs-wild-places 64 8192 > gpurun_out/r06_o_attn_bwd_bench.log 2>&1; cat gpurun_out/r06_o_attn_bwd_bench.log
bash tools/prof_train.sh r06_o_train > gpurun_out/r06_o_prof.log 2>&1; head -22 gpurun_out/r06_o_prof.log
