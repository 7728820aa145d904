#!/bin/bash
# RPE-table gradient of the attention backward on the matrix cores
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" > gpurun_out/r06_p_kernel_tests.log 2>&1; tail -5 gpurun_out/r06_p_kernel_tests.log
timeout 600 python tools/attn_bwd_bench.py cs-wild-places 64 8192 > gpurun_out/r06_p_attn_bwd_bench.log 2>&1; cat gpurun_out/r06_p_attn_bwd_bench.log
timeout 600 python tools/attn_bwd_bench.py wild-places 32 4096 > gpurun_out/r06_p_attn_bwd_bench_wp.log 2>&1; cat gpurun_out/r06_p_attn_bwd_bench_wp.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_loss.py -q -m gpu -k "backward or grad or train or multistaged or checkpoint" > gpurun_out/r06_p_grad_tests.log 2>&1; tail -5 gpurun_out/r06_p_grad_tests.log
for i in 1 2; do
  timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_p_train_new_$i.json 2>gpurun_out/r06_p_train_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_p_train_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
