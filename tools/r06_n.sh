#!/bin/bash
# training path, second pass: tap weight gradient with 16-byte operand loads, depth-wise weight gradient reduced inside the workgroup
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dwconv or tap or cpe or conv or attention_backward" > gpurun_out/r06_n_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_n_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_loss.py -q -m gpu -k "backward or grad or train or multistaged or checkpoint" > gpurun_out/r06_n_grad_tests.log 2>&1; tail -5 gpurun_out/r06_n_grad_tests.log
for i in 1 2; do
  timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_n_train_new_$i.json 2>gpurun_out/r06_n_train_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_n_train_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
timeout 900 python tools/train_ops_profile.py > gpurun_out/r06_n_train_ops.txt 2> gpurun_out/r06_n_train_ops.err; head -70 gpurun_out/r06_n_train_ops.txt
