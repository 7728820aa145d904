#!/bin/bash
# training glue: CPE writing the token rows of the block's buffer (no slices / concatenation), weight- and bias-gradient slabs reduced in one launch
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad or attention_backward or relay or linear_x3" > gpurun_out/r06_v_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_v_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_loss.py tests/test_variants.py -q -m gpu -k "backward or grad or train or multistaged or checkpoint" > gpurun_out/r06_v_grad_tests.log 2>&1; tail -5 gpurun_out/r06_v_grad_tests.log
for i in 1 2; do
  timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_v_train_new_$i.json 2>/dev/null
  HFL_PROBES=1 HFL_TRAIN_CPE_BUFFER=0 timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_v_train_cat_$i.json 2>/dev/null
done
HFL_CHECKPOINT=always timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_v_train_ckpt.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_v_train_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
