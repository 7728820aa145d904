#!/bin/bash
# hfl_slot_sum: tests, golden models, gradients, the step
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "slot or tap or conv or dwconv" > gpurun_out/r06_z_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_z_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_variants.py -q -m gpu -k "golden or backward or grad" > gpurun_out/r06_z_model_tests.log 2>&1; tail -3 gpurun_out/r06_z_model_tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_z_new_$i.json 2>/dev/null
done
timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_z_train.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_z_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
bash tools/profile_serial.sh r06_z > gpurun_out/r06_z_serial.log 2>&1; grep -i "slot_sum\|dwconv_fwd\|TOTAL" gpurun_out/r06_z_serial_summary_table.md
