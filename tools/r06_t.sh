#!/bin/bash
# inference: relay-token attention as one memory round trip per item, Mixer tail on 16 waves; A/B of the relay kernel in the step
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "relay or mixer or init" > gpurun_out/r06_t_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_t_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu -k "golden" > gpurun_out/r06_t_golden_tests.log 2>&1; tail -3 gpurun_out/r06_t_golden_tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_t_ab_new_$i.json 2>/dev/null
  HFL_VARIANTS="relay_fast=0" timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_t_ab_relay_general_$i.json 2>/dev/null
done
timeout 300 python bench.py --gemm x6 --no-extras --no-cpu-baseline > gpurun_out/r06_t_x6.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_t_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
bash tools/profile_serial.sh r06_t > gpurun_out/r06_t_serial.log 2>&1; grep -i "mixer_tail\|relay_attn\|TOTAL" gpurun_out/r06_t_serial_summary_table.md
