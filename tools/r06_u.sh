#!/bin/bash
# persistent grids of the depth-wise kernels as whole multiples of the resident workgroups: tests + A/B is by kernel time
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dwconv or cpe or tap or conv" > gpurun_out/r06_u_kernel_tests.log 2>&1; tail -3 gpurun_out/r06_u_kernel_tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_u_new_$i.json 2>/dev/null
done
timeout 600 python bench.py --train --config cs-wild-places --steps 5 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06_u_train.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_u_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
bash tools/profile_serial.sh r06_u > gpurun_out/r06_u_serial.log 2>&1; grep -i "dwconv_fwd\|cpe_fwd\|TOTAL" gpurun_out/r06_u_serial_summary_table.md
