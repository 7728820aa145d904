#!/bin/bash
# A/B: the finest level's CPE issued before the relay-token block on the main stream (plain schedule)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu -k "golden or schedule or executor" > gpurun_out/r06_aa_model_tests.log 2>&1; tail -3 gpurun_out/r06_aa_model_tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_aa_cpe_first_$i.json 2>/dev/null
  HFL_PROBES=1 HFL_CPE_FIRST=0 timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_aa_rtsa_first_$i.json 2>/dev/null
done
for i in 1 2; do
  timeout 300 python bench.py --config oxford --batch 64 --no-extras --no-cpu-baseline > gpurun_out/r06_aa_oxford_cpe_first_$i.json 2>/dev/null
  HFL_PROBES=1 HFL_CPE_FIRST=0 timeout 300 python bench.py --config oxford --batch 64 --no-extras --no-cpu-baseline > gpurun_out/r06_aa_oxford_rtsa_first_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_aa_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
