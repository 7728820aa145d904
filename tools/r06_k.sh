set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "x6" > gpurun_out/r06_k_x6_tests.log 2>&1
tail -4 gpurun_out/r06_k_x6_tests.log
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "golden" > gpurun_out/r06_k_golden_tests.log 2>&1
tail -4 gpurun_out/r06_k_golden_tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_k_ab_default_$i.json 2>/dev/null
  HFL_PROBES=1 HFL_MAIN_HI=1 timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_k_ab_mainhi_$i.json 2>/dev/null
done
timeout 300 python bench.py --gemm x6 --no-extras --no-cpu-baseline > gpurun_out/r06_k_x6.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_k_ab_*.json')) + ['gpurun_out/r06_k_x6.json']:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
