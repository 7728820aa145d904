set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --no-train-leg --no-oxford-leg --no-pinned-leg > gpurun_out/r06_j_bench.json 2> gpurun_out/r06_j_bench.err
tail -4 gpurun_out/r06_j_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_j_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'matched', d['matched_precision']['value'], d['matched_precision']['ms_per_step'], 'lib', d['matched_precision']['fp32_library_gemm']['value'], 'x6 frac', d['matched_precision']['roofline']['frac'])
PY
