set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "early_phase or native_block" 2>&1 | tail -3 > gpurun_out/r05_j_test.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/r05_j_ab.log; }
for i in 1 2 3; do
run reserve0 X=1
run reserve4 HFL_VARIANTS=cu_reserve=4
run reserve8 HFL_VARIANTS=cu_reserve=8
run reserve16 HFL_VARIANTS=cu_reserve=16
done
