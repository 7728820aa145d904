set -x
export HFL_PROBES=1   # the HFL_* schedule knobs below are probe switches (hotformerloc_amd/model.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_u_ab.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "cpe or forward_backward or checkpoint" 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r05_u_test.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --config cs-wild-places --train --steps 5 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'], d.get('peak_memory_GiB'))" >> gpurun_out/r05_u_ab.log; }
for i in 1 2 3; do
run gather X=1
run plain HFL_TRAIN_CPE_BWD_GATHER=0
run three HFL_TRAIN_CPE_FUSED=0
done
